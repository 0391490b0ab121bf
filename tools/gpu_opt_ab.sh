#!/bin/bash
# same-box A/B of one context option through bench.py: bash tools/gpu_opt_ab.sh <option> [configs...]   -> gpurun_out/ab_<option>.txt
export TMPDIR=/tmp; mkdir -p gpurun_out
OPT=$1; shift
: > gpurun_out/ab_$OPT.txt
for c in ${@:-smmnist_big5_ngf96}; do
for rep in 1 2; do for v in 0 1; do
  MCVD_BENCH_OPTS=$OPT=$v timeout 600 python bench.py --config $c --steps 2 --warmup 1 --no-cpu-baseline --no-f16x2-leg > gpurun_out/bench_ab.json 2> gpurun_out/bench_ab.err
  python - <<PY | tee -a gpurun_out/ab_$OPT.txt
import json
d=json.load(open('gpurun_out/bench_ab.json'))
print('$c $OPT=$v', d['value'], 'frames/s', d['ms_per_step'], 'ms/step', d['valid'])
PY
done; done; done

#!/bin/bash
# same-box A/B of two builds of the library: mcvd_pytorch_amd/libmcvd_hip_prev.so (built from the previous commit) against the in-tree one.
# Usage (one gpurun call): bash tools/gpu_ab_lib.sh [config]        -> gpurun_out/ab_lib.txt
mkdir -p gpurun_out
export TMPDIR=/tmp
CFG=${1:-smmnist_big5_ngf96}
: > gpurun_out/ab_lib.txt
for rep in 1 2; do
for v in prev new; do
  if [ $v = prev ]; then export MCVD_LIB_PATH=$PWD/mcvd_pytorch_amd/libmcvd_hip_prev.so; else unset MCVD_LIB_PATH; fi
  timeout 900 python bench.py --config $CFG --steps 3 --warmup 1 --no-cpu-baseline --no-f16x2-leg > gpurun_out/bench_ab.json 2> gpurun_out/bench_ab.err
  python - <<PY | tee -a gpurun_out/ab_lib.txt
import json
d=json.load(open('gpurun_out/bench_ab.json'))
print('$CFG $v', d['value'], 'frames/s', d['ms_per_step'], 'ms/step', {k:(v['launches'],v['ms']) for k,v in d['roofline']['breakdown'].items() if k in ('conv3x3','conv1x1','attention','gn_coef')}, 'valid', d['valid'])
PY
done
done

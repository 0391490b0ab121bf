#!/bin/bash
# same-box A/B of two diagnostics builds (libmcvd_hip_diagA.so = HEAD, libmcvd_hip_diagB.so = working tree): K-loop cycles + bench
mkdir -p gpurun_out
export TMPDIR=/tmp
for rep in 1 2; do
for v in A B; do
  export MCVD_LIB_PATH=$PWD/mcvd_pytorch_amd/libmcvd_hip_diag$v.so
  MCVD_WEXP_ONLY=0 MCVD_WEXP_CASES=${CASES:-0,1,2} timeout 600 python tests/gpu_diag.py w3exp > gpurun_out/w3exp.log 2>&1; echo "== $v"; grep "exp   0" gpurun_out/diag_w3exp.txt | cut -c1-200
  [ -n "$NOBENCH" ] || timeout 900 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-f16x2-leg > gpurun_out/bench_ab.json 2> gpurun_out/bench_ab.err; python - <<PY
import json
d=json.load(open('gpurun_out/bench_ab.json'))
print('$v', d['value'], d['ms_per_step'], {k:(v['launches'],v['ms']) for k,v in d['roofline']['breakdown'].items() if k in ('conv3x3','conv1x1','attention')})
PY
done
done

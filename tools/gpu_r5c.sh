#!/bin/bash
# round 5, call C: patch pitch 25 vs 24 (same box), fused-path fixes
mkdir -p gpurun_out
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1; tail -1 gpurun_out/build.log
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_wino3p.py tests/test_gpu_ksplit_deep.py -m gpu -q --tb=short -p no:cacheprovider -x -k "presplit or spade or wino3p or ksplit or conv2d or bit_deterministic or bench_kernel_table" > gpurun_out/pytest_new.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_new.log; tail -8 gpurun_out/pytest_new.log
for rep in 1 2; do
for v in pp24 pp25; do
  if [ $v = pp24 ]; then export MCVD_LIB_PATH=$PWD/mcvd_pytorch_amd/libmcvd_hip_pp24.so; else unset MCVD_LIB_PATH; fi
  timeout 600 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-f16x2-leg > gpurun_out/bench_$v.json 2> gpurun_out/bench_$v.err
  python - <<PY
import json
d=json.load(open('gpurun_out/bench_$v.json'))
print('$v rep$rep', d['value'], d['ms_per_step'], 'selfcheck', d['selfcheck_max_abs'], {k:(v['launches'],v['ms']) for k,v in d['roofline']['breakdown'].items() if k in ('conv3x3','conv1x1','attention')})
PY
done
done
for v in pp24 pp25; do
  if [ $v = pp24 ]; then export MCVD_LIB_PATH=$PWD/mcvd_pytorch_amd/libmcvd_hip_pp24.so; else unset MCVD_LIB_PATH; fi
  MCVD_TL_CASES=0,1,5,6,9 timeout 300 python tests/gpu_diag.py w3ptl > gpurun_out/w3ptl_$v.log 2>&1; echo "== $v"; grep "cin" gpurun_out/diag_w3ptl.txt 2>/dev/null | cut -c1-330 || tail -8 gpurun_out/w3ptl_$v.log
  cp gpurun_out/diag_w3ptl.txt gpurun_out/diag_w3ptl_$v.txt 2>/dev/null
done
unset MCVD_LIB_PATH
for nf in 1 0; do
  MCVD_BENCH_OPTS=spade_norm_fuse=$nf,spade_fuse_auto=0 timeout 600 python bench.py --config bair_big_spade --steps 1 --warmup 1 --subsample 200 --no-cpu-baseline --no-f16x2-leg > gpurun_out/bench_sn$nf.json 2> gpurun_out/bench_sn$nf.err
  python - <<PY
import json
d=json.load(open('gpurun_out/bench_sn$nf.json'))
print('spade_norm_fuse=$nf', d['value'], d['ms_per_step'], {k:(v['launches'],v['ms']) for k,v in d['roofline']['breakdown'].items() if k in ('conv3x3','gn_coef')})
PY
done
for ps in 1 0; do
  MCVD_BENCH_OPTS=attn_presplit=$ps timeout 600 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-f16x2-leg > gpurun_out/bench_ps$ps.json 2> gpurun_out/bench_ps$ps.err
  python - <<PY
import json
d=json.load(open('gpurun_out/bench_ps$ps.json'))
print('presplit=$ps', d['value'], d['ms_per_step'], {k:(v['launches'],v['ms']) for k,v in d['roofline']['breakdown'].items() if k in ('conv1x1','attention')})
PY
done

#!/bin/bash
# conv parity subset, K-loop cycles of conv_wino3 (diagnostics build), bench without the f16x2 leg -- the loop while tuning the kernel
mkdir -p gpurun_out
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1; tail -1 gpurun_out/build.log
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q --tb=short -p no:cacheprovider -x -k "conv or gn_coefficients" > gpurun_out/pytest_n.log 2>&1; tail -3 gpurun_out/pytest_n.log
MCVD_LIB_PATH=$PWD/mcvd_pytorch_amd/libmcvd_hip_diag.so MCVD_WEXP_ONLY=0 MCVD_WEXP_CASES=0,1,2,3 timeout 600 python tests/gpu_diag.py w3exp > gpurun_out/w3exp.log 2>&1; cat gpurun_out/diag_w3exp.txt | cut -c1-240; tail -3 gpurun_out/w3exp.log
timeout 900 python bench.py --steps 2 --warmup 1 --no-cpu-baseline ${BENCH_EXTRA:---no-f16x2-leg} > gpurun_out/bench_n.json 2> gpurun_out/bench_n.err; python - <<'PY'
import json
d=json.load(open('gpurun_out/bench_n.json'))
print(d['value'], d['ms_per_step'], d['roofline']['frac'], {k:(v['launches'],v['ms']) for k,v in d['roofline']['breakdown'].items()}, 'f16x2', d.get('f16x2_leg',{}).get('value'))
PY
tail -2 gpurun_out/bench_n.err

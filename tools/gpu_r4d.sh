#!/bin/bash
# persistent Winograd kernel: bit-identity tests, layer-by-layer A/B, stagger experiment (diag build), bench with the autotuner
mkdir -p gpurun_out
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1; tail -1 gpurun_out/build.log
timeout 900 python -m pytest tests/test_gpu_wino3p.py -m gpu -q --tb=short -p no:cacheprovider -x > gpurun_out/pytest_w3p.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_w3p.log; tail -4 gpurun_out/pytest_w3p.log
MCVD_TL_CASES=${CASES:-0,1,2,5,6,7,9} timeout 600 python tests/gpu_diag.py w3ptl > gpurun_out/w3ptl.log 2>&1; cat gpurun_out/diag_w3ptl.txt | cut -c1-420; tail -3 gpurun_out/w3ptl.log
for st in 0 40 80; do
  echo "== stagger $st x 256 cycles (diag build)"
  MCVD_LIB_PATH=$PWD/mcvd_pytorch_amd/libmcvd_hip_diag.so MCVD_W3P_STAGGER=$st MCVD_TL_CASES=0,5 timeout 600 python tests/gpu_diag.py w3ptl > gpurun_out/w3ptl_s$st.log 2>&1; cp gpurun_out/diag_w3ptl.txt gpurun_out/diag_w3ptl_stagger$st.txt; cat gpurun_out/diag_w3ptl.txt | cut -c1-420
done
timeout 900 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-f16x2-leg --no-tune-file --save-tuning gpurun_out/tune > gpurun_out/bench_r4b.json 2> gpurun_out/bench_r4b.err; echo "bench rc=$?" >> gpurun_out/bench_r4b.err; python - <<'PY'
import json
d=json.load(open('gpurun_out/bench_r4b.json'))
print(d['value'], d['ms_per_step'], d['roofline']['frac'], 'selfcheck', d['selfcheck_max_abs'], {k:(v['launches'],v['ms']) for k,v in d['roofline']['breakdown'].items()})
t=json.load(open('gpurun_out/tune/tune_smmnist_big5_ngf96_B64_bf16x3.json'))['64']
import collections
print(collections.Counter(x[0] for x in t))
PY
tail -2 gpurun_out/bench_r4b.err

#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1; tail -1 gpurun_out/build.log
timeout 600 python -m pytest tests/test_gpu_parity.py -q -x -p no:cacheprovider --tb=short -k "(test_conv2d and mfma and (s30 or s46 or s62 or s78 or s31 or s47 or s63)) or test_conv1x1_split or (test_default_kernels and conv1x1) or test_conv_epilogue" > gpurun_out/pytest_e.log 2>&1; echo "pytest_e rc=$?" >> gpurun_out/pytest_e.log; tail -5 gpurun_out/pytest_e.log
MCVD_TL_CASES=${CASES:-2,3,4,5} timeout 300 python tests/gpu_diag.py w2htl > gpurun_out/w2htl.log 2>&1; grep -- "---" gpurun_out/diag_w2htl.txt | cut -c1-420; tail -2 gpurun_out/w2htl.log
timeout 600 python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-f16x2-leg > gpurun_out/bench_b.json 2> gpurun_out/bench_b.err
echo "bench rc=$?"; python - <<'PY'
import json
d=json.load(open('gpurun_out/bench_b.json'))
print('value', d['value'], 'ms', d['ms_per_step'], 'frac', d['roofline']['frac'])
print({k:(v['launches'],v['ms']) for k,v in d['roofline']['breakdown'].items()})
PY
tail -3 gpurun_out/bench_b.err

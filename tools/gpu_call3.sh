#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -p no:cacheprovider -k "(test_conv2d and mfma and (s12 or s13)) or f16x2" > gpurun_out/pytest_f16x2.log 2>&1 ); echo "pytest rc=$?"; tail -3 gpurun_out/pytest_f16x2.log
MCVD_WEXP_SHAPE=12 MCVD_WEXP_ONLY=${WEXP:-0,4,16,27} timeout 300 python tests/gpu_diag.py w3exp > gpurun_out/w2hexp.log 2>&1; cp gpurun_out/diag_w3exp.txt gpurun_out/diag_w2hexp.txt; cut -c1-230 gpurun_out/diag_w2hexp.txt; tail -3 gpurun_out/w2hexp.log
python tests/gpu_diag.py w2htl > gpurun_out/w2htl.log 2>&1; cat gpurun_out/diag_w2htl.txt; tail -3 gpurun_out/w2htl.log
timeout 600 python bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/bench_f16x2.json 2> gpurun_out/bench_f16x2.err; echo "bench rc=$?"; python - <<'PY'
import json
d=json.load(open('gpurun_out/bench_f16x2.json'))
print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['avg_launch_us'], {k:(v['launches'],v['ms']) for k,v in d['roofline']['breakdown'].items()})
PY

#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -s -p no:cacheprovider -k "(test_conv2d and mfma and s14) or conv1x1_f16x2" > gpurun_out/pytest_q1.log 2>&1 ); echo "pytest rc=$?"; grep -E "accuracy|passed|failed|rror" gpurun_out/pytest_q1.log | sort -u | head -20; tail -12 gpurun_out/pytest_q1.log
( timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -p no:cacheprovider -k "test_forward_vs_reference_golden and mfma" > gpurun_out/pytest_fwd.log 2>&1 ); echo "pytest fwd rc=$?"; tail -5 gpurun_out/pytest_fwd.log
timeout 600 python bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/bench_f16x2.json 2> gpurun_out/bench_f16x2.err; echo "bench rc=$?"; python - <<'PY'
import json
d=json.load(open('gpurun_out/bench_f16x2.json'))
print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['avg_launch_us'], {k:(v['launches'],v['ms'],v['gbs']) for k,v in d['roofline']['breakdown'].items()})
PY
tail -3 gpurun_out/bench_f16x2.err

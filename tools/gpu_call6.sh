#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_gpu_parity.py -q -p no:cacheprovider -k "test_attention or (test_forward_vs_reference_golden and mfma)" > gpurun_out/pytest_attn.log 2>&1 ); echo "pytest rc=$?"; tail -25 gpurun_out/pytest_attn.log | cut -c1-300
timeout 600 python bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/bench_attn.json 2> gpurun_out/bench_attn.err; echo "bench rc=$?"; python - <<'PY'
import json
d=json.load(open('gpurun_out/bench_attn.json'))
print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['avg_launch_us'], {k:(v['launches'],v['ms'],v['tflops']) for k,v in d['roofline']['breakdown'].items()})
PY
tail -3 gpurun_out/bench_attn.err

#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -p no:cacheprovider -k "(test_conv2d and mfma and (s30 or s46 or s62 or s78)) or conv1x1_f16x2" > gpurun_out/pytest_q1.log 2>&1 ); echo "pytest rc=$?"; tail -5 gpurun_out/pytest_q1.log
timeout 300 python tests/gpu_diag.py convops > gpurun_out/convops.log 2>&1; grep " 1x1 " gpurun_out/diag_convops.txt | awk '{print $2,$4,$6,$8,$10,$12,$13,$14,$15,$16,$17,$18}' | head -50; tail -2 gpurun_out/diag_convops.txt
timeout 600 python bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/bench_f16x2.json 2> gpurun_out/bench_f16x2.err; echo "bench rc=$?"; python - <<'PY'
import json
d=json.load(open('gpurun_out/bench_f16x2.json'))
print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['avg_launch_us'], {k:(v['launches'],v['ms'],v['gbs']) for k,v in d['roofline']['breakdown'].items()})
PY

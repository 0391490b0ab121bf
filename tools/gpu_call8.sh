#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_gpu_parity.py -q -p no:cacheprovider -k "(test_conv2d and mfma and (s12 or s13)) or test_conv_f16x2 or (test_forward_vs_reference_golden and (f16x2 or mfma)) or test_f16x2_option or (test_forward_is_bit_deterministic and (12 or 13)) or test_gn_stats" > gpurun_out/pytest_f16x2.log 2>&1 ); echo "pytest rc=$?"; tail -12 gpurun_out/pytest_f16x2.log | cut -c1-400
timeout 300 python tests/gpu_diag.py convops > gpurun_out/convops.log 2>&1; grep " 3x3 H  8" gpurun_out/diag_convops.txt | head -20; tail -2 gpurun_out/diag_convops.txt
timeout 600 python bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/bench_x.json 2> gpurun_out/bench_x.err; echo "bench rc=$?"; python - <<'PY'
import json
d=json.load(open('gpurun_out/bench_x.json'))
print(d['value'], d['ms_per_step'], d.get('fp32_exact_leg'), d['roofline']['frac'], d['roofline']['avg_launch_us'], {k:(v['launches'],v['ms']) for k,v in d['roofline']['breakdown'].items()})
PY
tail -3 gpurun_out/bench_x.err

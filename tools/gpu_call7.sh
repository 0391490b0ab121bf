#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -p no:cacheprovider -k "(test_conv2d and mfma and (s12 or s13)) or test_conv_f16x2 or (test_forward_vs_reference_golden and f16x2) or (test_forward_is_bit_deterministic and (12 or 13))" > gpurun_out/pytest_f16x2.log 2>&1 ); echo "pytest rc=$?"; tail -4 gpurun_out/pytest_f16x2.log
python tests/gpu_diag.py w2hsub > gpurun_out/w2hsub.log 2>&1; cat gpurun_out/diag_w2hsub.txt; tail -2 gpurun_out/w2hsub.log
timeout 600 python bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/bench_x.json 2> gpurun_out/bench_x.err; echo "bench rc=$?"; python - <<'PY'
import json
d=json.load(open('gpurun_out/bench_x.json'))
print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['avg_launch_us'], {k:(v['launches'],v['ms']) for k,v in d['roofline']['breakdown'].items()})
PY

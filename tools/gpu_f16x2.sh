#!/bin/bash
# Two-piece fp16 kernels (conv_wino2h / conv1x1_h2 / attention_h2): parity subset, accuracy vs fp64, per-layer kernel table, K-loop and
# prologue cycle breakdowns, per-workgroup timelines, bench.  One gpurun call; results under gpurun_out/.
mkdir -p gpurun_out
export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_gpu_parity.py -q -s -p no:cacheprovider -k "(test_conv2d and mfma and (s12 or s13 or s30 or s46 or s62 or s78)) or f16x2 or test_attention or statistics or (test_forward_is_bit_deterministic and (12 or 13)) or rccl" > gpurun_out/pytest_f16x2.log 2>&1 ); echo "pytest rc=$?"; grep -E "accuracy" gpurun_out/pytest_f16x2.log | sort -u; tail -4 gpurun_out/pytest_f16x2.log | cut -c1-300
timeout 300 python tests/gpu_diag.py convops > gpurun_out/convops.log 2>&1; tail -2 gpurun_out/diag_convops.txt
MCVD_WEXP_SHAPE=12 MCVD_WEXP_ONLY=${WEXP:-0,4,15,16,27} MCVD_WEXP_CASES=${CASES:-all} timeout 300 python tests/gpu_diag.py w3exp > gpurun_out/w2hexp.log 2>&1; cp gpurun_out/diag_w3exp.txt gpurun_out/diag_w2hexp.txt; cut -c1-230 gpurun_out/diag_w2hexp.txt
python tests/gpu_diag.py w2hsub w2htl > gpurun_out/w2h_diag.log 2>&1; cut -c1-300 gpurun_out/diag_w2hsub.txt gpurun_out/diag_w2htl.txt
timeout 600 python bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/bench_f16x2.json 2> gpurun_out/bench_f16x2.err; echo "bench rc=$?"; python - <<'PY'
import json
d=json.load(open('gpurun_out/bench_f16x2.json'))
print(d['value'], d['ms_per_step'], d.get('fp32_exact_leg'), d['roofline']['frac'], d['roofline']['avg_launch_us'], {k:(v['launches'],v['ms']) for k,v in d['roofline']['breakdown'].items()})
PY

#!/bin/bash
# f16x2 Winograd kernel bring-up: parity cases, accuracy vs fp64, forced-forward fixtures, determinism, K-loop timing, bench A/B
mkdir -p gpurun_out
export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -s -p no:cacheprovider -k "(test_conv2d and mfma and (s12 or s13)) or f16x2 or test_forward_is_bit_deterministic" > gpurun_out/pytest_f16x2.log 2>&1 ); echo "pytest rc=$?"; grep -E "f16x2 accuracy|passed|failed|Error|error" gpurun_out/pytest_f16x2.log | head -40; tail -15 gpurun_out/pytest_f16x2.log
MCVD_WEXP_SHAPE=12 timeout 300 python tests/gpu_diag.py w3exp > gpurun_out/w2hexp.log 2>&1; cp gpurun_out/diag_w3exp.txt gpurun_out/diag_w2hexp.txt; cut -c1-230 gpurun_out/diag_w2hexp.txt; tail -3 gpurun_out/w2hexp.log
timeout 300 python tests/gpu_diag.py convops > gpurun_out/convops.log 2>&1; cp gpurun_out/diag_convops.txt gpurun_out/diag_convops_f16x2.txt; tail -62 gpurun_out/diag_convops_f16x2.txt
timeout 600 python bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/bench_f16x2.json 2> gpurun_out/bench_f16x2.err; echo "bench rc=$?"; cut -c1-2500 gpurun_out/bench_f16x2.json; tail -3 gpurun_out/bench_f16x2.err

#!/bin/bash
# bf16x3 Winograd kernel iteration: parity cases + accuracy vs fp64 + K-loop timing
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_parity.py -q -x -p no:cacheprovider -k "(test_conv2d and mfma and (s10 or s11)) or bf16x3" 2>&1 | tail -5
MCVD_WEXP_ONLY=${WEXP:-0} MCVD_WEXP_CASES=${CASES:-all} timeout 300 python tests/gpu_diag.py w3exp > gpurun_out/w3exp.log 2>&1; cat gpurun_out/diag_w3exp.txt | cut -c1-230; tail -3 gpurun_out/w3exp.log

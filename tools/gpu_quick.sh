#!/bin/bash
# quick kernel iteration: Winograd conv parity cases + the K-loop timing probe
mkdir -p gpurun_out
MCVD_WINO_EXP=${PYEXP:-0} timeout 300 python -m pytest tests/test_gpu_parity.py -q -x -p no:cacheprovider -k "test_conv2d and mfma and (s4 or s8)" 2>&1 | tail -3
MCVD_WEXP_ONLY=${WEXP:-0} MCVD_WEXP_CASES=${CASES:-all} timeout 300 python tests/gpu_diag.py wexp > gpurun_out/wexp.log 2>&1; cat gpurun_out/diag_wexp.txt | cut -c1-220; tail -2 gpurun_out/wexp.log

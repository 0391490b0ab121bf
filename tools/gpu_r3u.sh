#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
export MCVD_LIB_PATH=$PWD/mcvd_pytorch_amd/libmcvd_hip_diag.so
timeout 600 python tests/gpu_diag.py w3sub > gpurun_out/w3sub.log 2>&1; cat gpurun_out/diag_w3sub.txt | cut -c1-330; tail -3 gpurun_out/w3sub.log

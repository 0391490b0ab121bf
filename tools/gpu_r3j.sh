#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1; tail -1 gpurun_out/build.log
MCVD_LIB_PATH=$PWD/mcvd_pytorch_amd/libmcvd_hip_diag.so MCVD_WEXP_ONLY=0,256 MCVD_WEXP_CASES=1,2,3 timeout 600 python tests/gpu_diag.py w3exp > gpurun_out/w3exp.log 2>&1; cat gpurun_out/diag_w3exp.txt | cut -c1-240; tail -3 gpurun_out/w3exp.log

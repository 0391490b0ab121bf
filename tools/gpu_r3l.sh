#!/bin/bash
# round 3, step l: where the wino3 prologue goes; 1x1 timeline after the epilogue / coefficient-DMA change
mkdir -p gpurun_out
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1; tail -1 gpurun_out/build.log
export MCVD_LIB_PATH=$PWD/mcvd_pytorch_amd/libmcvd_hip_diag.so
timeout 600 python tests/gpu_diag.py w3pro > gpurun_out/w3pro.log 2>&1; cat gpurun_out/diag_w3pro.txt | cut -c1-260; tail -3 gpurun_out/w3pro.log
MCVD_TL_CASES=2,3,4,5 MCVD_TL_ACT=0 timeout 600 python tests/gpu_diag.py w2htl > gpurun_out/w2htl.log 2>&1; cat gpurun_out/diag_w2htl.txt | cut -c1-400; tail -3 gpurun_out/w2htl.log

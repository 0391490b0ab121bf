#!/bin/bash
# round 3: K-loop ablations of the three-piece bf16 Winograd kernel (diagnostics library) + f16x2 regression after the clamp removal
mkdir -p gpurun_out
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1; echo "build rc=$?" >> gpurun_out/build.log; tail -2 gpurun_out/build.log
MCVD_LIB_PATH=$PWD/mcvd_pytorch_amd/libmcvd_hip_diag.so MCVD_WEXP_CASES=${CASES:-0,2} timeout 600 python tests/gpu_diag.py w3exp > gpurun_out/w3exp.log 2>&1; cat gpurun_out/diag_w3exp.txt | cut -c1-240; tail -3 gpurun_out/w3exp.log
timeout 600 python -m pytest tests/test_gpu_parity.py -q -x -p no:cacheprovider --tb=short -k "(test_conv2d and mfma and (s12 or s13 or s46 or s62 or s30 or s78)) or test_conv_f16x2 or test_wino2h or (test_attention and f16x2)" > gpurun_out/pytest_c.log 2>&1; echo "pytest_c rc=$?" >> gpurun_out/pytest_c.log; tail -5 gpurun_out/pytest_c.log

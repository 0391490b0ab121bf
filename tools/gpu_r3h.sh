#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1; tail -1 gpurun_out/build.log
timeout 600 python -m pytest tests/test_gpu_parity.py -q -x -p no:cacheprovider --tb=short -k "test_f16x2_overflow or test_bench_kernel_table or test_f16x2_option" > gpurun_out/pytest_f.log 2>&1; echo "pytest_f rc=$?" >> gpurun_out/pytest_f.log; tail -12 gpurun_out/pytest_f.log
for e in 0 1 2 4 6 7 15; do
  MCVD_Q1_EXP=$e MCVD_TL_ACT=0 MCVD_LIB_PATH=$PWD/mcvd_pytorch_amd/libmcvd_hip_diag.so MCVD_TL_CASES=3,4 timeout 200 python tests/gpu_diag.py w2htl > gpurun_out/w2htl.log 2>&1; grep -- "---" gpurun_out/diag_w2htl.txt | cut -c1-400; tail -1 gpurun_out/w2htl.log | grep -i error
done

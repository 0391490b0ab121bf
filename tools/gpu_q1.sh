#!/bin/bash
# conv1x1_h2 (shape 15): parity subset, then per-chunk cycles of the K loop under the diagnostics build's ablations (MCVD_Q1_EXP)
mkdir -p gpurun_out
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1; tail -1 gpurun_out/build.log
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q --tb=short -p no:cacheprovider -x -k "conv or gn_coefficients or bf16x3" > gpurun_out/pytest_n.log 2>&1; tail -3 gpurun_out/pytest_n.log
: > gpurun_out/q1_exp.txt
for e in ${EXPS:-0 16}; do
  MCVD_LIB_PATH=$PWD/mcvd_pytorch_amd/libmcvd_hip_diag.so MCVD_Q1_EXP=$e MCVD_TL_CASES=${CASES:-3,4,5} MCVD_TL_ACT=0 timeout 300 python tests/gpu_diag.py w2htl > gpurun_out/w2htl.log 2>&1
  grep "^---" gpurun_out/diag_w2htl.txt | cut -c1-330 >> gpurun_out/q1_exp.txt
done
cat gpurun_out/q1_exp.txt
MCVD_LIB_PATH=$PWD/mcvd_pytorch_amd/libmcvd_hip_diag.so MCVD_Q1_NOPRIO=1 timeout 300 python tests/gpu_diag.py convops > gpurun_out/convops_old.log 2>&1; cp gpurun_out/diag_convops.txt gpurun_out/convops_q1_old.txt
MCVD_LIB_PATH=$PWD/mcvd_pytorch_amd/libmcvd_hip_diag.so timeout 300 python tests/gpu_diag.py convops > gpurun_out/convops_new.log 2>&1; cp gpurun_out/diag_convops.txt gpurun_out/convops_q1_new.txt
grep -h "total\|1x1" gpurun_out/convops_q1_old.txt | tail -4; grep -h "total\|1x1" gpurun_out/convops_q1_new.txt | tail -4

#!/bin/bash
# round 3, step p: 1x1 GEMM with three workgroups per CU (3 LDS buffers) where registers and LDS allow -- A/B inside one box
mkdir -p gpurun_out
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1; tail -1 gpurun_out/build.log
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q --tb=short -p no:cacheprovider -x -k "conv1x1 or (conv and 1x1) or conv_vs" > gpurun_out/pytest_p.log 2>&1; tail -3 gpurun_out/pytest_p.log
export MCVD_LIB_PATH=$PWD/mcvd_pytorch_amd/libmcvd_hip_diag.so
for occ in 2 3 2 3; do
  MCVD_Q1_OCC=$occ timeout 600 python tests/gpu_diag.py convops > gpurun_out/convops.log 2>&1
  cp gpurun_out/diag_convops.txt gpurun_out/diag_convops_occ$occ.txt
  echo "occ $occ: $(grep totals gpurun_out/diag_convops.txt)"
done
grep " 1x1 " gpurun_out/diag_convops_occ2.txt | cut -c1-75 > /tmp/a.txt; grep " 1x1 " gpurun_out/diag_convops_occ3.txt | cut -c36-75 > /tmp/b.txt; paste /tmp/a.txt /tmp/b.txt

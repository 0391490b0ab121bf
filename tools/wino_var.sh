#!/bin/bash
# Time the Winograd software-pipeline variants (MCVD_WINO_VAR) on the per-op profile of the BASELINE config-2 forward.
mkdir -p gpurun_out
for v in ${@:-0 1 2 3}; do
  MCVD_WINO_VAR=$v timeout 300 python tests/gpu_diag.py ops > gpurun_out/diag_ops_var$v.log 2>&1
  cp gpurun_out/diag_ops.txt gpurun_out/diag_ops_var$v.txt
  echo "VAR $v: $(grep 's4c' gpurun_out/diag_ops_var$v.txt | awk '{n+=1; t+=$(NF-5)} END {print n, "winograd launches", t/1000, "ms"}') total $(tail -2 gpurun_out/diag_ops_var$v.txt | head -1)"
done

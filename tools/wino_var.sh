#!/bin/bash
# Time Winograd kernel variants on the per-op profile of the BASELINE config-2 forward.
# usage: wino_var.sh "VAR PIPE" ...   (MCVD_WINO_VAR: 0-3 = 512-thread pipeline variants, 4 = 1024-thread kernel; MCVD_WINO_PIPE 0/1)
mkdir -p gpurun_out
for vp in "${@:-"4 1"}"; do
  set -- $vp; v=$1; p=${2:-1}
  MCVD_WINO_VAR=$v MCVD_WINO_PIPE=$p timeout 300 python tests/gpu_diag.py ops > gpurun_out/diag_ops_var${v}_$p.log 2>&1
  cp gpurun_out/diag_ops.txt gpurun_out/diag_ops_var${v}_$p.txt
  echo "VAR $v PIPE $p: $(grep 's4c' gpurun_out/diag_ops_var${v}_$p.txt | awk '{n+=1; t+=$(NF-5)} END {print n, "winograd launches", t/1000, "ms"}') total $(tail -2 gpurun_out/diag_ops_var${v}_$p.txt | head -1)"
done

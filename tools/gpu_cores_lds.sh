#!/bin/bash
mkdir -p gpurun_out
{
echo "#### other instructions that read one register (pair) twice, beside the 16x16x32 bf16 MFMA loop and beside the attention kernel"
AITER=400 AGRID=512 AGGRS="11 0" VARS="14 15 16 17 6" timeout 200 tools/bin/repro_coresident_bisect 2
} > gpurun_out/cores_forms.txt 2>&1
tail -5 gpurun_out/cores_forms.txt

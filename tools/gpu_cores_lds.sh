#!/bin/bash
mkdir -p gpurun_out
L=$PWD/mcvd_pytorch_amd/libmcvd_hip.so; LP=$PWD/mcvd_pytorch_amd/libmcvd_hip_prev.so; R=tools/bin/repro_coresident_base
run() { lib=$1; shift; echo "## $(basename $lib) $*"; env "$@" ONLY_LIB=1 timeout 120 $R ${SECS:-2} $lib | grep -v "^# aggressor attn\|first bad launch" | cut -c1-200; }
{
echo "#### the stand-alone reproducer (tools/repro_pk_fma_beside_mfma.cpp)"
timeout 100 tools/bin/repro_pk_fma_beside_mfma 2; echo "exit code $?"
echo "#### the library's former victims, previous build (v_pk_fma_f32 d, x, c, c in the prologue) and this build (fma_unpacked)"
for lib in $LP $L; do
run $lib VICTIM_SHAPE=0
run $lib VICTIM_SHAPE=2
run $lib VICTIM_SHAPE=3
run $lib VICTIM_SHAPE=0 VICTIM_KS=1
done
echo "#### two streams of one process, this build (tools/diag_concurrent_streams.py)"
SECS=3 timeout 300 python tools/diag_concurrent_streams.py 2>&1 | grep -v "^   diff"
} > gpurun_out/cores_fixed.txt 2>&1
tail -3 gpurun_out/cores_fixed.txt
timeout 1500 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "conv or spade or direct" 2>&1 | tail -3

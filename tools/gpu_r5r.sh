#!/bin/bash
# attention, timing-only: ablm1 = 1/6 of the S product's MFMAs (6 of 36 per key tile), ablm2 = the PV product reduced to 12 of 36 MFMAs
# (one step, two of three cout sub-tiles), ablm3 = both (18 of 72 MFMAs left); everything else (DMA, LDS reads of the kept steps, softmax, split, barriers) unchanged
mkdir -p gpurun_out
R=$PWD
for v in base ablm1 ablm2 ablm3 base ablm3; do
  lib=""; [ $v != base ] && lib=$R/mcvd_pytorch_amd/libmcvd_hip_$v.so
  MCVD_LIB_PATH=$lib timeout 600 python bench.py --no-cpu-baseline --no-f16x2-leg --no-selfcheck > gpurun_out/bench_attn_$v.json 2> gpurun_out/bench_attn_$v.err
  python -c "
import json
d=json.load(open('gpurun_out/bench_attn_$v.json'))
print('$v', d['value'], d['ms_per_step'], {k:(v['launches'],v['ms']) for k,v in d['roofline']['breakdown'].items() if k in ('attention',)})"
done

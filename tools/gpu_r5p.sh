#!/bin/bash
mkdir -p gpurun_out
R=$PWD
for v in base prio base prio; do
  lib=""; [ $v != base ] && lib=$R/mcvd_pytorch_amd/libmcvd_hip_$v.so
  MCVD_LIB_PATH=$lib timeout 600 python bench.py --no-cpu-baseline --no-f16x2-leg > gpurun_out/bench_attn_$v.json 2> gpurun_out/bench_attn_$v.err
  python -c "
import json
d=json.load(open('gpurun_out/bench_attn_$v.json'))
print('$v', d['value'], d['ms_per_step'], d['selfcheck_max_abs'], {k:(v['launches'],v['ms']) for k,v in d['roofline']['breakdown'].items() if k in ('attention',)})"
done

#!/bin/bash
# attention: V operands one group ahead of their products (PF) vs the plain loop (-DMCVD_AH_NOPF build), same box
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$PWD
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1; tail -1 gpurun_out/build.log
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q --tb=short -p no:cacheprovider -x -k "attention or presplit" > gpurun_out/pytest_attn.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_attn.log; tail -3 gpurun_out/pytest_attn.log
for v in nopf pf nopf pf; do
  lib=""; [ $v = nopf ] && lib=$R/mcvd_pytorch_amd/libmcvd_hip_nopf.so
  MCVD_LIB_PATH=$lib timeout 600 python bench.py --no-cpu-baseline --no-f16x2-leg > gpurun_out/bench_attn_$v.json 2> gpurun_out/bench_attn_$v.err
  python -c "
import json
d=json.load(open('gpurun_out/bench_attn_$v.json'))
print('$v', d['value'], d['ms_per_step'], 'selfcheck', d['selfcheck_max_abs'], {k:(v['launches'],v['ms']) for k,v in d['roofline']['breakdown'].items() if k in ('attention','conv1x1')})"
done
for c in smmnist_big5 cityscapes_big; do
for v in nopf pf; do
  lib=""; [ $v = nopf ] && lib=$R/mcvd_pytorch_amd/libmcvd_hip_nopf.so
  MCVD_LIB_PATH=$lib timeout 600 python bench.py --config $c --steps 1 --warmup 1 --no-cpu-baseline --no-f16x2-leg > gpurun_out/bench_attn_${c}_$v.json 2> gpurun_out/bench_attn_${c}_$v.err
  python -c "
import json
d=json.load(open('gpurun_out/bench_attn_${c}_$v.json'))
print('$c $v', d['value'], d['ms_per_step'], 'selfcheck', d['selfcheck_max_abs'], {k:(v['launches'],v['ms']) for k,v in d['roofline']['breakdown'].items() if k in ('attention',)})"
done
done

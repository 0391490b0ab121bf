#!/bin/bash
# lazy softmax reference + score scale folded into Q (attention_h2.cpp): parity, then a same-box A/B against the eager build
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$PWD
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1; tail -1 gpurun_out/build.log
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_big_batch.py -m gpu -q --tb=short -p no:cacheprovider -x -k "attention or presplit or forward_matches or golden or benchmarked or other_baseline or fp32_range" > gpurun_out/pytest_new.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_new.log; tail -6 gpurun_out/pytest_new.log
for v in eager lazy eager lazy; do
  lib=""; [ $v = eager ] && lib=$R/mcvd_pytorch_amd/libmcvd_hip_eager.so
  MCVD_LIB_PATH=$lib timeout 600 python bench.py --no-cpu-baseline --no-f16x2-leg > gpurun_out/bench_attn_$v.json 2> gpurun_out/bench_attn_$v.err
  python -c "
import json
d=json.load(open('gpurun_out/bench_attn_$v.json'))
print('$v', d['value'], d['ms_per_step'], 'selfcheck', d['selfcheck_max_abs'], {k:(v['launches'],v['ms']) for k,v in d['roofline']['breakdown'].items() if k in ('attention','conv1x1')})"
done
for c in kth64_big_ngf128 cityscapes_big; do
for v in eager lazy; do
  lib=""; [ $v = eager ] && lib=$R/mcvd_pytorch_amd/libmcvd_hip_eager.so
  MCVD_LIB_PATH=$lib timeout 600 python bench.py --config $c --steps 1 --warmup 1 --no-cpu-baseline --no-f16x2-leg > gpurun_out/bench_attn_${c}_$v.json 2> gpurun_out/bench_attn_${c}_$v.err
  python -c "
import json
d=json.load(open('gpurun_out/bench_attn_${c}_$v.json'))
print('$c $v', d['value'], d['ms_per_step'], 'selfcheck', d['selfcheck_max_abs'], {k:(v['launches'],v['ms']) for k,v in d['roofline']['breakdown'].items() if k in ('attention',)})"
done
done

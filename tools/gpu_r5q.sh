#!/bin/bash
# attn_h2q_kernel (S product of tile t + 1 under the softmax of tile t inside a wave) vs attn_h2p_kernel (MCVD_ATTN_FORM=0), same box
mkdir -p gpurun_out
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1; tail -1 gpurun_out/build.log
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q --tb=short -p no:cacheprovider -x -k "attention or presplit" > gpurun_out/pytest_attn.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_attn.log; tail -3 gpurun_out/pytest_attn.log
for f in 0 1 0 1; do
  MCVD_ATTN_FORM=$f timeout 600 python bench.py --no-cpu-baseline --no-f16x2-leg > gpurun_out/bench_attnq$f.json 2> gpurun_out/bench_attnq$f.err
  python -c "
import json
d=json.load(open('gpurun_out/bench_attnq$f.json'))
print('form $f', d['value'], d['ms_per_step'], 'selfcheck', d['selfcheck_max_abs'], {k:(v['launches'],v['ms']) for k,v in d['roofline']['breakdown'].items() if k in ('attention','conv1x1')})"
done
for c in smmnist_big5 cityscapes_big; do
for f in 0 1; do
  MCVD_ATTN_FORM=$f timeout 600 python bench.py --config $c --steps 1 --warmup 1 --no-cpu-baseline --no-f16x2-leg > gpurun_out/bench_attnq_${c}_$f.json 2> gpurun_out/bench_attnq_${c}_$f.err
  python -c "
import json
d=json.load(open('gpurun_out/bench_attnq_${c}_$f.json'))
print('$c form $f', d['value'], d['ms_per_step'], 'selfcheck', d['selfcheck_max_abs'], {k:(v['launches'],v['ms']) for k,v in d['roofline']['breakdown'].items() if k in ('attention',)})"
done
done

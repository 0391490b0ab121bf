#!/bin/bash
# attention prefetch depth / workgroup width A/B (MCVD_ATTN_FORM 0: 4 waves x 2 buffers, two workgroups per CU; 1: 8 waves x 3 buffers; 2: 8 waves x 4 buffers)
mkdir -p gpurun_out
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1; tail -1 gpurun_out/build.log
for f in 1 2; do
MCVD_ATTN_FORM=$f timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q --tb=short -p no:cacheprovider -x -k "attention or presplit" > gpurun_out/pytest_attn$f.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_attn$f.log; tail -3 gpurun_out/pytest_attn$f.log
done
for f in 0 1 2 0 1 2; do
  MCVD_ATTN_FORM=$f timeout 600 python bench.py --no-cpu-baseline --no-f16x2-leg > gpurun_out/bench_attnform$f.json 2> gpurun_out/bench_attnform$f.err
  python -c "
import json
d=json.load(open('gpurun_out/bench_attnform$f.json'))
print('form $f', d['value'], d['ms_per_step'], 'selfcheck', d['selfcheck_max_abs'], {k:(v['launches'],v['ms']) for k,v in d['roofline']['breakdown'].items() if k in ('attention','conv1x1')})"
done

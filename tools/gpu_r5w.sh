#!/bin/bash
# flakiness check: the full GPU suite once more on another box + a longer default bench run
mkdir -p gpurun_out
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1; tail -1 gpurun_out/build.log
timeout 2400 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -x > gpurun_out/pytest_gpu2.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu2.log; tail -4 gpurun_out/pytest_gpu2.log
timeout 900 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-f16x2-leg > gpurun_out/bench_steps10.json 2> gpurun_out/bench_steps10.err
python -c "
import json
d=json.load(open('gpurun_out/bench_steps10.json'))
print('steps 10', d['value'], d['ms_per_step'], d['selfcheck_max_abs'], d['valid'], d['per_rank_s'])"

#!/bin/bash
# full GPU suite + smoke + default bench (the round's state)
mkdir -p gpurun_out
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1; tail -1 gpurun_out/build.log
timeout 2400 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider --durations=6 > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log; tail -14 gpurun_out/pytest_gpu.log
timeout 300 python __graft_entry__.py smoke > gpurun_out/smoke.log 2>&1; echo "smoke rc=$?" >> gpurun_out/smoke.log; tail -2 gpurun_out/smoke.log
timeout 900 python bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err; echo "bench rc=$?" >> gpurun_out/bench_default.err; python - <<'PY'
import json
d=json.load(open('gpurun_out/bench_default.json'))
print(d['value'], d['ms_per_step'], d['roofline']['frac'], 'selfcheck', d['selfcheck_max_abs'], {k:(v['launches'],v['ms']) for k,v in d['roofline']['breakdown'].items()}, 'f16x2', d.get('f16x2_leg',{}).get('value'), 'cpu', d['cpu_baseline'].get('value'), d['cpu_baseline'].get('speedup'))
PY
tail -2 gpurun_out/bench_default.err

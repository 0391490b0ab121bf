#!/bin/bash
# closing run of a round: full GPU suite, smoke, the default bench line, every other BASELINE config under its committed table
mkdir -p gpurun_out
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1; tail -1 gpurun_out/build.log
timeout 2400 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider --durations=5 > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log; tail -12 gpurun_out/pytest_gpu.log
timeout 300 python __graft_entry__.py smoke > gpurun_out/smoke.log 2>&1; echo "smoke rc=$?" >> gpurun_out/smoke.log; tail -2 gpurun_out/smoke.log
timeout 900 python bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err; echo "bench rc=$?" >> gpurun_out/bench_default.err; python - <<'PY'
import json
d=json.load(open('gpurun_out/bench_default.json'))
print(d['value'], d['ms_per_step'], d['roofline']['frac'], 'selfcheck', d['selfcheck_max_abs'], {k:(v['launches'],v['ms']) for k,v in d['roofline']['breakdown'].items()}, 'f16x2', d.get('f16x2_leg',{}).get('value'), 'cpu', d['cpu_baseline'].get('value'), d['cpu_baseline'].get('speedup'))
PY
for c in smmnist_big5 kth64_big_ngf128 bair_big_spade cityscapes_big cityscapes_big_variant; do
  timeout 900 python bench.py --config $c --steps 1 --warmup 1 --no-cpu-baseline > gpurun_out/bench_other_$c.json 2> gpurun_out/bench_other_$c.err
  python -c "
import json
d=json.load(open('gpurun_out/bench_other_$c.json'))
print('$c', d['value'], d['ms_per_step'], d['roofline']['frac'], 'selfcheck', d['selfcheck_max_abs'], 'f16x2 leg', d.get('f16x2_leg', {}).get('value'), d['config']['kernel_table'])"
done

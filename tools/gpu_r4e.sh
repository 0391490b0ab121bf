#!/bin/bash
# attention XCD mapping, lazy fp16 weight forms, two-rank bench test: the affected tests + the bench
mkdir -p gpurun_out
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1; tail -1 gpurun_out/build.log
timeout 1200 python -m pytest tests/test_gpu_parity.py -m gpu -q --tb=short -p no:cacheprovider -x -k "attention or f16x2 or two_ranks or bench_kernel_table or forward_vs_reference_golden" > gpurun_out/pytest_e.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_e.log; tail -6 gpurun_out/pytest_e.log
timeout 900 python bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/bench_r4e.json 2> gpurun_out/bench_r4e.err; echo "bench rc=$?" >> gpurun_out/bench_r4e.err; python - <<'PY'
import json
d=json.load(open('gpurun_out/bench_r4e.json'))
print(d['value'], d['ms_per_step'], d['roofline']['frac'], 'selfcheck', d['selfcheck_max_abs'], {k:(v['launches'],v['ms']) for k,v in d['roofline']['breakdown'].items()}, 'f16x2', d.get('f16x2_leg',{}).get('value'))
PY
tail -2 gpurun_out/bench_r4e.err

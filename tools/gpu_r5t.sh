#!/bin/bash
# full GPU suite + smoke + default bench + kernel stats of the new kernels
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$PWD
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1; tail -1 gpurun_out/build.log
timeout 2400 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider --durations=5 > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log; tail -9 gpurun_out/pytest_gpu.log
timeout 300 python __graft_entry__.py smoke > gpurun_out/smoke.log 2>&1; echo "smoke rc=$?" >> gpurun_out/smoke.log; tail -2 gpurun_out/smoke.log
timeout 900 python bench.py --no-cpu-baseline > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/bench_default.json')); r=d['roofline']
print('bench', d['value'], d['ms_per_step'], r['frac'], 'valid', d.get('valid'), 'selfcheck', d['selfcheck_max_abs'], {k:(v['launches'],v['ms'],v['gbs']) for k,v in r['breakdown'].items()}, 'f16x2', d.get('f16x2_leg',{}).get('value'), r['conv3x3_families'])
PY
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_new -o bench -- python $R/bench.py --steps 1 --warmup 0 --subsample 20 --no-cpu-baseline --no-f16x2-leg --no-selfcheck > $R/gpurun_out/prof_new.json 2> $R/gpurun_out/prof_new.err
cd $R
python - <<'PY'
import csv, glob
for f in glob.glob('gpurun_out/prof_new/**/*kernel_stats.csv', recursive=True):
    rows=list(csv.DictReader(open(f)))
    tot=sum(float(r['TotalDurationNs']) for r in rows)
    for r in sorted(rows, key=lambda r:-float(r['TotalDurationNs']))[:26]:
        print(f"{r['Name'][:76]:76s} {int(r['Calls']):6d} {float(r['TotalDurationNs'])/1e6:9.2f} ms {float(r['TotalDurationNs'])/int(r['Calls'])/1e3:8.1f} us {100*float(r['TotalDurationNs'])/tot:5.2f}%")
PY

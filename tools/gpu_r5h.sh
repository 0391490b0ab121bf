#!/bin/bash
# FIR x2 resamplers through the LDS: parity, A/B of the bench line, kernel durations under rocprofv3
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$PWD
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1; tail -1 gpurun_out/build.log
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q --tb=short -p no:cacheprovider -k "fir or forward_matches or other_baseline" > gpurun_out/pytest_new.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_new.log; tail -8 gpurun_out/pytest_new.log
for f in 1 0 1 0; do
  MCVD_BENCH_OPTS=fir_form=$f timeout 600 python bench.py --no-cpu-baseline --no-f16x2-leg > gpurun_out/bench_fir$f.json 2> gpurun_out/bench_fir$f.err
  python -c "
import json
d=json.load(open('gpurun_out/bench_fir$f.json'))
print('fir_form $f', d['value'], d['ms_per_step'], 'selfcheck', d['selfcheck_max_abs'], {k:(v['launches'],v['ms'],v.get('gbs')) for k,v in d['roofline']['breakdown'].items() if k in ('fir2','conv1x1','gn_coef')})"
done
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_fir -o bench -- python $R/bench.py --steps 1 --warmup 0 --subsample 20 --no-cpu-baseline --no-f16x2-leg --no-selfcheck > $R/gpurun_out/prof_fir.json 2> $R/gpurun_out/prof_fir.err
cd $R
python - <<'PY'
import csv, glob
for f in glob.glob('gpurun_out/prof_fir/**/*kernel_stats.csv', recursive=True):
    rows=list(csv.DictReader(open(f)))
    tot=sum(float(r['TotalDurationNs']) for r in rows)
    for r in sorted(rows, key=lambda r:-float(r['TotalDurationNs'])):
        if 'fir' in r['Name'] or 'conv_mfma' in r['Name'] or 'wino3p_kernel<1' in r['Name']:
            print(f"{r['Name'][:72]:72s} {int(r['Calls']):6d} {float(r['TotalDurationNs'])/1e6:9.2f} ms {float(r['TotalDurationNs'])/int(r['Calls'])/1e3:8.1f} us min {float(r['MinNs'])/1e3:.1f} max {float(r['MaxNs'])/1e3:.1f} {100*float(r['TotalDurationNs'])/tot:5.2f}%")
PY

#!/bin/bash
# second closing run of round 5 (after the FIR strip kernels, the lazy softmax and the GEMM forms): full GPU suite, smoke, the default bench line (+ traffic measured in the run), every other BASELINE config under its
# committed table, rocprofv3 stats + PMC passes of the final code
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$PWD
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1; tail -1 gpurun_out/build.log
timeout 2400 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider --durations=5 > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log; tail -12 gpurun_out/pytest_gpu.log
timeout 300 python __graft_entry__.py smoke > gpurun_out/smoke.log 2>&1; echo "smoke rc=$?" >> gpurun_out/smoke.log; tail -2 gpurun_out/smoke.log
timeout 900 python bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err; echo "bench rc=$?" >> gpurun_out/bench_default.err
timeout 900 python bench.py --pmc-traffic --no-cpu-baseline > gpurun_out/bench_pmc_traffic.json 2> gpurun_out/bench_pmc_traffic.err; echo "bench rc=$?" >> gpurun_out/bench_pmc_traffic.err
python - <<'PY'
import json
for f in ('bench_default', 'bench_pmc_traffic'):
    d=json.load(open('gpurun_out/%s.json' % f))
    r=d['roofline']
    print(f, d['value'], d['ms_per_step'], r['frac'], 'valid', d.get('valid'), 'selfcheck', d['selfcheck_max_abs'], 'traffic', r.get('traffic'), r.get('traffic_measured_in_run'), r.get('traffic_vs_algorithmic'), 'fwd_vs_step', r.get('forward_events_vs_step'))
    print('   ', {k:(v['launches'],v['ms']) for k,v in r['breakdown'].items()}, 'f16x2', d.get('f16x2_leg',{}).get('value'), 'cpu', {k:v for k,v in d.get('cpu_baseline',{}).items() if k in ('value','cores','speedup','port_vs_reference','reference_estimate')})
PY
for c in smmnist_big5 kth64_big_ngf128 bair_big_spade cityscapes_big cityscapes_big_variant; do
  timeout 900 python bench.py --config $c --steps 1 --warmup 1 --no-cpu-baseline > gpurun_out/bench_other_$c.json 2> gpurun_out/bench_other_$c.err
  python -c "
import json
d=json.load(open('gpurun_out/bench_other_$c.json'))
print('$c', d['value'], d['ms_per_step'], d['roofline']['frac'], 'selfcheck', d['selfcheck_max_abs'], 'f16x2 leg', d.get('f16x2_leg', {}).get('value'), d['config']['kernel_table'])"
done
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof -o bench -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-f16x2-leg --no-selfcheck > $R/gpurun_out/prof_bench.json 2> $R/gpurun_out/prof_bench.err
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES --output-format csv -d $R/gpurun_out/pmc_mfma -o bench -- python $R/bench.py --steps 1 --warmup 0 --subsample 5 --no-cpu-baseline --no-f16x2-leg --no-selfcheck --graph 0 > $R/gpurun_out/pmc_mfma.json 2> $R/gpurun_out/pmc_mfma.err
cd $R
python tools/summarize_prof.py > gpurun_out/prof_summary.txt 2>&1; head -42 gpurun_out/prof_summary.txt
python tools/summarize_prof.py mfma > gpurun_out/pmc_mfma_summary.txt 2>&1; head -24 gpurun_out/pmc_mfma_summary.txt
cp gpurun_out/prof/*/*kernel_stats.csv gpurun_out/prof_kernel_stats.csv 2>/dev/null || find gpurun_out/prof -name "*kernel_stats.csv" -exec cp {} gpurun_out/prof_kernel_stats.csv \;
find gpurun_out/prof gpurun_out/pmc_mfma -name "*.csv" -size +20M -delete
# config 4 under rocprofv3 (SPADE path)
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof4 -o bench -- python $R/bench.py --config bair_big_spade --steps 1 --warmup 0 --subsample 100 --no-cpu-baseline --no-f16x2-leg --no-selfcheck > $R/gpurun_out/prof4_bench.json 2> $R/gpurun_out/prof4_bench.err
cd $R
python - <<'PY'
import csv, glob
from collections import defaultdict
for f in glob.glob('gpurun_out/prof4/**/*kernel_stats.csv', recursive=True):
    rows=list(csv.DictReader(open(f)))
    tot=sum(float(r['TotalDurationNs']) for r in rows)
    out=open('gpurun_out/prof4_summary.txt','w')
    for r in sorted(rows, key=lambda r:-float(r['TotalDurationNs']))[:30]:
        line=f"{r['Name'][:72]:72s} {int(r['Calls']):6d} {float(r['TotalDurationNs'])/1e6:9.2f} ms {float(r['TotalDurationNs'])/int(r['Calls'])/1e3:8.1f} us {100*float(r['TotalDurationNs'])/tot:5.2f}%"
        print(line); out.write(line+'\n')
    out.write(f"total {tot/1e6:.1f} ms\n")
PY

#!/bin/bash
# rocprofv3 kernel stats of one sampler call (subsample 100) of another BASELINE config: CFG=<name> bash tools/gpu_prof_cfg.sh
export CFG=${CFG:-bair_big_spade}
mkdir -p gpurun_out; export TMPDIR=/tmp; R=$(pwd)
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$CFG -o bench -- python $R/bench.py --config $CFG --steps 1 --warmup 0 --subsample 100 --no-cpu-baseline --no-f16x2-leg --no-selfcheck > $R/gpurun_out/prof_${CFG}_bench.json 2> $R/gpurun_out/prof_${CFG}_bench.err
cd $R
find gpurun_out/prof_$CFG -name "*.csv" -size +20M -delete
python - <<'PY'
import csv,glob,os
f=glob.glob('gpurun_out/prof_'+os.environ.get('CFG','bair_big_spade')+'/**/bench_kernel_stats.csv',recursive=True)[0]
rows=list(csv.DictReader(open(f)))
tot=sum(float(r['TotalDurationNs']) for r in rows)
for r in sorted(rows,key=lambda r:-float(r['TotalDurationNs']))[:22]:
    print(f"{r['Name'][:60]:60s} {int(r['Calls']):7d} {float(r['TotalDurationNs'])/1e6:9.2f} ms {float(r['AverageNs'])/1e3:8.1f} us {100*float(r['TotalDurationNs'])/tot:5.2f}%")
PY

#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$PWD
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_big_batch.py -m gpu -q --tb=short -p no:cacheprovider -k "gemm_forms or bench_kernel_table or benchmarked" > gpurun_out/pytest_new.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_new.log; tail -4 gpurun_out/pytest_new.log
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_new -o bench -- python $R/bench.py --steps 1 --warmup 0 --subsample 20 --no-cpu-baseline --no-f16x2-leg --no-selfcheck > $R/gpurun_out/prof_new.json 2> $R/gpurun_out/prof_new.err
cd $R
python - <<'PY'
import csv, glob
for f in glob.glob('gpurun_out/prof_new/**/*kernel_stats.csv', recursive=True):
    rows=list(csv.DictReader(open(f)))
    for r in rows:
        if any(k in r['Name'] for k in ('im2col','taps_shift','h2_kernel<3, 1, 0','h2_kernel<3, 2, 2','fir_')):
            print(f"{r['Name'][:76]:76s} {int(r['Calls']):6d} {float(r['TotalDurationNs'])/int(r['Calls'])/1e3:8.1f} us")
PY

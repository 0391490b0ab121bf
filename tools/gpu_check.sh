#!/bin/bash
# One gpurun call.  Usage (from the repo root on the GPU box): bash tools/gpu_check.sh MODE...
# Modes (any number, run in order): test | smoke | bench | bench2 (N=2 gloo plumbing on one GPU) | graphab | others | prof | pmc_mfma | pmc_sq | pmc_l2
mkdir -p gpurun_out
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
rocm-smi --showproductname > gpurun_out/rocm_smi.log 2>&1
nproc > gpurun_out/nproc.log; lscpu | head -20 >> gpurun_out/nproc.log
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1
echo "build rc=$?" >> gpurun_out/build.log
# (bench.py pins the committed kernel tables profiles/tune_<config>_B<batch>_<arithmetic>.json by default: no autotune launches in the profiles)
for MODE in "$@"; do
case $MODE in
smoke)
  timeout 300 python __graft_entry__.py smoke > gpurun_out/smoke.log 2>&1; echo "smoke rc=$?" >> gpurun_out/smoke.log; tail -3 gpurun_out/smoke.log;;
test)
  timeout 2400 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -x > gpurun_out/pytest_gpu.log 2>&1
  echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log; tail -30 gpurun_out/pytest_gpu.log;;
testall)
  timeout 2400 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider > gpurun_out/pytest_gpu.log 2>&1
  echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log; tail -40 gpurun_out/pytest_gpu.log;;
bench)
  timeout 900 python bench.py --steps 2 --warmup 1 > gpurun_out/bench.json 2> gpurun_out/bench.err
  echo "bench rc=$?" >> gpurun_out/bench.err; cat gpurun_out/bench.json | cut -c1-1500; tail -5 gpurun_out/bench.err;;
graphab)
  for g in 0 1; do
    timeout 600 python bench.py --steps 2 --warmup 1 --graph $g --no-cpu-baseline > gpurun_out/bench_graph$g.json 2> gpurun_out/bench_graph$g.err
    python -c "import json;d=json.load(open('gpurun_out/bench_graph$g.json'));print('graph$g', d['value'], d['ms_per_step'])"
  done;;
gnab)
  for g in 0 1; do
    MCVD_GN_STATS=$g timeout 600 python bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/bench_gn$g.json 2> gpurun_out/bench_gn$g.err
    python -c "import json;d=json.load(open('gpurun_out/bench_gn$g.json'));print('gn_stats$g', d['value'], d['ms_per_step'], d['roofline']['frac'], {k:(v['launches'],v['ms']) for k,v in d['roofline']['breakdown'].items()})"
  done;;
bench2)
  MCVD_DIST_BACKEND=gloo timeout 900 python bench.py --gpus 2 --steps 1 --warmup 1 --batch 8 --no-cpu-baseline > gpurun_out/bench_2proc.json 2> gpurun_out/bench_2proc.err
  echo "bench2 rc=$?" >> gpurun_out/bench_2proc.err; cat gpurun_out/bench_2proc.json | cut -c1-600; tail -3 gpurun_out/bench_2proc.err;;
others)
  for c in smmnist_big5 kth64_big_ngf128 bair_big_spade cityscapes_big cityscapes_big_variant; do
    for g in ${GRAPHS:-1}; do
      timeout 900 python bench.py --config $c --steps 1 --warmup 1 --graph $g --no-cpu-baseline > gpurun_out/bench_other_${c}_g$g.json 2> gpurun_out/bench_other_${c}_g$g.err
      python -c "import json;d=json.load(open('gpurun_out/bench_other_${c}_g$g.json'));print('$c graph$g', d['value'], d['ms_per_step'], d['roofline']['frac'], 'f16x2 leg', d.get('f16x2_leg', {}).get('value'))"
    done
  done;;
prof)
  cd /tmp
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof -o bench -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-f16x2-leg --no-selfcheck > $R/gpurun_out/prof_bench.json 2> $R/gpurun_out/prof_bench.err
  # HBM-side PMC counters, each in its own run (kernel-trace only); the kernel table comes from the cache: no autotune launches
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/pmc_fetch -o bench -- python $R/bench.py --steps 1 --warmup 0 --subsample 5 --no-cpu-baseline --no-f16x2-leg --no-selfcheck --graph 0 > $R/gpurun_out/pmc_fetch.json 2> $R/gpurun_out/pmc_fetch.err
  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/pmc_write -o bench -- python $R/bench.py --steps 1 --warmup 0 --subsample 5 --no-cpu-baseline --no-f16x2-leg --no-selfcheck --graph 0 > $R/gpurun_out/pmc_write.json 2> $R/gpurun_out/pmc_write.err
  cd $R; python tools/summarize_prof.py > gpurun_out/prof_summary.txt 2>&1; head -40 gpurun_out/prof_summary.txt
  python tools/summarize_prof.py traffic gpurun_out/conv3x3_traffic.json > gpurun_out/traffic.log 2>&1; tail -3 gpurun_out/traffic.log
  find gpurun_out/prof gpurun_out/pmc_fetch gpurun_out/pmc_write -name "*.csv" -size +20M -delete;;
pmc_mfma)
  cd /tmp
  rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES --output-format csv -d $R/gpurun_out/pmc_mfma -o bench -- python $R/bench.py --steps 1 --warmup 0 --subsample 5 --no-cpu-baseline --no-f16x2-leg --no-selfcheck --graph 0 > $R/gpurun_out/pmc_mfma.json 2> $R/gpurun_out/pmc_mfma.err
  cd $R; python tools/summarize_prof.py mfma > gpurun_out/pmc_mfma_summary.txt 2>&1; head -30 gpurun_out/pmc_mfma_summary.txt
  find gpurun_out/pmc_mfma -name "*.csv" -size +20M -delete;;
pmc_l2)
  # L2 side of the dominant kernel: hit rate and request counts (one pass per TCC / TCP group)
  cd /tmp
  rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum --output-format csv -d $R/gpurun_out/pmc_l2a -o bench -- python $R/bench.py --steps 1 --warmup 0 --subsample 5 --no-cpu-baseline --no-f16x2-leg --no-selfcheck --graph 0 > $R/gpurun_out/pmc_l2a.json 2> $R/gpurun_out/pmc_l2a.err
  rocprofv3 --kernel-trace --pmc TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_LATENCY_sum --output-format csv -d $R/gpurun_out/pmc_l2b -o bench -- python $R/bench.py --steps 1 --warmup 0 --subsample 5 --no-cpu-baseline --no-f16x2-leg --no-selfcheck --graph 0 > $R/gpurun_out/pmc_l2b.json 2> $R/gpurun_out/pmc_l2b.err
  cd $R; python tools/summarize_prof.py l2 > gpurun_out/pmc_l2_summary.txt 2>&1; head -40 gpurun_out/pmc_l2_summary.txt; tail -3 gpurun_out/pmc_l2a.err gpurun_out/pmc_l2b.err
  find gpurun_out/pmc_l2a gpurun_out/pmc_l2b -name "*.csv" -size +20M -delete;;
pmc_sq)
  # wave-state split of the kernels: parked (s_waitcnt / barrier) vs issue-stalled (pipe busy) vs issuing; LDS conflicts
  cd /tmp
  rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $R/gpurun_out/pmc_sq -o bench -- python $R/bench.py --steps 1 --warmup 0 --subsample 5 --no-cpu-baseline --no-f16x2-leg --no-selfcheck --graph 0 > $R/gpurun_out/pmc_sq.json 2> $R/gpurun_out/pmc_sq.err
  cd $R; python tools/summarize_prof.py sq > gpurun_out/pmc_sq_summary.txt 2>&1; head -30 gpurun_out/pmc_sq_summary.txt
  find gpurun_out/pmc_sq -name "*.csv" -size +20M -delete;;
*) echo "unknown mode $MODE";;
esac
cd $R
done

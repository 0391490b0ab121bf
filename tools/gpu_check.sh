#!/bin/bash
# One gpurun call: build check, GPU parity tests, smoke, short bench, (optional) rocprofv3 kernel trace.
# Usage (from the repo root on the GPU box): bash tools/gpu_check.sh [quick|full|prof]
MODE=${1:-full}
mkdir -p gpurun_out
export TMPDIR=/tmp
rocm-smi --showproductname > gpurun_out/rocm_smi.log 2>&1
nproc > gpurun_out/nproc.log; lscpu | head -20 >> gpurun_out/nproc.log
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1
echo "build rc=$?" >> gpurun_out/build.log
if [ "$MODE" != "prof" ] && [ "$MODE" != "pmc3" ]; then
  timeout 300 python __graft_entry__.py smoke > gpurun_out/smoke.log 2>&1
  echo "smoke rc=$?" >> gpurun_out/smoke.log
  tail -3 gpurun_out/smoke.log
  timeout 1500 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider > gpurun_out/pytest_gpu.log 2>&1
  echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log
  tail -40 gpurun_out/pytest_gpu.log
fi
if [ "$MODE" != "quick" ] && [ "$MODE" != "prof" ] && [ "$MODE" != "pmc3" ]; then
  timeout 900 python bench.py --steps 1 --warmup 1 > gpurun_out/bench.json 2> gpurun_out/bench.err
  echo "bench rc=$?" >> gpurun_out/bench.err
  cat gpurun_out/bench.json; tail -5 gpurun_out/bench.err
fi
if [ "$MODE" = "pmc2" ]; then
  R=$GRAFT_REPO_ROOT
  cd /tmp
  rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES --output-format csv -d $R/gpurun_out/pmc_mfma -o bench -- python $R/bench.py --steps 1 --warmup 0 --subsample 5 --no-cpu-baseline > $R/gpurun_out/pmc_mfma.json 2> $R/gpurun_out/pmc_mfma.err
  cd $R; python tools/summarize_prof.py mfma > gpurun_out/pmc_mfma_summary.txt 2>&1; head -30 gpurun_out/pmc_mfma_summary.txt
fi
if [ "$MODE" = "pmc3" ]; then
  # wave-state split of the kernels: parked (s_waitcnt / barrier) vs issue-stalled (pipe busy) vs issuing; LDS conflicts
  R=$GRAFT_REPO_ROOT
  cd /tmp
  rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $R/gpurun_out/pmc_sq -o bench -- python $R/bench.py --steps 1 --warmup 0 --subsample 5 --no-cpu-baseline > $R/gpurun_out/pmc_sq.json 2> $R/gpurun_out/pmc_sq.err
  cd $R; python tools/summarize_prof.py sq > gpurun_out/pmc_sq_summary.txt 2>&1; head -30 gpurun_out/pmc_sq_summary.txt
fi
if [ "$MODE" = "prof" ]; then
  R=$GRAFT_REPO_ROOT
  cd /tmp
  # pass 1: kernel trace + stats of the default bench command (same workload as the bench line)
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof -o bench -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline > $R/gpurun_out/prof_bench.json 2> $R/gpurun_out/prof_bench.err
  # passes 2,3: HBM-side PMC counters, each in its own run (kernel-trace only), on a shortened sampler (6 forwards)
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/pmc_fetch -o bench -- python $R/bench.py --steps 1 --warmup 0 --subsample 5 --no-cpu-baseline > $R/gpurun_out/pmc_fetch.json 2> $R/gpurun_out/pmc_fetch.err
  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/pmc_write -o bench -- python $R/bench.py --steps 1 --warmup 0 --subsample 5 --no-cpu-baseline > $R/gpurun_out/pmc_write.json 2> $R/gpurun_out/pmc_write.err
  cd $R; python tools/summarize_prof.py > gpurun_out/prof_summary.txt 2>&1; head -40 gpurun_out/prof_summary.txt
  # keep the merged payload small: raw per-dispatch CSVs can be large
  find gpurun_out/prof gpurun_out/pmc_fetch gpurun_out/pmc_write -name "*.csv" -size +20M -delete
fi

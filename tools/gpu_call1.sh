#!/bin/bash
# session re-entry check: full GPU suite at HEAD, per-layer kernel table with / without the bf16x3 kernel, bench A/B
mkdir -p gpurun_out
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1; echo "build rc=$?"
( time timeout 1500 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider --durations=15 > gpurun_out/pytest_gpu.log 2>&1 ); echo "pytest rc=$?"; tail -40 gpurun_out/pytest_gpu.log
timeout 300 python tests/gpu_diag.py convops > gpurun_out/convops.log 2>&1; cp gpurun_out/diag_convops.txt gpurun_out/diag_convops_bf16x3.txt; tail -3 gpurun_out/diag_convops_bf16x3.txt
MCVD_BF16X3=0 timeout 300 python tests/gpu_diag.py convops > gpurun_out/convops0.log 2>&1; cp gpurun_out/diag_convops.txt gpurun_out/diag_convops_f32.txt; tail -3 gpurun_out/diag_convops_f32.txt
MCVD_WEXP_ONLY=0 timeout 300 python tests/gpu_diag.py w3exp > gpurun_out/w3exp.log 2>&1; cut -c1-230 gpurun_out/diag_w3exp.txt; tail -3 gpurun_out/w3exp.log
timeout 600 python bench.py --steps 2 --warmup 1 > gpurun_out/bench_bf16x3.json 2> gpurun_out/bench_bf16x3.err; echo "bench rc=$?"; cut -c1-1500 gpurun_out/bench_bf16x3.json
MCVD_BF16X3=0 timeout 600 python bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/bench_f32.json 2> gpurun_out/bench_f32.err; echo "bench0 rc=$?"; cut -c1-600 gpurun_out/bench_f32.json

#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1; tail -1 gpurun_out/build.log
timeout 2400 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider --durations=15 > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log; tail -45 gpurun_out/pytest_gpu.log
timeout 300 python __graft_entry__.py smoke > gpurun_out/smoke.log 2>&1; echo "smoke rc=$?" >> gpurun_out/smoke.log; tail -2 gpurun_out/smoke.log
timeout 300 python tests/gpu_diag.py hostloop > gpurun_out/hostloop.log 2>&1; cat gpurun_out/diag_hostloop.txt; tail -2 gpurun_out/hostloop.log

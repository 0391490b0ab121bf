#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1; tail -1 gpurun_out/build.log
timeout 600 python -m pytest tests/test_gpu_parity.py -q -p no:cacheprovider --tb=short -k "test_f16x2_overflow or test_bench_kernel_table or test_f16x2_option" > gpurun_out/pytest_f.log 2>&1; echo "pytest_f rc=$?" >> gpurun_out/pytest_f.log; tail -25 gpurun_out/pytest_f.log

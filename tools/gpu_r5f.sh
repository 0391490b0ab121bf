#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1; tail -1 gpurun_out/build.log
timeout 300 tools/bin/ubench_wino64 > gpurun_out/ubench_wino64.log 2>&1; cat gpurun_out/ubench_wino64.log
timeout 300 tools/bin/ubench_wino64 > gpurun_out/ubench_wino64_b.log 2>&1; grep -E "everything|matrix|staging" gpurun_out/ubench_wino64_b.log | cut -c1-150
timeout 300 python tools/diag_small_cout.py > gpurun_out/diag_small_cout.log 2>&1; grep -v amdgpu.ids gpurun_out/diag_small_cout.log | tail -6
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q --tb=short -p no:cacheprovider -k "small_cout or bit_deterministic or two_streams" > gpurun_out/pytest_new.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_new.log; tail -6 gpurun_out/pytest_new.log

#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1; tail -1 gpurun_out/build.log
for i in 1 2 3; do timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q --tb=short -p no:cacheprovider -x -k "two_ranks" 2>&1 | tail -2; done
timeout 600 python -m pytest tests/test_gpu_wino3p.py tests/test_gpu_parity.py -m gpu -q --tb=short -p no:cacheprovider -x -k "selftest or attention or two_contexts" 2>&1 | tail -3
timeout 600 python tests/gpu_diag.py w3ptl > gpurun_out/w3ptl.log 2>&1; cat gpurun_out/diag_w3ptl.txt | cut -c1-420

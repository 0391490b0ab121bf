#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1; tail -1 gpurun_out/build.log
timeout 300 python tests/gpu_diag.py w2htl > gpurun_out/w2htl.log 2>&1; cat gpurun_out/diag_w2htl.txt | cut -c1-420; tail -3 gpurun_out/w2htl.log

#!/bin/bash
# round 3, step q: GroupNorm coefficients computed by the consuming conv (gn_inline)
mkdir -p gpurun_out
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1; tail -1 gpurun_out/build.log
timeout 1500 python -m pytest tests/test_gpu_parity.py -m gpu -q --tb=short -p no:cacheprovider -x -k "gn_ or forward or sampler or determin or table or video or fpndm" > gpurun_out/pytest_q.log 2>&1; tail -12 gpurun_out/pytest_q.log
for gi in 1 0 1 0; do
MCVD_GN_INLINE=$gi timeout 900 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-f16x2-leg > gpurun_out/bench_q$gi.json 2> gpurun_out/bench_q.err; python - <<PY
import json
d=json.load(open('gpurun_out/bench_q$gi.json'))
print('gn_inline $gi', d['value'], d['ms_per_step'], d['roofline']['frac'], {k:(v['launches'],v['ms']) for k,v in d['roofline']['breakdown'].items()})
PY
done
tail -2 gpurun_out/bench_q.err

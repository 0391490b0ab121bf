#!/bin/bash
# gn_inline A/B on the small-batch configs
mkdir -p gpurun_out
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1; tail -1 gpurun_out/build.log
for c in cityscapes_big kth64_big_ngf128 bair_big_spade; do
for gi in 1 0 1 0; do
MCVD_GN_INLINE=$gi timeout 900 python bench.py --config $c --steps 1 --warmup 1 --no-cpu-baseline --no-f16x2-leg > gpurun_out/bench_r.json 2> gpurun_out/bench_r.err; python - <<PY
import json
d=json.load(open('gpurun_out/bench_r.json'))
print('$c gn_inline $gi', d['value'], d['ms_per_step'])
PY
done
done
tail -2 gpurun_out/bench_r.err

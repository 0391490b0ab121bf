#!/bin/bash
# round 5, call B: new fused paths (SPADE norm, presplit attention, device fence) + bench A/B
mkdir -p gpurun_out
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1; tail -1 gpurun_out/build.log
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q --tb=short -p no:cacheprovider -x -k "presplit or spade or two_streams or second_process or attention or forward_vs_reference or foreign or two_contexts" > gpurun_out/pytest_new.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_new.log; tail -25 gpurun_out/pytest_new.log
for ps in 1 0; do
  MCVD_BENCH_OPTS=attn_presplit=$ps timeout 600 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-f16x2-leg > gpurun_out/bench_ps$ps.json 2> gpurun_out/bench_ps$ps.err
  python - <<PY
import json
d=json.load(open('gpurun_out/bench_ps$ps.json'))
print('presplit=$ps', d['value'], d['ms_per_step'], 'selfcheck', d['selfcheck_max_abs'], {k:(v['launches'],v['ms']) for k,v in d['roofline']['breakdown'].items()})
PY
done
for nf in 1 0; do
  MCVD_BENCH_OPTS=spade_norm_fuse=$nf,spade_fuse_auto=0 timeout 600 python bench.py --config bair_big_spade --steps 1 --warmup 1 --subsample 200 --no-cpu-baseline --no-f16x2-leg > gpurun_out/bench_sn$nf.json 2> gpurun_out/bench_sn$nf.err
  python - <<PY
import json
d=json.load(open('gpurun_out/bench_sn$nf.json'))
print('spade_norm_fuse=$nf', d['value'], d['ms_per_step'], 'selfcheck', d['selfcheck_max_abs'], {k:(v['launches'],v['ms']) for k,v in d['roofline']['breakdown'].items()})
PY
done
# re-tune cfg 4 with the per-layer fused-loader candidates offered
timeout 900 python bench.py --config bair_big_spade --steps 1 --warmup 1 --subsample 200 --no-cpu-baseline --no-f16x2-leg --no-tune-file --save-tuning gpurun_out/tune_new > gpurun_out/bench_sn_tuned.json 2> gpurun_out/bench_sn_tuned.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/bench_sn_tuned.json'))
print('re-tuned cfg4', d['value'], d['ms_per_step'], 'selfcheck', d['selfcheck_max_abs'], {k:(v['launches'],v['ms']) for k,v in d['roofline']['breakdown'].items()})
t=json.load(open('gpurun_out/tune_new/tune_bair_big_spade_B16_bf16x3.json'))['16']
from collections import Counter
print('shapes', Counter(s for s,_ in t))
PY

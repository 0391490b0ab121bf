#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1; tail -1 gpurun_out/build.log
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q --tb=short -p no:cacheprovider -k "presplit or two_streams or small_cout or epilogue_group_norm or statistics_paths or forward_vs_reference or other_baseline" > gpurun_out/pytest_new.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_new.log; tail -12 gpurun_out/pytest_new.log
timeout 300 python tools/diag_small_cout.py > gpurun_out/diag_small_cout.log 2>&1; grep -v amdgpu.ids gpurun_out/diag_small_cout.log | tail -8
for c in kth64_big_ngf128 cityscapes_big; do
for ps in 1 0; do
  MCVD_BENCH_OPTS=attn_presplit=$ps timeout 600 python bench.py --config $c --steps 1 --warmup 1 --no-cpu-baseline --no-f16x2-leg > gpurun_out/bench_${c}_ps$ps.json 2> gpurun_out/bench_${c}_ps$ps.err
  python - <<PY
import json
d=json.load(open('gpurun_out/bench_${c}_ps$ps.json'))
print('$c presplit=$ps', d['value'], d['ms_per_step'], 'selfcheck', d['selfcheck_max_abs'], {k:(v['launches'],v['ms']) for k,v in d['roofline']['breakdown'].items() if k in ('conv1x1','attention')})
PY
done
done

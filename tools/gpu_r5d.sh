#!/bin/bash
# round 5, call D: full suite with the new kernels, then re-tune every BASELINE config (shape id 21, stem statistics, pre-split attention)
mkdir -p gpurun_out/tune
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1; tail -1 gpurun_out/build.log
timeout 1500 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider --durations=5 > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log; tail -15 gpurun_out/pytest_gpu.log
timeout 600 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-f16x2-leg > gpurun_out/bench_oldtable.json 2> gpurun_out/bench_oldtable.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/bench_oldtable.json'))
print('old table', d['value'], d['ms_per_step'], 'selfcheck', d['selfcheck_max_abs'], {k:(v['launches'],v['ms']) for k,v in d['roofline']['breakdown'].items()})
PY
for c in smmnist_big5_ngf96 smmnist_big5 kth64_big_ngf128 bair_big_spade cityscapes_big cityscapes_big_variant; do
  ss=""; [ $c = bair_big_spade ] && ss="--subsample 200"
  timeout 900 python bench.py --config $c --steps 1 --warmup 1 $ss --no-cpu-baseline --no-tune-file --save-tuning gpurun_out/tune > gpurun_out/bench_tune_$c.json 2> gpurun_out/bench_tune_$c.err
  python -c "
import json
d=json.load(open('gpurun_out/bench_tune_$c.json'))
print('$c', d['value'], d['ms_per_step'], d['roofline']['frac'], 'selfcheck', d['selfcheck_max_abs'], 'f16x2 leg', d.get('f16x2_leg', {}).get('value'), {k:(v['launches'],v['ms']) for k,v in d['roofline']['breakdown'].items()})"
done
ls gpurun_out/tune

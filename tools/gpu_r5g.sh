#!/bin/bash
mkdir -p gpurun_out/tune
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1; tail -1 gpurun_out/build.log
timeout 300 python tools/diag_small_cout.py > gpurun_out/diag_small_cout.log 2>&1; grep -v amdgpu.ids gpurun_out/diag_small_cout.log | tail -6
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q --tb=short -p no:cacheprovider -k "small_cout or vs_reference_golden" > gpurun_out/pytest_new.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_new.log; tail -6 gpurun_out/pytest_new.log
rm -f gpurun_out/tune/*.json
for c in smmnist_big5_ngf96 smmnist_big5 kth64_big_ngf128 bair_big_spade cityscapes_big cityscapes_big_variant; do
  ss=""; [ $c = bair_big_spade ] && ss="--subsample 200"
  timeout 900 python bench.py --config $c --steps 1 --warmup 1 $ss --no-cpu-baseline --no-tune-file --save-tuning gpurun_out/tune > gpurun_out/bench_tune_$c.json 2> gpurun_out/bench_tune_$c.err
  python -c "
import json
d=json.load(open('gpurun_out/bench_tune_$c.json'))
print('$c', d['value'], d['ms_per_step'], d['roofline']['frac'], 'selfcheck', d['selfcheck_max_abs'], 'f16x2 leg', d.get('f16x2_leg', {}).get('value'), {k:(v['launches'],v['ms']) for k,v in d['roofline']['breakdown'].items()})"
done
python - <<'PY'
import json,glob
from collections import Counter
for f in sorted(glob.glob('gpurun_out/tune/*.json')):
    t=json.load(open(f)); b=list(t)[0]
    print(f.split('/')[-1], dict(sorted(Counter(s for s,_ in t[b]).items())))
PY

#!/bin/bash
# 3x3 convs as 1x1 GEMMs (shape ids 22 / 23): parity, then re-tune config 2 / 1 / 3 and compare with the committed tables
mkdir -p gpurun_out/tune
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1; tail -1 gpurun_out/build.log
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q --tb=short -p no:cacheprovider -k "gemm_forms or forward_matches" > gpurun_out/pytest_new.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_new.log; tail -8 gpurun_out/pytest_new.log
rm -f gpurun_out/tune/*.json
for c in smmnist_big5_ngf96 smmnist_big5 kth64_big_ngf128; do
  timeout 900 python bench.py --config $c --steps 1 --warmup 1 --no-cpu-baseline --no-f16x2-leg > gpurun_out/bench_old_$c.json 2> gpurun_out/bench_old_$c.err
  timeout 900 python bench.py --config $c --steps 1 --warmup 1 --no-cpu-baseline --no-f16x2-leg --no-tune-file --save-tuning gpurun_out/tune > gpurun_out/bench_tune_$c.json 2> gpurun_out/bench_tune_$c.err
  python -c "
import json
for w in ('old','tune'):
    d=json.load(open('gpurun_out/bench_%s_$c.json' % w))
    print('$c', w, d['value'], d['ms_per_step'], 'selfcheck', d['selfcheck_max_abs'], {k:(v['launches'],v['ms']) for k,v in d['roofline']['breakdown'].items() if k in ('conv3x3','conv1x1')}, d['roofline']['conv3x3_families'].get('direct'))"
done
python - <<'PY'
import json,glob
from collections import Counter
for f in sorted(glob.glob('gpurun_out/tune/*.json')):
    t=json.load(open(f)); b=list(t)[0]
    print(f.split('/')[-1], dict(sorted(Counter(s for s,_ in t[b]).items())), [(i,s,c) for i,(s,c) in enumerate(t[b]) if s in (22,23)])
PY

#!/bin/bash
# round 4: the persistent Winograd kernel -- bit-identity tests, whole-net fixtures under the forced modes, bench with the autotuner
mkdir -p gpurun_out
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1; tail -1 gpurun_out/build.log
timeout 900 python -m pytest tests/test_gpu_wino3p.py -m gpu -q --tb=short -p no:cacheprovider -x > gpurun_out/pytest_w3p.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_w3p.log; tail -15 gpurun_out/pytest_w3p.log
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q --tb=short -p no:cacheprovider -k "bf16x3p or (bit_deterministic and (16 or 17 or 272))" > gpurun_out/pytest_w3p_net.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_w3p_net.log; tail -8 gpurun_out/pytest_w3p_net.log
timeout 900 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-f16x2-leg --no-tune-file --save-tuning gpurun_out/tune > gpurun_out/bench_r4b.json 2> gpurun_out/bench_r4b.err; echo "bench rc=$?" >> gpurun_out/bench_r4b.err; python - <<'PY'
import json
d=json.load(open('gpurun_out/bench_r4b.json'))
print(d['value'], d['ms_per_step'], d['roofline']['frac'], 'selfcheck', d['selfcheck_max_abs'], {k:(v['launches'],v['ms']) for k,v in d['roofline']['breakdown'].items()})
t=json.load(open('gpurun_out/tune/tune_smmnist_big5_ngf96_B64_bf16x3.json'))['64']
import collections
print(collections.Counter(x[0] for x in t))
PY
tail -2 gpurun_out/bench_r4b.err

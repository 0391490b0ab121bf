#!/bin/bash
# round 4, call 1: the parity tests at the benchmarked batch + the default bench line with its self-check
mkdir -p gpurun_out
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1; tail -1 gpurun_out/build.log
timeout 1200 python -m pytest tests/test_gpu_big_batch.py -m gpu -q --tb=short -p no:cacheprovider --durations=12 > gpurun_out/pytest_big.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_big.log; tail -40 gpurun_out/pytest_big.log
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q --tb=short -p no:cacheprovider -x -k "f16x2_overflow or fpndm or gamma" > gpurun_out/pytest_rng.log 2>&1; tail -3 gpurun_out/pytest_rng.log
timeout 900 python bench.py --no-cpu-baseline > gpurun_out/bench_r4a.json 2> gpurun_out/bench_r4a.err; echo "bench rc=$?" >> gpurun_out/bench_r4a.err; python - <<'PY'
import json
d=json.load(open('gpurun_out/bench_r4a.json'))
print(d['value'], d['ms_per_step'], d['roofline']['frac'], 'selfcheck', d['selfcheck_max_abs'], {k:(v['launches'],v['ms']) for k,v in d['roofline']['breakdown'].items()}, 'f16x2', d.get('f16x2_leg',{}).get('value'), d.get('f16x2_leg',{}).get('selfcheck_max_abs'))
PY
tail -2 gpurun_out/bench_r4a.err

#!/bin/bash
# one launch for all coef2 tables (SPADE): tests + config 4 bench; then the profile set of the headline config
mkdir -p gpurun_out
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1; tail -1 gpurun_out/build.log
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_big_batch.py -m gpu -q --tb=short -p no:cacheprovider -x -k "spade or bair or noise_in_cond or ddim_100" > gpurun_out/pytest_f.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_f.log; tail -5 gpurun_out/pytest_f.log
timeout 900 python bench.py --config bair_big_spade --steps 1 --warmup 1 --no-cpu-baseline --no-f16x2-leg > gpurun_out/bench_bair.json 2> gpurun_out/bench_bair.err; python - <<'PY'
import json
d=json.load(open('gpurun_out/bench_bair.json'))
print('bair', d['value'], d['ms_per_step'], d['roofline']['frac'], 'selfcheck', d['selfcheck_max_abs'], {k:(v['launches'],v['ms']) for k,v in d['roofline']['breakdown'].items()})
PY
bash tools/gpu_check.sh prof pmc_mfma pmc_l2 pmc_sq > gpurun_out/profile_set.log 2>&1; tail -5 gpurun_out/profile_set.log | cut -c1-300

#!/bin/bash
# SPADE configs: GPU suite, the cfg 4 bench line, rocprofv3 kernel stats of cfg 4 (subsample 100)
mkdir -p gpurun_out; export TMPDIR=/tmp; R=$(pwd)
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1; tail -1 gpurun_out/build.log
timeout 1200 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log; tail -4 gpurun_out/pytest_gpu.log
timeout 900 python bench.py --config bair_big_spade --steps 1 --warmup 1 --no-cpu-baseline > gpurun_out/bench_other_bair_big_spade.json 2> gpurun_out/bench_other_bair_big_spade.err
python -c "
import json
d=json.load(open('gpurun_out/bench_other_bair_big_spade.json'))
print('bair_big_spade', d['value'], d['ms_per_step'], d['roofline']['frac'], 'selfcheck', d['selfcheck_max_abs'], 'f16x2 leg', d.get('f16x2_leg', {}).get('value'), {k:(v['launches'],v['ms']) for k,v in d['roofline']['breakdown'].items()})"
bash tools/gpu_prof_cfg.sh 2>&1 | head -12

#!/bin/bash
# round 3, step t: attention -- two S accumulators / paired PV items, 4-wave form vs 8-wave complementary-phase form (A/B in one box)
mkdir -p gpurun_out
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1; tail -1 gpurun_out/build.log
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q --tb=short -p no:cacheprovider -x -k "attention or forward_vs" > gpurun_out/pytest_t.log 2>&1; tail -4 gpurun_out/pytest_t.log
export MCVD_LIB_PATH=$PWD/mcvd_pytorch_amd/libmcvd_hip_diag.so
for pp in 1 0 1 0; do
MCVD_ATTN_PP=$pp timeout 900 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-f16x2-leg > gpurun_out/bench_t.json 2> gpurun_out/bench_t.err; python - <<PY
import json
d=json.load(open('gpurun_out/bench_t.json'))
print('pp $pp', d['value'], d['ms_per_step'], {k:(v['launches'],v['ms']) for k,v in d['roofline']['breakdown'].items() if k in ('attention','conv3x3','conv1x1')})
PY
done
tail -2 gpurun_out/bench_t.err

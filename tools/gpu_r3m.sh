#!/bin/bash
# round 3, step m: prologue weight loads spread over the patch activation
mkdir -p gpurun_out
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1; tail -1 gpurun_out/build.log
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q --tb=short -p no:cacheprovider -x -k "conv or forward" > gpurun_out/pytest_m.log 2>&1; tail -3 gpurun_out/pytest_m.log
MCVD_LIB_PATH=$PWD/mcvd_pytorch_amd/libmcvd_hip_diag.so timeout 600 python tests/gpu_diag.py w3pro > gpurun_out/w3pro.log 2>&1; cat gpurun_out/diag_w3pro.txt | cut -c1-260; tail -3 gpurun_out/w3pro.log
timeout 900 python bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/bench_m.json 2> gpurun_out/bench_m.err; python - <<'PY'
import json
d=json.load(open('gpurun_out/bench_m.json'))
print(d['value'], d['ms_per_step'], d['roofline']['frac'], {k:(v['launches'],v['ms']) for k,v in d['roofline']['breakdown'].items()}, 'f16x2', d.get('f16x2_leg',{}).get('value'))
PY
tail -2 gpurun_out/bench_m.err

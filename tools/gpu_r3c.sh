#!/bin/bash
# round 3: the merged K loop of conv_wino3 -- parity, ablations (diagnostics library), bench
mkdir -p gpurun_out
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1; echo "build rc=$?" >> gpurun_out/build.log; tail -2 gpurun_out/build.log
timeout 600 python -m pytest tests/test_gpu_parity.py -q -x -p no:cacheprovider --tb=short -k "(test_conv2d and mfma and (s10 or s11)) or test_conv_bf16x3_is_fp32_accurate or test_default_kernels or test_conv_epilogue or (test_forward_vs_reference_golden and bf16x3) or (test_forward_is_bit_deterministic and (10 or 11 or 266))" > gpurun_out/pytest_d.log 2>&1; echo "pytest_d rc=$?" >> gpurun_out/pytest_d.log; tail -6 gpurun_out/pytest_d.log
MCVD_LIB_PATH=$PWD/mcvd_pytorch_amd/libmcvd_hip_diag.so MCVD_WEXP_ONLY=${EXPS:-0,4} MCVD_WEXP_CASES=${CASES:-0,2} timeout 600 python tests/gpu_diag.py w3exp > gpurun_out/w3exp.log 2>&1; cat gpurun_out/diag_w3exp.txt | cut -c1-240; tail -3 gpurun_out/w3exp.log
timeout 600 python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-f16x2-leg > gpurun_out/bench_b.json 2> gpurun_out/bench_b.err
echo "bench rc=$?"; python - <<'PY'
import json
d=json.load(open('gpurun_out/bench_b.json'))
print('value', d['value'], 'ms', d['ms_per_step'], 'frac', d['roofline']['frac'])
print({k:(v['launches'],v['ms']) for k,v in d['roofline']['breakdown'].items()})
PY
tail -3 gpurun_out/bench_b.err

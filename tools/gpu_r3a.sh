#!/bin/bash
# round 3, first GPU call: the new three-piece bf16 kernels -- op-level parity, whole-network fixtures, a first bench with both legs
mkdir -p gpurun_out
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1; echo "build rc=$?" >> gpurun_out/build.log; tail -2 gpurun_out/build.log
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -p no:cacheprovider --tb=short \
  -k "(test_conv2d and mfma and (s10 or s11 or s15 or s47 or s63 or s31)) or test_conv_bf16x3_is_fp32_accurate or test_default_kernels or test_non_finite or test_conv1x1_split or test_attention or test_conv_epilogue" \
  > gpurun_out/pytest_a.log 2>&1; echo "pytest_a rc=$?" >> gpurun_out/pytest_a.log; tail -15 gpurun_out/pytest_a.log
timeout 900 python -m pytest tests/test_gpu_parity.py -q -p no:cacheprovider --tb=short \
  -k "test_forward_vs_reference_golden or test_f16x2_option or test_imported_table" \
  > gpurun_out/pytest_b.log 2>&1; echo "pytest_b rc=$?" >> gpurun_out/pytest_b.log; tail -15 gpurun_out/pytest_b.log
timeout 600 python bench.py --steps 1 --warmup 1 --no-cpu-baseline --save-tuning gpurun_out/tune > gpurun_out/bench_a.json 2> gpurun_out/bench_a.err
echo "bench rc=$?"; python - <<'PY'
import json
d=json.load(open('gpurun_out/bench_a.json'))
print('value', d['value'], 'ms', d['ms_per_step'], 'frac', d['roofline']['frac'])
print({k:(v['launches'],v['ms']) for k,v in d['roofline']['breakdown'].items()})
print(d['roofline']['conv3x3_families'], d['roofline']['conv1x1_kernels'])
if 'f16x2_leg' in d:
    l=d['f16x2_leg']; print('f16x2', l['value'], l['ms_per_step'], {k:(v['launches'],v['ms']) for k,v in l['roofline']['breakdown'].items()})
PY
tail -3 gpurun_out/bench_a.err
timeout 300 python tests/gpu_diag.py convops > gpurun_out/convops.log 2>&1; cp gpurun_out/diag_convops.txt gpurun_out/diag_convops_bf16x3.txt; tail -4 gpurun_out/diag_convops.txt
MCVD_WEXP_ONLY=0 timeout 300 python tests/gpu_diag.py w3exp > gpurun_out/w3exp.log 2>&1; cat gpurun_out/diag_w3exp.txt | cut -c1-230

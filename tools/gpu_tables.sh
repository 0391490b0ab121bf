#!/bin/bash
# autotune every BASELINE config at its per-GPU batch and save the kernel tables (copied to profiles/ and committed: the bench pins them,
# tests/test_gpu_big_batch.py checks the pinned kernels against the reference fixtures at that batch)
mkdir -p gpurun_out/tune
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1; tail -1 gpurun_out/build.log
for c in ${CONFIGS:-smmnist_big5_ngf96 smmnist_big5 kth64_big_ngf128 bair_big_spade cityscapes_big cityscapes_big_variant}; do
  timeout 900 python bench.py --config $c --steps 1 --warmup 1 --no-cpu-baseline --no-tune-file --save-tuning gpurun_out/tune > gpurun_out/bench_tune_$c.json 2> gpurun_out/bench_tune_$c.err
  python -c "
import json
d=json.load(open('gpurun_out/bench_tune_$c.json'))
print('$c', d['value'], d['ms_per_step'], d['roofline']['frac'], 'selfcheck', d['selfcheck_max_abs'], 'f16x2 leg', d.get('f16x2_leg', {}).get('value'))"
done
ls gpurun_out/tune

#!/bin/bash
# Is a freshly autotuned kernel table better than the committed one?  For each config: autotune (saved to gpurun_out/tune), then same-box A/B
# committed table vs fresh table, two repetitions.   bash tools/gpu_retune_ab.sh bair_big_spade cityscapes_big   -> gpurun_out/retune_ab.txt
mkdir -p gpurun_out/tune; export TMPDIR=/tmp
: > gpurun_out/retune_ab.txt
for c in "$@"; do
  timeout 1200 python bench.py --config $c --steps 1 --warmup 1 --no-cpu-baseline --no-f16x2-leg --no-tune-file --save-tuning gpurun_out/tune > gpurun_out/bench_tune_$c.json 2> gpurun_out/bench_tune_$c.err
  f=$(ls gpurun_out/tune/tune_${c}_B*_bf16x3.json | head -1)
  for rep in 1 2; do for v in committed fresh; do
    if [ $v = fresh ]; then extra="--tune-cache $f"; else extra=""; fi
    timeout 900 python bench.py --config $c --steps 2 --warmup 1 --no-cpu-baseline --no-f16x2-leg $extra > gpurun_out/bench_ab.json 2> gpurun_out/bench_ab.err
    python - <<PY | tee -a gpurun_out/retune_ab.txt
import json
d=json.load(open('gpurun_out/bench_ab.json'))
print('$c $v', d['value'], 'frames/s', d['ms_per_step'], 'ms/step', d['config']['kernel_table'], d['valid'])
PY
  done; done
done

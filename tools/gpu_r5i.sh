#!/bin/bash
# upper bound of the tail-launch question: the graph WITHOUT its gn_finalize launches (timing only, wrong results) vs the shipped one
mkdir -p gpurun_out
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1; tail -1 gpurun_out/build.log
for c in smmnist_big5_ngf96 bair_big_spade cityscapes_big; do
  ss=""; [ $c = bair_big_spade ] && ss="--subsample 100"
  for f in 0 1 0 1; do
    MCVD_BENCH_OPTS=dbg_skip_finalize=$f timeout 600 python bench.py --config $c --steps 2 --warmup 1 $ss --no-cpu-baseline --no-f16x2-leg --no-selfcheck > gpurun_out/bench_tail_${c}_$f.json 2> gpurun_out/bench_tail_${c}_$f.err
    python -c "
import json
d=json.load(open('gpurun_out/bench_tail_${c}_$f.json'))
print('$c dbg_skip_finalize $f', d['value'], d['ms_per_step'])"
  done
done

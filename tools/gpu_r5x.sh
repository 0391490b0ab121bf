#!/bin/bash
# the K-split reduce pass finalizes the norm over its output: parity, then A/B (option gn_producer) on configs 2 / 4 / 5
mkdir -p gpurun_out
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1; tail -1 gpurun_out/build.log
timeout 1200 python -m pytest tests/test_gpu_parity.py -m gpu -q --tb=short -p no:cacheprovider -k "ksplit_reduce_pass or forward_matches or statistics or gn_" > gpurun_out/pytest_new.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_new.log; tail -8 gpurun_out/pytest_new.log
for c in smmnist_big5_ngf96 bair_big_spade cityscapes_big; do
  ss=""; [ $c = bair_big_spade ] && ss="--subsample 100"
  for f in 0 1 0 1; do
    MCVD_BENCH_OPTS=gn_producer=$f timeout 600 python bench.py --config $c --steps 2 --warmup 1 $ss --no-cpu-baseline --no-f16x2-leg > gpurun_out/bench_gnp_${c}_$f.json 2> gpurun_out/bench_gnp_${c}_$f.err
    python -c "
import json
d=json.load(open('gpurun_out/bench_gnp_${c}_$f.json'))
print('$c gn_producer $f', d['value'], d['ms_per_step'], d['selfcheck_max_abs'], {k:(v['launches'],v['ms']) for k,v in d['roofline']['breakdown'].items() if k in ('gn_coef',)})"
  done
done

#!/bin/bash
# N = 2 ranks on one GPU (gloo) vs N = 1 of the same global batch under forced kernel families: which kernel breaks under concurrency?
mkdir -p gpurun_out/tr
export TMPDIR=/tmp
C="--steps 1 --warmup 1 --subsample 5 --no-cpu-baseline --no-f16x2-leg --no-selfcheck --graph 0"
for opts in ${OPTLIST:-"naive_conv=1,naive_attn=1" "conv_shape=4,conv_shape1=5,naive_attn=2" "conv_shape=10,conv_shape1=5,naive_attn=2" "conv_shape=16,conv_shape1=5,naive_attn=2" "conv_shape=4,conv_shape1=15,naive_attn=2" "conv_shape=4,conv_shape1=5,naive_attn=4"}; do
  tag=$(echo $opts | tr ',=' '__')
  MCVD_BENCH_OPTS=$opts python bench.py --gpus 1 --batch 6 $C --dump-frames gpurun_out/tr/f1_$tag.pt > gpurun_out/tr/n1_$tag.json 2> gpurun_out/tr/n1_$tag.err
  for rep in ${REPS:-1 2 3}; do
    MCVD_BENCH_OPTS=$opts MCVD_DIST_BACKEND=gloo python bench.py --gpus 2 --batch 3 $C --dump-frames gpurun_out/tr/f2_${tag}_$rep.pt > gpurun_out/tr/n2_${tag}_$rep.json 2> gpurun_out/tr/n2_${tag}_$rep.err
  done
  python - <<PY
import torch
f1=torch.load('gpurun_out/tr/f1_$tag.pt')
for rep in range(1, 9):
    try:
        f2=torch.load(f'gpurun_out/tr/f2_${tag}_{rep}.pt')
        d=(f1-f2).abs().flatten(1).max(dim=1).values
        print('$opts', 'rep', rep, [f'{v:.1e}' for v in d.tolist()])
    except Exception as e:
        print('$opts', rep, 'failed', e)
PY
done

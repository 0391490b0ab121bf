#!/bin/bash
# deep K split (shape ids 18 / 19 / 20): layer tests, whole-network modes, determinism, then fresh kernel tables
mkdir -p gpurun_out/tune
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1; tail -1 gpurun_out/build.log
timeout 900 python -m pytest tests/test_gpu_ksplit_deep.py tests/test_gpu_wino3p.py -m gpu -q --tb=short -p no:cacheprovider -x 2>&1 | tail -8
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q --tb=short -p no:cacheprovider -k "bf16x3ks8 or bf16x3pks4 or deterministic-19 or deterministic-20 or deterministic[19] or deterministic[20]" 2>&1 | tail -8
CONFIGS="${CONFIGS:-cityscapes_big bair_big_spade cityscapes_big_variant smmnist_big5_ngf96 kth64_big_ngf128 smmnist_big5}" bash tools/gpu_tables.sh 2>&1 | grep -v "^libmcvd"

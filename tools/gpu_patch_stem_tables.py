"""Patch ONE entry of the committed kernel tables: the stem conv (the first 3x3 conv op of the plan) of the configs whose stem became eligible for
the GEMM form with the LDS-staged im2col in round 6 (15 / 21 input channels).  Everything else in the tables stays as committed and tested.
For each config: autotune at the bench's per-GPU batch (timing launches), read what the tuner chose for the stem, and if it is shape id 23 write
[23, cot] into that entry of profiles/tune_<config>_B<batch>_{bf16x3,f16x2}.json (under gpurun_out/tune_patched/ -- copy to profiles/ and commit).
Run on the GPU box:  python tools/gpu_patch_stem_tables.py"""
import ctypes as C
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import DEFAULTS, make_config  # noqa: E402
from mcvd_pytorch_amd import HipScoreNet, _lib, synthetic  # noqa: E402

OUT = os.path.join(ROOT, "gpurun_out", "tune_patched")
os.makedirs(OUT, exist_ok=True)
for name in ("kth64_big_ngf128", "cityscapes_big", "cityscapes_big_variant", "smmnist_big5", "smmnist_big5_ngf96", "bair_big_spade"):
    B = DEFAULTS[name][0]
    config = make_config(name)
    config.device = "cuda:0"
    net = HipScoreNet(config)
    net.load_state_dict(synthetic.random_state_dict(net, seed=123), strict=True)
    x, cond = synthetic.random_inputs(config, 0, B)
    t = torch.full((B,), 500, dtype=torch.long, device="cuda")
    net(x.cuda(), t, cond=cond.cuda())                      # autotunes at B
    torch.cuda.synchronize()
    table = net.get_tuning(B)
    info = (C.c_int * 8)()
    stem = None
    for i in range(len(table)):
        _lib.check(_lib.lib.mcvd_model_op_info(net._model, i, info), "op_info")
        if info[0] == 3 and info[2] == 3 and info[1] == 2:   # conv, 3x3, reference module 2 = the stem
            stem = i
            break
    print(name, "B", B, "stem op", stem, "tuner chose", table[stem] if stem is not None else None, flush=True)
    if stem is None or table[stem][0] != 23:
        continue
    for arith in ("bf16x3", "f16x2"):
        path = os.path.join(ROOT, "profiles", f"tune_{name}_B{B}_{arith}.json")
        if not os.path.exists(path):
            continue
        d = json.load(open(path))
        old = d[str(B)][stem]
        d[str(B)][stem] = [23, int(table[stem][1])]
        json.dump(d, open(os.path.join(OUT, os.path.basename(path)), "w"))
        print("   ", os.path.basename(path), "entry", stem, old, "->", d[str(B)][stem], flush=True)
    del net
    torch.cuda.empty_cache()

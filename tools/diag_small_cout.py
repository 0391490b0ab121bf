"""Event timing of the network's LAST conv (ngf -> C * num_frames couts) under the kernel families that can serve it (tests/hiputil.Ctx, product
library): 10 / 16 = three-piece bf16 Winograd on a padded 32-cout tile (per item / persistent), 4 = fp32 Winograd, 0 = direct fp32 MFMA,
21 = the fp32 VALU direct conv for <= 16 couts (conv_small_cout.cpp)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from tests.hiputil import Ctx
from mcvd_pytorch_amd import _lib

ctx = Ctx()
for (B, cin, cout, H) in ((64, 96, 5, 64), (64, 64, 5, 64), (32, 128, 5, 64), (16, 96, 15, 64), (8, 128, 15, 128)):
    g = torch.Generator().manual_seed(3)
    x = torch.randn(B, cin, H, H, generator=g).cuda()
    w = (torch.randn(cout, cin, 3, 3, generator=g) / (cin * 9) ** 0.5).cuda()
    b = torch.zeros(cout).cuda()
    coef = torch.ones(B, cin, 2).cuda()
    line = f"B{B:3d} cin{cin:4d} cout{cout:3d} H{H:4d}:"
    ref = None
    for shp in (10, 16, 4, 0, 21):
        ctx.opt("conv_shape", shp)
        for _ in range(2):
            y = ctx.conv2d(x, w, b, coef=coef, act=1)
        ran = _lib.lib.mcvd_last_conv_kernel()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            ctx.conv2d(x, w, b, coef=coef, act=1)
        e1.record()
        torch.cuda.synchronize()
        if ref is None:
            ref = y
        err = float((y - ref).abs().max() / ref.abs().max())
        line += f"  [{shp}->{ran}: {e0.elapsed_time(e1) * 1e3 / 5:7.1f} us, {err:.1e}]"
    ctx.opt("conv_shape", -1)
    print(line, "   (each call repacks the weights: +15-20 us of pack kernels in every column)", flush=True)

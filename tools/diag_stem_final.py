"""What the stem (10 -> 96) and the final (96 -> 5 / 15) 3x3 convs of a 64 x 64 network would cost as 1x1 GEMMs on the bf16x3 kernel:
stem = im2col (90 -> 96 rows) + 1x1 96 -> 96; final = 1x1 96 -> 9 * Cout ("taps as outputs") + a shift-and-add pass.  Run under
rocprofv3 --kernel-trace --stats: the kernel durations are the answer (the op-level entry point also packs weights per call)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from tests.hiputil import Ctx

ctx = Ctx()
g = torch.Generator().manual_seed(3)
B, H = int(os.environ.get("B", 64)), int(os.environ.get("H", 64))
x = torch.randn(B, 96, H, H, generator=g).cuda()
coef = torch.stack([1 + 0.3 * torch.randn(B, 96, generator=g), 0.3 * torch.randn(B, 96, generator=g)], dim=-1).cuda()
for cout in (45, 135):
    w = (torch.randn(cout, 96, 1, 1, generator=g) / 10).cuda(); b = torch.zeros(cout).cuda()
    for cot in (1, 2, 3):
        ctx.opt("conv_shape", 15); ctx.opt("conv_cot", cot)
        for _ in range(5):
            y = ctx.conv2d(x, w, b, coef=coef, act=1)
        torch.cuda.synchronize()
w = (torch.randn(96, 96, 1, 1, generator=g) / 10).cuda(); b = torch.zeros(96).cuda()
for cot in (1, 2, 3):
    ctx.opt("conv_shape", 15); ctx.opt("conv_cot", cot)
    for _ in range(5):
        y = ctx.conv2d(x, w, b)
    torch.cuda.synchronize()
# the real layers for comparison
ctx.opt("conv_shape", -1); ctx.opt("conv_cot", 0)
x10 = torch.randn(B, 10, H, H, generator=g).cuda()
w3 = (torch.randn(96, 10, 3, 3, generator=g) / 10).cuda()
for _ in range(5):
    y = ctx.conv2d(x10, w3, b)
w5 = (torch.randn(5, 96, 3, 3, generator=g) / 30).cuda(); b5 = torch.zeros(5).cuda()
for _ in range(5):
    y = ctx.conv2d(x, w5, b5, coef=coef, act=1)
torch.cuda.synchronize()
print("done")

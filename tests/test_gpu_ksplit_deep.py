"""GPU tests of the 4- and 8-part K split of the three-piece bf16 Winograd kernels (shape ids 18 / 19 on conv_wino3.cpp, 20 on the
persistent conv_wino3p.cpp).  At the small per-GPU batches of BASELINE configs 4 / 5 the 8x8 and 16x16 layers have fewer (region, cout
tile) pairs than the chip has CUs even after the 2-way split; more parts of the input channels put a workgroup on every CU.  Every
part computes the same exact piece products over its own channel range, the reduce pass sums the parts in index order, so

  * the result differs from the unsplit kernel only by the fp32 association of the channel sum (held to 2e-6 of the output scale) and
    meets the same F.conv2d contract as every other kernel;
  * it is bit-deterministic, and the persistent form (20) is bit-identical to the plain 4-part form (18): same parts, same reduce;
  * the GroupNorm partials are those of the reduce pass (final values), one per (sample, channel) plane;
  * a layer without the chunks for the requested depth degrades to the next shallower split (19 -> 18 -> 11 -> 10, 20 -> 17 -> 16).
"""
import pytest
import torch
import torch.nn.functional as F

from oracle import unet_ref

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    from tests.hiputil import Ctx
    return Ctx()


# (B, C0, C1, Cout, H, coef, act, res, conv_shape asked, kernel expected to run)
CASES = [
    (2, 256, 0, 256, 8, True, 1, True, 18, 18),      # 8x8, 16 chunks in 4 parts of 4
    (2, 256, 0, 256, 8, True, 1, True, 19, 19),      # ... in 8 parts of 2 (the minimum a part takes)
    (5, 256, 0, 128, 8, True, 1, False, 19, 19),     # odd batch: the last 8x8 region holds one image
    (1, 512, 256, 256, 16, True, 1, True, 19, 19),   # 16x16 over a concat: 48 chunks, 6 per part, the seam at a part boundary
    (2, 160, 96, 96, 16, True, 1, True, 18, 18),     # the concat seam inside part 2 (chunks 8..11, seam at chunk 10)
    (3, 256, 0, 128, 16, True, 1, True, 20, 20),     # persistent, 4 parts of 4 chunks as items
    (2, 512, 0, 96, 16, False, 0, False, 20, 20),    # ... raw input, 8 chunks per part
    (3, 192, 0, 128, 16, True, 1, True, 20, 17),     # 12 chunks: parts of 3 < 4 -> two halves of 6
    (2, 96, 0, 96, 8, True, 1, True, 19, 11),        # 6 chunks: neither 8 nor 4 parts -> 2
    (2, 160, 0, 64, 16, True, 1, False, 18, 11),     # 10 chunks: not a multiple of 4 -> 2
    (2, 32, 0, 64, 16, True, 0, False, 19, 10),      # 2 chunks: no split at all
    (2, 256, 0, 96, 32, True, 1, True, 19, 19),      # 32x32 through the op API (the model keeps a partial-sum buffer for planes up to 16x16 only)
]


def _inputs(case, seed=23):
    B, C0, C1, Cout, H = case[:5]
    g = torch.Generator().manual_seed(seed)
    x0 = torch.randn(B, C0, H, H, generator=g)
    x1 = torch.randn(B, C1, H, H, generator=g) if C1 else None
    Cin = C0 + C1
    w = torch.randn(Cout, Cin, 3, 3, generator=g) / (Cin * 9) ** 0.5
    bias = 0.1 * torch.randn(Cout, generator=g)
    coef = torch.stack([1 + 0.3 * torch.randn(B, Cin, generator=g), 0.3 * torch.randn(B, Cin, generator=g)], dim=-1) if case[5] else None
    res = torch.randn(B, Cout, H, H, generator=g) if case[7] else None
    return x0, x1, w, bias, coef, res


@pytest.mark.parametrize("case", CASES, ids=lambda c: "B{}_c{}+{}_o{}_H{}_s{}".format(c[0], c[1], c[2], c[3], c[4], c[8]))
def test_deep_k_split(ctx, case):
    from mcvd_pytorch_amd import _lib
    B, C0, C1, Cout, H, use_coef, act, use_res, ask, expect = case
    x0, x1, w, bias, coef, res = _inputs(case)
    scale = 0.70710678 if use_res else 1.0
    dev = lambda t: t.cuda().contiguous() if t is not None else None
    kw = dict(x1=dev(x1), coef=dev(coef), act=act, res=dev(res), scale=scale)
    try:
        ctx.opt("conv_shape", 10)
        base, base_st, base_np = ctx.conv2d_stats(dev(x0), dev(w), dev(bias), **kw)
        assert _lib.lib.mcvd_last_conv_kernel() == 10
        ctx.opt("conv_shape", ask)
        got, got_st, got_np = ctx.conv2d_stats(dev(x0), dev(w), dev(bias), **kw)
        ran = _lib.lib.mcvd_last_conv_kernel()
        again, again_st, _ = ctx.conv2d_stats(dev(x0), dev(w), dev(bias), **kw)
        twin = None
        if expect == 20:                         # the plain 4-part launch: same parts, same reduce
            ctx.opt("conv_shape", 18)
            twin, twin_st, _ = ctx.conv2d_stats(dev(x0), dev(w), dev(bias), **kw)
            assert _lib.lib.mcvd_last_conv_kernel() == 18
    finally:
        ctx.opt("conv_shape", -1)
    assert ran == expect, f"{case}: kernel family {ran} ran, expected {expect}"
    assert torch.equal(got, again) and (not got_np or torch.equal(got_st, again_st)), f"{case}: two launches differ"
    if twin is not None:
        assert torch.equal(got, twin) and torch.equal(got_st, twin_st), f"{case}: persistent 4-part result differs from the plain 4-part one"
    sc = max(base.abs().max().item(), 1.0)
    assert (got - base).abs().max().item() <= 2e-6 * sc, f"{case}: differs from the unsplit kernel by {(got - base).abs().max().item():.3e}"
    if ran == 10:
        assert torch.equal(got, base)
    elif got_np:                                 # partials (sum, M2 about the partial's own mean) of the FINAL values, one per plane here
        assert got_np == 1 and ran in (11, 17, 18, 19, 20)
        flat = got.flatten(2).double()
        want_sum, want_m2 = flat.sum(-1), ((flat - flat.mean(-1, keepdim=True)) ** 2).sum(-1)
        assert (got_st[:, :, 0, 0].double() - want_sum).abs().max().item() <= 1e-4 * max(want_sum.abs().max().item(), 1.0)
        assert (got_st[:, :, 0, 1].double() - want_m2).abs().max().item() <= 1e-4 * want_m2.abs().max().item()
    else:
        assert ran in (18, 19) and H == 32       # planes above 16x16: the reduce pass emits no partials
    xin = torch.cat([x0, x1], 1) if C1 else x0
    if use_coef:
        xin = xin * coef[..., 0][:, :, None, None] + coef[..., 1][:, :, None, None]
    if act:
        xin = unet_ref.silu(xin)
    ref = F.conv2d(xin.double(), w.double(), bias.double(), padding=1)
    if use_res:
        ref = ref + res.double()
    ref = (ref * scale).float()
    assert (got.cpu() - ref).abs().max().item() <= 2e-5 * sc + 1e-4 * ref.abs().max().item()

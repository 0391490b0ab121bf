"""GPU tests of the PERSISTENT three-piece bf16 Winograd kernel (conv_wino3p.cpp, shape ids 16 / 17): the same arithmetic in the same
order as conv_wino3_kernel (shape ids 10 / 11), so every output value and every GroupNorm partial must be BIT-IDENTICAL to that
kernel's -- which the rest of the suite holds to F.conv2d, to fp64 and to the reference fixtures.  What is new is the control flow:
one workgroup walks a range of (region, cout tile[, K half]) items and the staging pipeline runs on across item boundaries; a new
sample restarts it.  The context option "persist_grid" launches the kernel with a FEW workgroups so that small tensors exercise long
item ranges, ranges that start / end in the middle of a sample, several runs per workgroup and more workgroups than items.
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


def _g(seed):
    return torch.Generator().manual_seed(seed)


# (B, C0, C1, Cout, H, coef, act, res, K split, kernel expected to run under conv_shape 16 / 17)
CASES = [
    (2, 96, 0, 96, 64, True, 1, True, 0, 16),        # ResBlock Conv_1 @64: 64 items of 6 chunks, one cout tile
    (3, 192, 96, 192, 32, True, 1, False, 0, 16),    # up-path concat input, two cout tiles (the patch is reused by the next item)
    (6, 96, 0, 288, 32, True, 1, True, 0, 16),       # three cout tiles; 144 items, ranges cross samples at every grid size
    (2, 96, 0, 5, 64, True, 1, False, 0, 16),        # final conv: Cout = 5, one padded 32-channel tile (COT = 1)
    (3, 64, 0, 128, 16, True, 1, True, 0, 16),       # 64-channel cout tile (COT = 2), exactly 4 chunks (the minimum the stream takes)
    (1, 64, 0, 32, 128, True, 1, True, 0, 16),       # 128x128: 128 regions of one sample
    (2, 72, 0, 32, 16, True, 0, False, 0, 16),       # affine prologue only (PRO 1), ragged last chunk (72 channels in 5 chunks)
    (3, 96, 0, 96, 16, False, 0, False, 0, 16),      # raw input (PRO 0)
    (2, 10, 0, 96, 64, False, 0, False, 0, 10),      # stem: one chunk -- not served, conv_wino3_kernel takes the launch
    (5, 48, 16, 96, 8, True, 1, True, 0, 10),        # 8x8 images: stay with the two-images-per-workgroup form
    (2, 288, 0, 288, 16, True, 1, True, 1, 17),      # 2-way K split, the halves are items (9 chunks each: odd -> the buffer parity flips per item)
    (2, 384, 288, 288, 16, True, 1, True, 1, 17),    # ... over a concat (21 chunks per half, the seam falls inside the first half)
    (3, 96, 0, 96, 16, True, 1, False, 1, 16),       # 6 chunks: halves of 3 < 4 -> runs unsplit, persistent
    (3, 32, 0, 64, 16, True, 1, False, 1, 10),       # 2 chunks: neither split nor persistent
]


def _inputs(case, seed=11):
    B, C0, C1, Cout, H, use_coef, act, use_res, ks2, _ = case
    g = _g(seed)
    x0 = torch.randn(B, C0, H, H, generator=g)
    x1 = torch.randn(B, C1, H, H, generator=g) if C1 else None
    Cin = C0 + C1
    w = torch.randn(Cout, Cin, 3, 3, generator=g) / (Cin * 9) ** 0.5
    bias = 0.1 * torch.randn(Cout, generator=g)
    coef = torch.stack([1 + 0.3 * torch.randn(B, Cin, generator=g), 0.3 * torch.randn(B, Cin, generator=g)], dim=-1) if use_coef else None
    res = torch.randn(B, Cout, H, H, generator=g) if use_res else None
    return x0, x1, w, bias, coef, res


@pytest.mark.parametrize("case", CASES, ids=lambda c: "B{}_c{}+{}_o{}_H{}_pro{}_ks{}".format(c[0], c[1], c[2], c[3], c[4], int(c[5]) + c[6], c[8]))
@pytest.mark.parametrize("grid", [0, 1, 7, 40], ids=["per_cu", "g1", "g7", "g40"])
def test_persistent_kernel_is_bit_identical(ctx, case, grid):
    from mcvd_pytorch_amd import _lib
    B, C0, C1, Cout, H, use_coef, act, use_res, ks2, expect = case
    x0, x1, w, bias, coef, res = _inputs(case)
    scale = 0.70710678 if use_res else 1.0
    dev = lambda t: t.cuda().contiguous() if t is not None else None
    kw = dict(x1=dev(x1), coef=dev(coef), act=act, res=dev(res), scale=scale)
    try:
        ctx.opt("conv_shape", 11 if expect == 17 else 10)       # the same K split (or none) as the persistent launch will take
        want, want_st, want_np = ctx.conv2d_stats(dev(x0), dev(w), dev(bias), **kw)
        base_ran = _lib.lib.mcvd_last_conv_kernel()
        ctx.opt("conv_shape", 17 if ks2 else 16)
        ctx.opt("persist_grid", grid)
        got, got_st, got_np = ctx.conv2d_stats(dev(x0), dev(w), dev(bias), **kw)
        ran = _lib.lib.mcvd_last_conv_kernel()
    finally:
        ctx.opt("conv_shape", -1)
        ctx.opt("persist_grid", 0)
    assert ran == expect, f"{case}: kernel family {ran} ran, expected {expect}"
    assert base_ran == (11 if expect == 17 else 10)
    assert torch.equal(got, want), f"{case} grid {grid}: {int((got != want).sum())} of {got.numel()} values differ from conv_wino3_kernel, max {float((got - want).abs().max()):.3e}"
    assert got_np == want_np
    if want_np:
        assert torch.equal(got_st, want_st), f"{case} grid {grid}: GroupNorm partials differ"
    # and the usual contract against the direct convolution
    xin = torch.cat([x0, x1], 1) if C1 else x0
    if use_coef:
        xin = xin * coef[..., 0][:, :, None, None] + coef[..., 1][:, :, None, None]
    if act:
        xin = unet_ref.silu(xin)
    ref = F.conv2d(xin, w, bias, padding=1)
    if use_res:
        ref = ref + res
    ref = ref * scale
    sc = max(ref.abs().max().item(), 1.0)
    assert (got.cpu() - ref).abs().max().item() <= 2e-5 * sc + 1e-4 * ref.abs().max().item()


def test_persistent_kernel_leaves_no_stale_pipeline_state(ctx):
    """Two different problems back to back on the same stream (different region geometry, channel count and prologue): nothing of the
    first launch's tables may leak into the second (the slot tables and the coefficient table live in LDS per launch)."""
    from mcvd_pytorch_amd import _lib
    outs = []
    try:
        for rep in range(2):
            for case in (CASES[0], CASES[4], CASES[7], CASES[1]):
                x0, x1, w, bias, coef, res = _inputs(case, seed=5)
                dev = lambda t: t.cuda().contiguous() if t is not None else None
                ctx.opt("conv_shape", 16)
                ctx.opt("persist_grid", 5)
                y = ctx.conv2d(dev(x0), dev(w), dev(bias), x1=dev(x1), coef=dev(coef), act=case[6], res=dev(res), scale=1.0)
                assert _lib.lib.mcvd_last_conv_kernel() == 16
                outs.append(y.clone())
    finally:
        ctx.opt("conv_shape", -1)
        ctx.opt("persist_grid", 0)
    for a, b in zip(outs[:4], outs[4:]):
        assert torch.equal(a, b)


def test_runtime_selftest_of_the_hand_scheduled_kernels(ctx):
    """mcvd_ctx_selftest: shape 10 vs the compiler-scheduled fp32 Winograd kernel within 1e-4, shape 16 bit-equal to shape 10,
    on the running device; 0 the first time, 1 (cached verdict) afterwards; the forced kernel options of the context are restored."""
    from mcvd_pytorch_amd import _lib
    ctx.opt("conv_shape", 12)
    try:
        assert _lib.lib.mcvd_ctx_selftest(ctx.h) in (0, 1), _lib.last_error()
        assert _lib.lib.mcvd_ctx_selftest(ctx.h) == 1
        # the option was put back: a conv under it still takes the forced family's fallback chain, not shape 16
        x = torch.randn(1, 64, 16, 16, device="cuda")
        w = torch.randn(32, 64, 3, 3, device="cuda") / 24.0
        ctx.conv2d(x, w, torch.zeros(32, device="cuda"))
        assert _lib.lib.mcvd_last_conv_kernel() in (4, 12)          # (raw input: the f16x2 family applies only with a GroupNorm prologue)
    finally:
        ctx.opt("conv_shape", -1)


def test_models_run_the_selftest_at_finalize():
    from oracle import synth
    from mcvd_pytorch_amd import _lib
    from mcvd_pytorch_amd.scorenet import HipScoreNet
    config = synth.make_config("tiny")
    config.device = "cuda:0"
    net = HipScoreNet(config)
    net.load_state_dict(synth.make_state_dict(config, seed=123), strict=True)
    net.sync_parameters(force=True)
    assert _lib.lib.mcvd_ctx_selftest(net._ctx) == 1          # already run (and passed) by mcvd_model_finalize

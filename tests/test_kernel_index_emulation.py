"""CPU: lane-level emulation of the MFMA kernels' addressing (tools/emulate_kernels.py) against plain conv /
attention.  Finds layout bugs (LDS offsets, operand lane maps, accumulator->NCHW map) without a GPU."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from tools import emulate_kernels as E


def _round_up(a, b):
    return (a + b - 1) // b * b


@pytest.mark.parametrize("B,C0,C1,Cout,H,ks,COT,PXT,SPLIT", [
    (1, 6, 5, 40, 16, 3, 2, 2, False),     # 256-px tile = whole 16x16 image, concat, ragged Cin/Cout
    (2, 8, 0, 32, 16, 3, 1, 1, False),     # 128-px tile = half image (halo rows come from the neighbour rows)
    (3, 8, 8, 32, 8, 3, 1, 2, False),      # 4 images per tile, B not a multiple of 4
    (2, 8, 0, 32, 8, 3, 1, 2, True),       # split-K across waves
    (1, 8, 0, 32, 32, 3, 1, 1, False),     # W=32: 4 rows per tile
    (2, 20, 0, 32, 8, 1, 1, 2, True),      # 1x1, CK=16 split-K (k-pairs 0..7 over 4 waves)
    (1, 16, 16, 64, 16, 1, 2, 1, False),   # 1x1 concat
])
def test_conv_index_maps(B, C0, C1, Cout, H, ks, COT, PXT, SPLIT):
    g = torch.Generator().manual_seed(1)
    x0 = torch.randn(B, C0, H, H, generator=g)
    x1 = torch.randn(B, C1, H, H, generator=g) if C1 else None
    Cin = C0 + C1
    w = torch.randn(Cout, Cin, ks, ks, generator=g)
    bias = torch.randn(Cout, generator=g)
    coef = torch.stack([1 + 0.3 * torch.randn(B, Cin, generator=g), 0.3 * torch.randn(B, Cin, generator=g)], -1)
    res = torch.randn(B, Cout, H, H, generator=g)
    CK = 8 if ks == 3 else 16
    CinP, CoutP = _round_up(Cin, CK), _round_up(Cout, 32 * COT)
    wp = E.pack_weight(w.numpy(), CinP, CoutP)
    got = E.conv_emulate(x0.numpy(), None if x1 is None else x1.numpy(), wp, bias.numpy(), coef.numpy(), 1, res.numpy(),
                         0.5, Cout, CoutP, CinP, ks, CK, COT, PXT, SPLIT)
    xin = torch.cat([x0, x1], 1) if C1 else x0
    xin = xin * coef[..., 0][:, :, None, None] + coef[..., 1][:, :, None, None]
    xin = xin * torch.sigmoid(xin)
    want = (F.conv2d(xin.double(), w.double(), bias.double(), padding=ks // 2) + res.double()) * 0.5
    np.testing.assert_allclose(got, want.numpy(), rtol=1e-6, atol=1e-6)


@pytest.mark.parametrize("B,C,heads,S", [(1, 64, 2, 64), (1, 96, 1, 160)])
def test_attention_index_maps(B, C, heads, S):
    g = torch.Generator().manual_seed(2)
    qkv = torch.randn(B, 3 * C, S, generator=g).double()
    D = C // heads
    q, k, v = (qkv[:, i * C:(i + 1) * C].reshape(B * heads, D, S) for i in range(3))
    w = torch.softmax(torch.matmul(q.transpose(1, 2), k) * D ** -0.5, dim=-1)
    want = torch.matmul(v, w.transpose(1, 2)).reshape(B, C, S)
    got = E.attn_emulate(qkv.numpy(), heads)
    np.testing.assert_allclose(got, want.numpy(), rtol=1e-8, atol=1e-8)


@pytest.mark.parametrize("B,C0,C1,Cout,H,W,COT", [(1, 16, 5, 40, 16, 16, 2), (2, 8, 0, 96, 8, 32, 3), (1, 16, 8, 32, 16, 32, 1), (1, 10, 0, 32, 8, 16, 1), (3, 16, 20, 40, 8, 16, 2),
                                                    (3, 16, 16, 40, 8, 8, 2), (2, 10, 0, 96, 8, 8, 3)])
def test_winograd_index_maps(B, C0, C1, Cout, H, W, COT):
    """conv_wino.cpp: patch slots, transform tasks, V layout, operand-major weights, MFMA lane maps, block id -> (region, cout
    tile), the LDS exchange and the 2x2 inverse transform."""
    g = torch.Generator().manual_seed(4)
    x0 = torch.randn(B, C0, H, W, generator=g)
    x1 = torch.randn(B, C1, H, W, generator=g) if C1 else None
    Cin = C0 + C1
    w = torch.randn(Cout, Cin, 3, 3, generator=g)
    bias = torch.randn(Cout, generator=g)
    coef = torch.stack([1 + 0.3 * torch.randn(B, Cin, generator=g), 0.3 * torch.randn(B, Cin, generator=g)], -1)
    res = torch.randn(B, Cout, H, W, generator=g)
    CinP, CoutP = _round_up(Cin, 16), _round_up(Cout, 32 * COT)
    up = E.pack_wino_weight(w.numpy(), CinP, CoutP, COT)
    got = E.wino_emulate(x0.numpy(), None if x1 is None else x1.numpy(), up, bias.numpy(), coef.numpy(), 1, res.numpy(), 0.5,
                         Cout, CoutP, CinP, COT)
    xin = torch.cat([x0, x1], 1) if C1 else x0
    xin = xin * coef[..., 0][:, :, None, None] + coef[..., 1][:, :, None, None]
    xin = xin * torch.sigmoid(xin)
    want = (F.conv2d(xin.double(), w.double(), bias.double(), padding=1) + res.double()) * 0.5
    np.testing.assert_allclose(got, want.numpy(), rtol=1e-6, atol=1e-6)


@pytest.mark.parametrize("CK", [16, 32])
@pytest.mark.parametrize("B,C0,C1,Cout,H,W,COT,pro", [
    (3, 16, 0, 40, 8, 8, 2, 1),        # HW=64: two images per pixel tile, B odd -> ragged last tile, ragged Cout
    (1, 16, 16, 96, 16, 16, 3, 0),     # concat, COT=3 (384 weight pieces: 1.5 DMA rounds)
    (5, 32, 0, 32, 4, 8, 1, 2),        # HW=32: four images per tile, ragged; affine + SiLU
    (1, 16, 0, 288, 16, 8, 9, 1),      # COT=9
    (2, 16, 0, 64, 16, 16, 1, 1),      # several cout tiles per pixel tile (XCD-grouped block ids)
])
def test_gemm1x1_index_maps(B, C0, C1, Cout, H, W, COT, pro, CK):
    """conv1x1_dma.cpp: DMA piece maps, coefficient table, block id -> tile map, epilogue map."""
    g = torch.Generator().manual_seed(5)
    if CK == 32:
        C0, C1 = 2 * C0, 2 * C1            # 32-channel chunks: channel counts and the concat seam on 32-boundaries
    x0 = torch.randn(B, C0, H, W, generator=g)
    x1 = torch.randn(B, C1, H, W, generator=g) if C1 else None
    Cin = C0 + C1
    w = torch.randn(Cout, Cin, 1, 1, generator=g)
    bias = torch.randn(Cout, generator=g)
    coef = torch.stack([1 + 0.3 * torch.randn(B, Cin, generator=g), 0.3 * torch.randn(B, Cin, generator=g)], -1)
    res = torch.randn(B, Cout, H, W, generator=g)
    CinP, CoutP = _round_up(Cin, 32), _round_up(Cout, 32 * COT)
    wp = E.pack_weight(w.numpy(), CinP, CoutP)
    got = E.gemm1x1_emulate(x0.numpy(), None if x1 is None else x1.numpy(), wp, bias.numpy(),
                            coef.numpy() if pro else None, pro == 2, res.numpy(), 0.5, Cout, CoutP, CinP, COT, CK)
    xin = torch.cat([x0, x1], 1) if C1 else x0
    if pro:
        xin = xin * coef[..., 0][:, :, None, None] + coef[..., 1][:, :, None, None]
    if pro == 2:
        xin = xin * torch.sigmoid(xin)
    want = (F.conv2d(xin.double(), w.double(), bias.double()) + res.double()) * 0.5
    np.testing.assert_allclose(got, want.numpy(), rtol=1e-6, atol=1e-6)

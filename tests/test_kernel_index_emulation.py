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


# ------------------------------------------------------------------------------------------------ two-piece fp16 kernels
# The f16x2 kernels read pre-split weights in an operand-major layout written by the pack kernels, and B operands the staging
# threads park in LDS.  Both sides must agree on (a) where a (cout, cin, position, piece) halfword lives and (b) the K-slot
# convention of the 32x32x16 MFMA (lane half h, element e <-> channel 2e + h of the 16-channel chunk).  The index arithmetic of
# conv_wino2h.cpp / conv1x1_h2.cpp is restated here lane by lane; the product of the operands as the MFMA would contract them must
# equal the plain contraction.  Pieces are marked by a factor (piece 1 = f * piece 0) so that a mixed-up piece shows.
def _mfma_32x32x16(A, Bm):
    """A[m][h][e], Bm[n][h][e] -> D[m][n]: both operands hold K slot (h, e) of their row / column."""
    return np.einsum("mhe,nhe->mn", A, Bm)


PIECE_F = {2: (1.0, 7.0), 3: (1.0, 7.0, 11.0)}               # weight piece markers
PIECE_G = {2: (1.0, 3.0), 3: (1.0, 3.0, 13.0)}               # activation piece markers
PRODUCTS = {2: ((0, 1), (1, 0), (0, 0)),                      # pieces.h: (weight piece, activation piece), smallest product first
            3: ((0, 2), (2, 0), (1, 1), (0, 1), (1, 0), (0, 0))}


@pytest.mark.parametrize("NP", [2, 3], ids=["f16x2", "bf16x3"])
@pytest.mark.parametrize("COT,Cout,Cin,CinP", [(3, 192, 40, 48), (2, 128, 32, 32), (1, 32, 16, 16)])
def test_wino_split_operand_layout(NP, COT, Cout, Cin, CinP):
    """conv_wino2h.cpp (NP = 2) / conv_wino3.cpp (NP = 3): pack_wino{2h,3}_weight_kernel vs *_LOAD_A / *_MF1, and *_WRITE_V vs *_LOAD_B."""
    rng = np.random.default_rng(0)
    BCO, CoutP, nch = 32 * COT, Cout, CinP // 16
    NQ, PW = NP * COT, 16 * 2 * 4 * 32
    U = rng.standard_normal((Cout, CinP, 16))
    U[:, Cin:] = 0
    mem = np.zeros(CinP * 16 * CoutP * NP)                      # halfwords (behind the header for NP = 2)
    for co in range(Cout):
        for ci in range(Cin):
            cotile, ct, cc = co // BCO, (co % BCO) // 32, ci & 15
            h, el = cc & 1, cc >> 1
            lane = h * 32 + (co & 31)
            base = (((cotile * nch + (ci >> 4)) * 16) * COT + ct) * (NP * 512) + (lane * 4 + (el >> 1)) * 2 + (el & 1)
            for xi in range(16):
                for p in range(NP):
                    mem[base + xi * COT * NP * 512 + p * 512] = PIECE_F[NP][p] * U[co, ci, xi]
    chunk = nch - 1
    V = rng.standard_normal((16, 16, 32))                       # [channel in chunk][position][tile]
    sV = np.zeros((NP * PW, 2))                                 # *_WRITE_V: words [piece][position][half][pair][tile] = (lo, hi)
    for tid in range(512):
        rg, s_tile, s_cp = (tid >> 6) >> 2, tid & 31, (tid & 255) >> 5
        s_ca = 4 * (s_cp >> 1) + (s_cp & 1)
        v_wr = ((8 * rg * 2 + (s_cp & 1)) * 4 + (s_cp >> 1)) * 32 + s_tile
        for row in range(2):
            for q in range(4):
                xi = (2 * rg + row) * 4 + q
                for p in range(NP):
                    g = PIECE_G[NP][p]
                    sV[v_wr + (row * 4 + q) * 256 + p * PW] = (g * V[s_ca, xi, s_tile], g * V[s_ca + 2, xi, s_tile])
    for cotile in range(Cout // BCO):
        for wave in range(8):
            for i in range(2):
                xi = 2 * wave + i
                for ct in range(COT):
                    for pa, pb in PRODUCTS[NP]:
                        f = PIECE_F[NP][pa] * PIECE_G[NP][pb]
                        A, Bm = np.zeros((32, 2, 8)), np.zeros((32, 2, 8))
                        for lane in range(64):
                            half, l31 = lane >> 5, lane & 31
                            q = (i * COT + ct) * NP + pa                                # *_LOAD_A / *_MF1: quad of (position, sub-tile, piece)
                            dw0 = (cotile * nch * 16 + 2 * wave) * (NQ * 256) + chunk * (16 * NQ * 256) + q * 256 + lane * 4
                            A[l31, half] = [mem[(dw0 + j) * 2 + k] for j in range(4) for k in range(2)]
                            qb = (((2 * wave + i) * 2 + half) * 4) * 32 + l31           # *_LOAD_B
                            Bm[l31, half] = [sV[pb * PW + qb + jp * 32][k] for jp in range(4) for k in range(2)]
                        co0 = cotile * BCO + ct * 32
                        want = f * np.einsum("mc,cn->mn", U[co0:co0 + 32, chunk * 16:chunk * 16 + 16, xi], V[:, xi, :])
                        assert np.allclose(_mfma_32x32x16(A, Bm), want), (cotile, wave, i, ct, pa, pb)


@pytest.mark.parametrize("NP", [2, 3], ids=["f16x2", "bf16x3"])
@pytest.mark.parametrize("COT,CinP,CoutP", [(3, 32, 192), (2, 16, 128), (1, 32, 96), (4, 16, 128)])
def test_conv1x1_split_operand_layout(NP, COT, CinP, CoutP):
    """conv1x1_h2.cpp: pack_conv1x1_{h2,b3}_kernel vs the DMA rounds (Q1_DMA) and the A-operand reads."""
    rng = np.random.default_rng(1)
    NS, WPC = CoutP // 32, COT * NP * 64
    W = rng.standard_normal((CinP, CoutP))
    mem = np.zeros(CinP * CoutP * NP)
    for ci in range(CinP):
        for co in range(CoutP):
            cc = ci & 15
            h, el = cc & 1, cc >> 1
            lane = h * 32 + (co & 31)
            o = (((ci >> 4) * NS + (co >> 5)) * NP) * 512 + (lane * 4 + (el >> 1)) * 2 + (el & 1)
            for p in range(NP):
                mem[o + p * 512] = PIECE_F[NP][p] * W[ci, co]
    X = rng.standard_normal((CinP, 128))
    for ctile in range(CoutP // (32 * COT)):
        for ch in range(CinP // 16):
            lds = np.full(COT * NP * 256 * 2, np.nan)           # the chunk's LDS image, filled by the DMA rounds (Q1_DMA)
            for s in range((WPC + 255) // 256):
                for wave in range(4):
                    q0 = (s * 256 + wave * 64) % WPC
                    for lane in range(64):
                        g = (ch * NS + ctile * COT) * (NP * 256) + (q0 + lane) * 4      # global dword of the lane's 16 bytes
                        lds[(q0 + lane) * 8:(q0 + lane) * 8 + 8] = mem[g * 2:g * 2 + 8]
            assert not np.isnan(lds).any()
            Bm = np.zeros((128, 2, 8))
            for n in range(128):
                for h in range(2):
                    Bm[n, h] = [X[ch * 16 + 2 * e + h, n] for e in range(8)]
            for ct in range(COT):
                for piece in range(NP):
                    A = np.zeros((32, 2, 8))
                    for lane in range(64):
                        idx = ((ct * NP + piece) * 64 + lane) * 8
                        A[lane & 31, lane >> 5] = lds[idx:idx + 8]
                    co0 = (ctile * COT + ct) * 32
                    want = PIECE_F[NP][piece] * np.einsum("cm,cn->mn", W[ch * 16:ch * 16 + 16, co0:co0 + 32], X[ch * 16:ch * 16 + 16])
                    assert np.allclose(_mfma_32x32x16(A, Bm), want), (ctile, ch, ct, piece)


@pytest.mark.parametrize("NP,D", [(2, 64), (3, 96), (3, 32)])
def test_attention_split_operand_staging_layout(NP, D):
    """attention_h2.cpp: the K planes [step][piece][dword j][64 lanes] (one b128 store of 4 consecutive keys per piece) against the S
    product's four b32 reads, and the permuted V staging lanes against the PV product's b128 read; both as the MFMA contracts them."""
    rng = np.random.default_rng(2)
    NST, DT = D // 16, D // 32
    K, V = rng.standard_normal((D, 32)), rng.standard_normal((D, 32))       # one key tile: [channel][key]
    sK = np.full((NST * NP * 4 * 64, 2), np.nan)                # dwords = (lo, hi)
    NIK = (4 * D + 255) // 256
    for tid in range(256):
        for i in range(NIK):
            idx = i * 256 + tid
            if idx >= 4 * D:
                continue
            rp, kq = idx >> 3, idx & 7
            st, j, h = rp >> 3, (rp >> 1) & 3, rp & 1
            c = 16 * st + 4 * j + h                             # rows c, c + 2 (gload)
            for p in range(NP):
                d1 = (st * NP * 4 + j) * 64 + h * 32 + kq * 4 + p * 256
                for i4 in range(4):
                    g = PIECE_G[NP][p]
                    sK[d1 + i4] = (g * K[c, 4 * kq + i4], g * K[c + 2, 4 * kq + i4])
    assert not np.isnan(sK).any()
    Q = rng.standard_normal((D, 32))                            # [channel][query]
    for s in range(NST):
        for p in range(NP):
            A, Bm = np.zeros((32, 2, 8)), np.zeros((32, 2, 8))
            for lane in range(64):
                half, l31 = lane >> 5, lane & 31
                A[l31, half] = [sK[((s * NP + p) * 4 + j) * 64 + lane][k] for j in range(4) for k in range(2)]
                Bm[l31, half] = [Q[16 * s + 4 * j + half + 2 * k, l31] for j in range(4) for k in range(2)]      # qp: dword j = channels c0, c0 + 2
            want = PIECE_G[NP][p] * np.einsum("ck,cq->kq", K[16 * s:16 * s + 16], Q[16 * s:16 * s + 16])
            assert np.allclose(_mfma_32x32x16(A, Bm), want), (s, p)
    sV = np.full((DT * 2 * NP * 64 * 4, 2), np.nan)
    for tid in range(256):
        wave, lane = tid >> 6, tid & 63
        v_c, v_h, v_s2, v_g = wave * 8 + ((lane >> 1) & 7), (lane >> 4) & 1, lane >> 5, lane & 1
        v_q = v_s2 * 4 + v_g * 2 + v_h
        for i in range(DT):
            c = i * 32 + v_c
            ct, m, j0 = c >> 5, c & 31, 2 * v_g
            for p in range(NP):
                g = PIECE_G[NP][p]
                d1 = ((ct * 2 + v_s2) * NP * 64 + v_h * 32 + m) * 4 + j0 + p * 256
                sV[d1] = (g * V[c, 4 * v_q], g * V[c, 4 * v_q + 1])
                sV[d1 + 1] = (g * V[c, 4 * v_q + 2], g * V[c, 4 * v_q + 3])
    assert not np.isnan(sV).any()
    P = rng.standard_normal((32, 32))                           # [key][query]
    for ct in range(DT):
        for p in range(NP):
            acc = np.zeros((32, 32))
            for s2 in range(2):
                A, Bm = np.zeros((32, 2, 8)), np.zeros((32, 2, 8))
                for lane in range(64):
                    half, l31 = lane >> 5, lane & 31
                    base = (((ct * 2 + s2) * NP + p) * 64 + lane) * 4
                    A[l31, half] = [sV[base + j][k] for j in range(4) for k in range(2)]
                    keys = [((r & 3) + 8 * (r >> 2) + 4 * half) for r in range(8 * s2, 8 * s2 + 8)]      # accumulator register r of S^T
                    Bm[l31, half] = [P[k, l31] for k in keys]
                acc += _mfma_32x32x16(A, Bm)
            want = PIECE_G[NP][p] * np.einsum("ck,kq->cq", V[ct * 32:ct * 32 + 32], P)
            assert np.allclose(acc, want), (ct, p)


def test_attention_h2_key_slot_convention():
    """PV product of attention_h2.cpp: the staging thread that holds keys 4q .. 4q+3 of a channel writes dwords (j0, j0 + 1) of lane
    half h in step s2; the probabilities of lane half h sit in accumulator registers r = 8 s2 + e with key (r&3) + 8 (r>>2) + 4h.
    Both must name the same key for every (h, s2, dword j, low / high half)."""
    for q in range(8):
        h, s2, j0 = q & 1, q >> 2, 2 * ((q >> 1) & 1)
        for pair in range(2):
            for bhalf in range(2):
                key_staged = 4 * q + 2 * pair + bhalf
                r = 8 * s2 + 2 * (j0 + pair) + bhalf
                assert (r & 3) + 8 * (r >> 2) + 4 * h == key_staged

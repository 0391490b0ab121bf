"""Lane-level numpy emulation of the index arithmetic of the MFMA conv and attention kernels
(mcvd_pytorch_amd/csrc/kernels/conv_mfma.h, attention.cpp).  It transcribes the kernels' addressing -- staging slots,
LDS layout, MFMA operand/accumulator lane maps (A[i=l&31][k=l>>5], B[k=l>>5][j=l&31], D col=l&31,
row=(r&3)+8*(r>>2)+4*(l>>5)) -- so layout bugs can be found on a machine without a GPU."""
import numpy as np


def mfma_32x32x2(a, b, acc):
    """a,b: [64] per-lane operands; acc: [16,64] per-lane accumulators."""
    A = np.zeros((32, 2), np.float64)
    Bm = np.zeros((2, 32), np.float64)
    for l in range(64):
        A[l & 31, l >> 5] = a[l]
        Bm[l >> 5, l & 31] = b[l]
    D = A @ Bm
    for l in range(64):
        for r in range(16):
            acc[r, l] += D[(r & 3) + 8 * (r >> 2) + 4 * (l >> 5), l & 31]


def conv_emulate(x0, x1, wp, bias, coef, act, res, scale, Cout, CoutP, CinP, ks, CK, COT, PXT, SPLIT):
    B, C0, H, W = x0.shape
    C1 = 0 if x1 is None else x1.shape[1]
    Cin = C0 + C1
    KK = ks * ks
    HALO = 1 if ks == 3 else 0
    BPX = PXT * 32 if SPLIT else 4 * PXT * 32
    BCO = COT * 32
    RT = BPX // W
    rpi = min(RT, H)
    nimg = RT // rpi
    P = ((20 if W == 8 else W + 4) if HALO else W)
    IS = (rpi + 2 * HALO) * P
    PS = (nimg * IS + (4 if HALO else 0) + 3) // 4 * 4
    HW = H * W
    MAXA = 4 if ks == 3 else (CK * BPX // 4 + 255) // 256
    WCOUNT = CK * KK * BCO // 4
    MAXW = (WCOUNT + 255) // 256
    y = np.zeros((B, Cout, H, W), np.float64)
    n_ptiles = (B * H + RT - 1) // RT
    silu = lambda v: v / (1 + np.exp(-v))
    src_all = x0 if x1 is None else np.concatenate([x0, x1], 1)
    for ptile in range(n_ptiles):
        grow0 = ptile * RT
        b0, y0 = grow0 // H, grow0 % H
        for cotile in range(CoutP // BCO):
            co0 = cotile * BCO
            sA = np.zeros(CK * PS)
            sW = np.zeros(CK * KK * BCO)
            W4 = W // 4
            rows_l = rpi + 2 * HALO
            per_cin = nimg * rows_l * W4
            countA = CK * per_cin
            assert countA <= MAXA * 256
            acc = np.zeros((4, COT, PXT, 16, 64))
            for wave in range(4):            # accumulators start at bias (+ residual); SPLIT adds them in the epilogue
                if SPLIT:
                    continue
                wpx0 = 0 if SPLIT else wave * PXT * 32
                for lane in range(64):
                    l31, half = lane & 31, lane >> 5
                    for pt in range(PXT):
                        m = wpx0 + pt * 32 + l31
                        rowt = m // W
                        c = m - rowt * W
                        img = rowt // rpi
                        r = rowt - img * rpi
                        b = b0 + img
                        for ct in range(COT):
                            for rg in range(16):
                                co = co0 + ct * 32 + (rg & 3) + 8 * (rg >> 2) + 4 * half
                                if co < Cout:
                                    v = bias[co]
                                    if res is not None and b < B:
                                        v += res[b, co, y0 + r, c]
                                    acc[wave, ct, pt, rg, lane] = v
            for ch in range(CinP // CK):
                cbase = ch * CK
                for tid in range(256):
                    for s in range(MAXA):
                        e = s * 256 + tid
                        if e >= countA:
                            continue
                        cin_l = e // per_cin
                        rem = e - cin_l * per_cin
                        rowi = rem // W4
                        c4 = rem - rowi * W4
                        img = rowi // rows_l
                        rl = rowi - img * rows_l
                        yy = y0 + rl - HALO
                        inimg = 0 <= yy < H and (b0 + img) < B
                        lds = cin_l * PS + img * IS + rl * P + (4 if HALO else 0) + c4 * 4
                        c = cbase + cin_l
                        v = np.zeros(4)
                        if inimg and c < Cin:
                            v = src_all[b0 + img, c].reshape(-1)[yy * W + c4 * 4: yy * W + c4 * 4 + 4].astype(np.float64)
                            if coef is not None:
                                v = v * coef[b0 + img, c, 0] + coef[b0 + img, c, 1]
                            if act:
                                v = silu(v)
                        sA[lds:lds + 4] = v
                    for s in range(MAXW):
                        e = s * 256 + tid
                        if e >= WCOUNT:
                            continue
                        row = e // (BCO // 4)
                        c4 = e - row * (BCO // 4)
                        goff = cbase * KK * CoutP + row * CoutP + co0 + c4 * 4
                        sW[e * 4:e * 4 + 4] = wp[goff:goff + 4]
                for wave in range(4):
                    wpx0 = 0 if SPLIT else wave * PXT * 32
                    NKP = CK // 2
                    for tap in range(KK):
                        tapoff = ((tap // 3) - 1) * P + ((tap % 3) - 1) if HALO else 0
                        for kq in range(NKP // 4 if SPLIT else NKP):
                            kp = wave + 4 * kq if SPLIT else kq
                            aw = np.zeros((COT, 64))
                            bx = np.zeros((PXT, 64))
                            for lane in range(64):
                                l31, half = lane & 31, lane >> 5
                                for ct in range(COT):
                                    aw[ct, lane] = sW[(2 * kp * KK + tap) * BCO + ct * 32 + half * KK * BCO + l31]
                                for pt in range(PXT):
                                    m = wpx0 + pt * 32 + l31
                                    rowt = m // W
                                    c = m - rowt * W
                                    img = rowt // rpi
                                    r = rowt - img * rpi
                                    pixoff = img * IS + (r + HALO) * P + (4 if HALO else 0) + c + half * PS
                                    bx[pt, lane] = sA[2 * kp * PS + pixoff + tapoff]
                            for ct in range(COT):
                                for pt in range(PXT):
                                    mfma_32x32x2(aw[ct], bx[pt], acc[wave, ct, pt])
            if SPLIT:
                acc[0] = acc.sum(0)
            for wave in range(1 if SPLIT else 4):
                wpx0 = 0 if SPLIT else wave * PXT * 32
                for lane in range(64):
                    l31, half = lane & 31, lane >> 5
                    for pt in range(PXT):
                        m = wpx0 + pt * 32 + l31
                        rowt = m // W
                        c = m - rowt * W
                        img = rowt // rpi
                        r = rowt - img * rpi
                        b = b0 + img
                        if b >= B:
                            continue
                        for ct in range(COT):
                            for rg in range(16):
                                co = co0 + ct * 32 + (rg & 3) + 8 * (rg >> 2) + 4 * half
                                if co < Cout:
                                    v = acc[wave, ct, pt, rg, lane]
                                    if SPLIT:
                                        v += bias[co] + (res[b, co, y0 + r, c] if res is not None else 0.0)
                                    y[b, co, y0 + r, c] = v * scale
    return y


def pack_weight(w, CinP, CoutP):
    Cout, Cin, ks, _ = w.shape
    KK = ks * ks
    wp = np.zeros(CinP * KK * CoutP)
    for co in range(Cout):
        for ci in range(Cin):
            for t in range(KK):
                wp[(ci * KK + t) * CoutP + co] = w[co, ci, t // ks, t % ks]
    return wp


def attn_emulate(qkv, heads):
    B, C3, S = qkv.shape
    C = C3 // 3
    D = C // heads
    DT = D // 32
    VP = 33
    scale = float(D) ** -0.5
    out = np.zeros((B, C, S))
    for bh in range(B * heads):
        b, hd = bh // heads, bh % heads
        qb = qkv[b, hd * D:(hd + 1) * D]
        kb = qkv[b, C + hd * D:C + (hd + 1) * D]
        vb = qkv[b, 2 * C + hd * D:2 * C + (hd + 1) * D]
        for blk in range((S + 127) // 128):
            for wave in range(4):
                q0 = blk * 128 + wave * 32
                if q0 >= S:
                    continue
                qreg = np.zeros((D // 2, 64))
                for lane in range(64):
                    for s in range(D // 2):
                        qreg[s, lane] = qb[2 * s + (lane >> 5), q0 + (lane & 31)]
                o = np.zeros((DT, 16, 64))
                m_run = np.full(64, -1e30)
                l_run = np.zeros(64)
                for t in range(S // 32):
                    sK = np.zeros(D * 32)
                    sV = np.zeros(D * VP)
                    for tid in range(256):
                        for i in range(D * 8 // 256):
                            e = i * 256 + tid
                            row, c4 = e >> 3, e & 7
                            sK[e * 4:e * 4 + 4] = kb[row, t * 32 + c4 * 4:t * 32 + c4 * 4 + 4]
                            sV[row * VP + c4 * 4:row * VP + c4 * 4 + 4] = vb[row, t * 32 + c4 * 4:t * 32 + c4 * 4 + 4]
                    st = np.zeros((16, 64))
                    for s in range(D // 2):
                        a = np.array([sK[(2 * s + (l >> 5)) * 32 + (l & 31)] for l in range(64)])
                        mfma_32x32x2(a, qreg[s], st)
                    st *= scale
                    mt = st.max(0)
                    mt = np.maximum(mt, mt[np.arange(64) ^ 32])
                    m_new = np.maximum(m_run, mt)
                    alpha = np.exp(m_run - m_new)
                    st = np.exp(st - m_new[None, :])
                    l_run = l_run * alpha + st.sum(0)
                    m_run = m_new
                    o *= alpha[None, None, :]
                    for s in range(16):
                        for ct in range(DT):
                            a = np.array([sV[(ct * 32 + (l & 31)) * VP + (s & 3) + 8 * (s >> 2) + 4 * (l >> 5)] for l in range(64)])
                            mfma_32x32x2(a, st[s], o[ct])
                l_tot = l_run + l_run[np.arange(64) ^ 32]
                for lane in range(64):
                    for ct in range(DT):
                        for r in range(16):
                            c = ct * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)
                            out[b, hd * D + c, q0 + (lane & 31)] = o[ct, r, lane] / l_tot[lane]
    return out

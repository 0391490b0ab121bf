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


# ---------------------------------------------------------------------------------------------------------------
def pack_wino_weight(w, CinP, CoutP, COT):
    """Mirror of pack_wino_weight_kernel (conv_wino.cpp): operand-major layout
    up[((((cotile*nunits + ci//8)*16 + xi)*COT + idx//4)*64 + lane)*4 + idx%4], idx = ((ci%8)//2)*COT + (co%BCO)//32,
    lane = (ci%2)*32 + co%32."""
    Cout, Cin = w.shape[:2]
    G = np.array([[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]], np.float64)
    BCO, nunits = 32 * COT, CinP // 8
    up = np.zeros(CinP * 16 * CoutP)
    for co in range(Cout):
        for ci in range(Cin):
            U = G @ w[co, ci].astype(np.float64) @ G.T
            cotile, ct, lane = co // BCO, (co % BCO) // 32, (ci & 1) * 32 + (co & 31)
            idx = ((ci & 7) >> 1) * COT + ct
            for xi in range(16):
                up[((((cotile * nunits + (ci >> 3)) * 16 + xi) * COT + (idx >> 2)) * 64 + lane) * 4 + (idx & 3)] = U[xi // 4, xi % 4]
    return up


def wino_emulate(x0, x1, up, bias, coef, act, res, scale, Cout, CoutP, CinP, COT):
    """Lane-level emulation of conv_wino_kernel's addressing (conv_wino.cpp): 1024 threads, 16-channel chunks, A operands
    straight from the operand-major packed weights, patch slots, two transform tasks per thread, block id -> (region, cout tile)."""
    B, C0, H, W = x0.shape
    C1 = 0 if x1 is None else x1.shape[1]
    Cin = C0 + C1
    CK, T, BCO, NT, PP = 16, 32, 32 * COT, 1024, 24
    VSZ, PSZ = CK * 16 * T, CK * 10 * PP
    assert CinP % CK == 0 and (C1 == 0 or C0 % CK == 0)
    HW = H * W
    x0f = x0.reshape(-1)
    x1f = None if x1 is None else x1.reshape(-1)
    coef_f = None if coef is None else coef.reshape(-1)
    y = np.full((B, Cout, H, W), np.nan)
    silu = lambda v: v / (1 + np.exp(-v))
    G8 = H == 8 and W == 8                     # two whole 8x8 images per region, halo = zero padding, only interiors loaded
    rx_n, ry_n = (1, 1) if G8 else (W // 16, H // 8)
    nreg, nct, nunits = ((B + 1) // 2 if G8 else B * rx_n * ry_n), CoutP // BCO, CinP // 8
    PCOUNT = CK * 2 * 64 if G8 else CK * 180
    for bid in range(((nreg + 7) // 8) * 8 * nct):
        xcd, slot = bid & 7, bid >> 3
        reg_id = (slot // nct) * 8 + xcd
        if reg_id >= nreg:
            continue
        b = 2 * reg_id if G8 else reg_id // (rx_n * ry_n)
        rr = 0 if G8 else reg_id - b * (rx_n * ry_n)
        oy0, ox0 = (rr // rx_n) * 8, (rr % rx_n) * 16
        cotile = slot - (slot // nct) * nct
        co0 = cotile * BCO
        acc = np.zeros((16, COT, 16, 64))              # wave, ct, reg, lane
        for ch in range(CinP // CK):
            sV = np.full(VSZ, np.nan)
            sP = np.full(PSZ + 4, np.nan)
            if G8:
                sP[:] = 0.0                                           # halo zeroed once in the prologue
            for tid in range(NT):
                for sl in range((PCOUNT + NT - 1) // NT):
                    e = sl * NT + tid
                    p_img = 0
                    if G8:
                        ci, img, r, cc_ = e >> 7, (e >> 6) & 1, (e >> 3) & 7, e & 7
                        valid = b + img < B
                        p_lds = ci * 10 * PP + (r + 1) * PP + img * 10 + cc_ + 1
                        goff, p_img, p_ci = r * 8 + cc_, (img if valid else 0), ci + (0 if valid else CK)
                    elif e >= PCOUNT:
                        sP[PSZ] = 0.0
                        continue
                    else:
                        ci, rem = e // 180, e % 180
                        r, cc_ = rem // 18, rem % 18
                        yy, xx = oy0 - 1 + r, ox0 - 1 + cc_
                        inside = 0 <= yy < H and 0 <= xx < W
                        p_ci = ci + (0 if inside else CK)
                        goff = min(max(yy, 0), H - 1) * W + min(max(xx, 0), W - 1)
                        p_lds = ci * 10 * PP + r * PP + cc_
                    cb = min(ch * CK, Cin - 1)
                    cmax = Cin - 1 - cb
                    second = cb >= C0
                    srcb = (x1f, (b * C1 + (cb - C0)) * HW) if second else (x0f, (b * C0 + cb) * HW)
                    istride = (C1 if second else C0) * HW
                    cl = min(p_ci & (CK - 1), cmax)
                    v = srcb[0][srcb[1] + cl * HW + goff + p_img * istride]
                    if coef is not None:
                        cch = min(ch * CK + (p_ci & (CK - 1)), Cin - 1)
                        v = v * coef_f[((b + p_img) * Cin + cch) * 2] + coef_f[((b + p_img) * Cin + cch) * 2 + 1]
                    if act:
                        v = silu(v)
                    nvalid = Cin - ch * CK
                    sP[p_lds] = v if p_ci < min(nvalid, CK) else 0.0
            for tid in range(NT):
                s_ci, s_tile, grp = (tid & 255) >> 5, tid & 31, tid >> 8
                s_ty, s_tx = ((s_tile >> 2) & 3, (s_tile & 3) + 5 * (s_tile >> 4)) if G8 else (s_tile >> 3, s_tile & 7)
                p_rd = s_ci * 10 * PP + 2 * s_ty * PP + 2 * s_tx
                p_rdA, p_rdB = p_rd + (0 if grp == 0 else 1) * PP, p_rd + (3 if grp == 3 else 2) * PP
                v_fa, v_fb = (-1.0 if grp == 2 else 1.0), (1.0 if grp in (1, 2) else -1.0)
                v_wr = s_ci * 16 * T + grp * 4 * T + s_tile
                for k2 in range(2):
                    m = [v_fb * sP[p_rdB + k2 * 8 * 10 * PP + j] + v_fa * sP[p_rdA + k2 * 8 * 10 * PP + j] for j in range(4)]
                    base = v_wr + k2 * 8 * 16 * T
                    sV[base + 0 * T] = m[0] - m[2]
                    sV[base + 1 * T] = m[1] + m[2]
                    sV[base + 2 * T] = m[2] - m[1]
                    sV[base + 3 * T] = m[1] - m[3]
            assert not np.isnan(sV).any()
            for wave in range(16):
                for g in range(8):
                    bv = np.array([sV[((2 * g + (l >> 5)) * 16 + wave) * T + (l & 31)] for l in range(64)])
                    u, kp = 2 * ch + (g >> 2), g & 3
                    for ct in range(COT):
                        idx = kp * COT + ct
                        av = np.array([up[((((cotile * nunits + u) * 16 + wave) * COT + (idx >> 2)) * 64 + l) * 4 + (idx & 3)]
                                       for l in range(64)])
                        mfma_32x32x2(av, bv, acc[wave, ct])
        for ct in range(COT):
            sM = np.full(16 * 32 * T, np.nan)
            for wave in range(16):
                for lane in range(64):
                    for r in range(16):
                        col = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)
                        sM[(wave * 32 + col) * T + (lane & 31)] = acc[wave, ct, r, lane]
            for tid in range(NT):
                e_tile, col = tid & 31, tid >> 5
                e_ty, e_tx = ((e_tile >> 2) & 3, e_tile & 3) if G8 else (e_tile >> 3, e_tile & 7)
                e_b = b + (e_tile >> 4 if G8 else 0)
                co = co0 + ct * 32 + col
                mm = np.array([sM[(xi * 32 + col) * T + e_tile] for xi in range(16)])
                t0 = [mm[0 + l] + mm[4 + l] + mm[8 + l] for l in range(4)]
                t1 = [mm[4 + l] - mm[8 + l] - mm[12 + l] for l in range(4)]
                ys = [[t0[0] + t0[1] + t0[2], t0[1] - t0[2] - t0[3]], [t1[0] + t1[1] + t1[2], t1[1] - t1[2] - t1[3]]]
                if co < Cout and e_b < B:
                    for r in range(2):
                        for cc in range(2):
                            yy, xx = oy0 + 2 * e_ty + r, ox0 + 2 * e_tx + cc
                            v = ys[r][cc] + bias[co]
                            if res is not None:
                                v += res[e_b, co, yy, xx]
                            assert np.isnan(y[e_b, co, yy, xx])
                            y[e_b, co, yy, xx] = v * scale
    assert not np.isnan(y).any()
    return y


def gemm1x1_emulate(x0, x1, wp, bias, coef, act, res, scale, Cout, CoutP, CinP, COT, CK=16):
    """Lane-level emulation of conv1x1_dma_kernel's addressing (conv1x1_dma.cpp): DMA piece maps of W and x, the
    coefficient table, MFMA lane maps, block id -> (pixel tile, cout tile), ragged last pixel tile."""
    B, C0, H, W = x0.shape
    C1 = 0 if x1 is None else x1.shape[1]
    Cin = C0 + C1
    PT, BCO, MAXIMG = 128, 32 * COT, 4
    HW = H * W
    NPX = B * HW
    assert Cin % CK == 0 and CinP % CK == 0 and (C1 == 0 or C0 % CK == 0) and HW % 32 == 0
    assert HW % PT == 0 or (PT % HW == 0 and PT // HW <= MAXIMG)
    WSZ, XSZ = CK * BCO, CK * PT
    WP, XP = WSZ // 4, XSZ // 4
    MAXW, MAXX = (WP + 255) // 256, XP // 256
    x0f = x0.reshape(-1)
    x1f = None if x1 is None else x1.reshape(-1)
    wpf = wp.reshape(-1)
    coef_f = None if coef is None else coef.reshape(-1)
    resf = None if res is None else res.reshape(-1)
    yf = np.full(B * Cout * HW, np.nan)
    silu = lambda v: v / (1 + np.exp(-v))
    ptiles = (NPX + PT - 1) // PT
    nct = CoutP // BCO
    for bid in range(((ptiles + 7) // 8) * 8 * nct):
        xcd, slot = bid & 7, bid >> 3
        ptile, ctile = (slot // nct) * 8 + xcd, slot % nct
        if ptile >= ptiles:
            continue
        co0 = ctile * BCO
        gp0 = ptile * PT
        acc = np.zeros((4, COT, 16, 64))
        b_first = gp0 // HW
        nimg = 1 if HW >= PT else PT // HW
        for ch in range(Cin // CK):
            cb = ch * CK
            sW = np.full(WSZ, np.nan)
            sX = np.full(XSZ, np.nan)
            sC = np.full(MAXIMG * CK * 2, np.nan)
            for tid in range(256):
                wave, lane = tid >> 6, tid & 63
                for s in range(MAXW):
                    if not (MAXW * 256 == WP or s * 256 + wave * 64 < WP):
                        continue
                    e = min(s * 256 + tid, WP - 1)
                    row, c4 = e // (BCO // 4), e % (BCO // 4)
                    g = cb * CoutP + row * CoutP + co0 + c4 * 4
                    dst = (s * 256 + wave * 64) * 4 + lane * 4
                    assert 0 <= g and g + 4 <= wpf.size and dst + 4 <= WSZ
                    sW[dst:dst + 4] = wpf[g:g + 4]
                xg = min(gp0 + (tid & 31) * 4, NPX - 4)
                xb, xp = xg // HW, xg % HW
                x_ci = tid >> 5
                second = cb >= C0
                src = x1f if second else x0f
                base = (cb - C0) * HW if second else cb * HW
                voff = ((xb * C1 + x_ci) * HW + xp) if second else ((xb * C0 + x_ci) * HW + xp)
                for s in range(MAXX):
                    g = base + s * 8 * HW + voff
                    dst = (s * 256 + wave * 64) * 4 + lane * 4
                    assert 0 <= g and g + 4 <= src.size and dst + 4 <= XSZ
                    sX[dst:dst + 4] = src[g:g + 4]
                if coef is not None and tid < nimg * CK:
                    c_img, c_ci = min(b_first + tid // CK, B - 1), tid % CK
                    o = (c_img * Cin + cb + c_ci) * 2
                    sC[tid * 2:tid * 2 + 2] = coef_f[o:o + 2]
            assert not np.isnan(sW).any() and not np.isnan(sX).any()
            for wave in range(4):
                my_img = 0 if HW >= PT else (wave * 32) // HW
                for kp in range(CK // 2):
                    bv = np.zeros(64)
                    for l in range(64):
                        row = 2 * kp + (l >> 5)
                        v = sX[row * PT + wave * 32 + (l & 31)]
                        if coef is not None:
                            v = v * sC[(my_img * CK + row) * 2] + sC[(my_img * CK + row) * 2 + 1]
                            if act:
                                v = silu(v)
                        bv[l] = v
                    for ct in range(COT):
                        av = np.array([sW[(2 * kp + (l >> 5)) * BCO + ct * 32 + (l & 31)] for l in range(64)])
                        mfma_32x32x2(av, bv, acc[wave, ct])
        for wave in range(4):
            for lane in range(64):
                gp = gp0 + wave * 32 + (lane & 31)
                if gp >= NPX:
                    continue
                ob, op = gp // HW, gp % HW
                obase = ob * Cout * HW + op
                for ct in range(COT):
                    for r in range(16):
                        co = co0 + ct * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)
                        if co < Cout:
                            v = acc[wave, ct, r, lane] + bias[co]
                            if resf is not None:
                                v += resf[obase + co * HW]
                            assert np.isnan(yf[obase + co * HW])        # every output written exactly once
                            yf[obase + co * HW] = v * scale
    assert not np.isnan(yf).any()
    return yf.reshape(B, Cout, H, W)

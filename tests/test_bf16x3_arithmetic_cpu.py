"""CPU: the arithmetic of the three-piece bf16 kernels (conv_wino3.cpp / conv1x1_h2.cpp / attention_h2.cpp with NP = 3: the DEFAULT
arithmetic of the library) restated in numpy.

Operands are split  v = v1 + v2 + v3,  v1 = bf16(v), v2 = bf16(v - v1), v3 = bf16(v - v1 - v2)  (round to nearest even at every level);
the product is accumulated in fp32 from the six piece products u1 v3 + u3 v1 + u2 v2 + u1 v2 + u2 v1 + u1 v1 (u2 v3, u3 v2, u3 v3
dropped).  The claims the kernels' headers make, checked here without a GPU:
  * the split is EXACT: v1 + v2 + v3 == v bit for bit for 2^-100 <= |v| < 2^127 (1e-30 ... 1e30 here), and both remainders are
    exactly representable (the subtractions the kernels do in fp32 lose nothing); below 2^-110 the third piece reaches the bf16
    denormals (ulp 2^-133) and the split loses bits gradually -- still within 2^-16 relative at 1e-38;
  * every piece product is exact in fp32 (8 x 8 significant bits) and the dropped terms are below 2^-23.4 of the product;
  * a K-deep dot product computed this way is within 1.2x of an fp32 FMA chain's distance from the fp64 result -- on Gaussian data
    AND on structured data (constant operands, one dominant term, sums that cancel), where the two-piece fp16 form is measurably
    worse (its 2^-22 operand error does not average out there);
  * nothing depends on the magnitude of the operands (no scale, no clamp).
"""
import numpy as np
import pytest


def bf16_rne(v):
    """fp32 array -> the nearest bf16 value (ties to even), as fp32."""
    u = np.ascontiguousarray(v, dtype=np.float32).view(np.uint32).astype(np.uint64)
    u = (u + 0x7fff + ((u >> 16) & 1)) & 0xffff0000
    return u.astype(np.uint32).view(np.float32)


def split3(v):
    v = v.astype(np.float32)
    v1 = bf16_rne(v)
    r1 = (v - v1).astype(np.float32)
    v2 = bf16_rne(r1)
    r2 = (r1 - v2).astype(np.float32)
    v3 = bf16_rne(r2)
    return v1, v2, v3, r1, r2


PRODUCTS = ((0, 2), (2, 0), (1, 1), (0, 1), (1, 0), (0, 0))        # pieces.h: (weight piece, activation piece), smallest product first


def dot_bf16x3(u, v, step=16):
    """sum_k u_k v_k along the last axis with three-piece operands: per MFMA K step the 16 products of one piece pair are summed exactly
    and added to the fp32 accumulator once."""
    up, vp = split3(u)[:3], split3(v)[:3]
    acc = np.zeros(u.shape[:-1], np.float32)
    for k0 in range(0, u.shape[-1], step):
        sl = slice(k0, k0 + step)
        for a, b in PRODUCTS:
            acc = (acc + np.sum(up[a][..., sl].astype(np.float64) * vp[b][..., sl].astype(np.float64), axis=-1).astype(np.float32)).astype(np.float32)
    return acc


def dot_f32_fma(u, v):
    """The fp32 MFMA / FMA chain: exact products, one rounding per accumulation."""
    acc = np.zeros(u.shape[:-1], np.float64)
    for k in range(u.shape[-1]):
        acc = (acc + u[..., k].astype(np.float64) * v[..., k].astype(np.float64)).astype(np.float32).astype(np.float64)
    return acc.astype(np.float32)


def dot_f16x2(u, v, step=16):
    def split2(x):
        x1 = x.astype(np.float16).astype(np.float32)
        return x1, (x - x1).astype(np.float16).astype(np.float32)
    u1, u2 = split2(u)
    v1, v2 = split2(v)
    acc = np.zeros(u.shape[:-1], np.float32)
    for k0 in range(0, u.shape[-1], step):
        sl = slice(k0, k0 + step)
        for a, b in ((u1, v2), (u2, v1), (u1, v1)):
            acc = (acc + np.sum(a[..., sl].astype(np.float64) * b[..., sl].astype(np.float64), axis=-1).astype(np.float32)).astype(np.float32)
    return acc


def test_split_is_exact_over_the_fp32_range():
    rng = np.random.default_rng(0)
    v = (rng.standard_normal(400000) * 10.0 ** rng.uniform(-30, 30, 400000)).astype(np.float32)
    v = v[np.abs(v) >= 2.0 ** -100]
    v1, v2, v3, r1, r2 = split3(v)
    assert np.array_equal(v1.astype(np.float64) + v2.astype(np.float64) + v3.astype(np.float64), v.astype(np.float64))
    # the fp32 subtractions the kernels perform are exact
    assert np.array_equal(r1.astype(np.float64), v.astype(np.float64) - v1.astype(np.float64))
    assert np.array_equal(r2.astype(np.float64), r1.astype(np.float64) - v2.astype(np.float64))
    # piece magnitudes: |v2| <= 2^-8 |v|, |v3| <= 2^-16 |v|  (what makes the dropped products small)
    nz = v != 0
    assert (np.abs(v2[nz]) <= 2.0 ** -8 * np.abs(v[nz])).all() and (np.abs(v3[nz]) <= 2.0 ** -16 * np.abs(v[nz])).all()


def test_split_degrades_gradually_towards_the_denormals():
    """|v| < 2^-110: v's last bits fall below the bf16 denormal ulp (2^-133); the error is that ulp, nothing worse."""
    rng = np.random.default_rng(4)
    v = (rng.standard_normal(100000) * 10.0 ** rng.uniform(-37.5, -30, 100000)).astype(np.float32)
    v1, v2, v3, _, _ = split3(v)
    err = np.abs(v.astype(np.float64) - v1.astype(np.float64) - v2.astype(np.float64) - v3.astype(np.float64))
    assert (err <= 2.0 ** -134).all()                     # half a bf16-denormal ulp


def test_piece_products_are_exact_and_the_dropped_terms_small():
    rng = np.random.default_rng(1)
    u = (rng.standard_normal(100000) * 10.0 ** rng.uniform(-6, 6, 100000)).astype(np.float32)
    v = (rng.standard_normal(100000) * 10.0 ** rng.uniform(-6, 6, 100000)).astype(np.float32)
    up, vp = split3(u)[:3], split3(v)[:3]
    kept = np.zeros(u.shape, np.float64)
    for a, b in PRODUCTS:
        p64 = up[a].astype(np.float64) * vp[b].astype(np.float64)
        assert np.array_equal(p64, (up[a] * vp[b]).astype(np.float64))          # exact in fp32: 8 x 8 significant bits
        kept += p64
    exact = u.astype(np.float64) * v.astype(np.float64)
    nz = exact != 0
    assert (np.abs(exact - kept)[nz] <= 2.0 ** -23.4 * np.abs(exact)[nz]).all()


def _data(kind, rng, n, K):
    u = rng.standard_normal((n, K)).astype(np.float32)
    v = rng.standard_normal((n, K)).astype(np.float32)
    if kind == "const":                          # constant operands: every product carries the same representation error
        u[:] = rng.standard_normal((n, 1)).astype(np.float32)
        v[:] = rng.standard_normal((n, 1)).astype(np.float32)
    elif kind == "dominant":
        v[:, 3] *= 1.0e4
    elif kind == "cancel":                       # pairs cancel to 2^-12 of their terms
        v[:, 1::2] = v[:, 0::2]
        u[:, 1::2] = -u[:, 0::2] * np.float32(1.0 + 2.0 ** -12)
    return u, v


@pytest.mark.parametrize("K", [96 * 16, 480 * 16, 192])
@pytest.mark.parametrize("kind", ["gauss", "const", "dominant", "cancel"])
@pytest.mark.parametrize("mag", [1.0, 1e-6, 1e5])
def test_dot_product_is_fp32_equivalent(K, kind, mag):
    rng = np.random.default_rng(2)
    u, v = _data(kind, rng, 256, K)
    v = (v * np.float32(mag)).astype(np.float32)
    want = np.sum(u.astype(np.float64) * v.astype(np.float64), axis=-1)
    scale = np.sum(np.abs(u.astype(np.float64) * v.astype(np.float64)), axis=-1)          # what fp32 rounding errors are proportional to
    e3 = np.abs(dot_bf16x3(u, v) - want) / scale
    e32 = np.abs(dot_f32_fma(u, v) - want) / scale
    # same error class as an fp32 chain: within 1.2x of its worst element (+ one ulp of slack); for data whose accumulation roundings
    # average out (everything but the all-equal-terms case, where any fp32 accumulator drifts ~K/16 * 2^-24) below sqrt(K) * 2^-24
    assert e3.max() <= 1.2 * e32.max() + 2.0 ** -24, (e3.max(), e32.max())
    assert e3.max() < (K / 16 if kind == "const" else np.sqrt(K)) * 2.0 ** -24


def _representation_error(pieces_u, pieces_v, products, u, v):
    """|sum of the kept piece products - u v| summed over K in fp64 (no accumulator rounding): what the operand form alone costs."""
    kept = np.zeros(u.shape, np.float64)
    for a, b in products:
        kept += pieces_u[a].astype(np.float64) * pieces_v[b].astype(np.float64)
    exact = u.astype(np.float64) * v.astype(np.float64)
    return np.abs(np.sum(kept - exact, axis=-1)) / np.sum(np.abs(exact), axis=-1)


@pytest.mark.parametrize("kind", ["const", "cancel"])
def test_two_piece_fp16_is_measurably_worse_on_structured_data(kind):
    """Why the two-piece fp16 form is an option and not the default: on constant operands its representation error (operand bits
    beyond 22 + the dropped u2 v2) is the same for every term and adds up coherently; the three-piece bf16 form drops only
    u2 v3 + u3 v2 + u3 v3, an order of magnitude less."""
    rng = np.random.default_rng(3)
    u, v = _data(kind, rng, 256, 96 * 16)

    def split2(x):
        x1 = x.astype(np.float16).astype(np.float32)
        return x1, (x - x1).astype(np.float16).astype(np.float32)
    s = np.float32(16.0)
    e2 = _representation_error(split2(u * s), split2(v * s), ((0, 1), (1, 0), (0, 0)), u * s, v * s)
    e3 = _representation_error(split3(u)[:3], split3(v)[:3], PRODUCTS, u, v)
    assert e3.max() < 2.0 ** -23.4
    assert np.median(e3) * 4 < np.median(e2) and e3.max() * 2 < e2.max(), (np.median(e3), np.median(e2), e3.max(), e2.max())


# ---- the lazy exponent reference of the three-piece attention kernel (attention_h2.cpp AH_SOFTMAX_LAZY), restated in fp32 numpy
def lazy_softmax_rows(scores2, v, lazy=40.0, tile=32):
    """scores2: [queries, keys] in the base-2 exponent domain (fp32), v: [keys, channels].  One row per 'lane': the reference m moves
    only when a tile's maximum exceeds it by more than `lazy`; O and l carry 2^(m_true - m) until the final division."""
    nq, nk = scores2.shape
    m = np.full(nq, -1e30, np.float32)
    l = np.zeros(nq, np.float32)
    o = np.zeros((nq, v.shape[1]), np.float32)
    rescales = 0
    for t in range(0, nk, tile):
        st = scores2[:, t:t + tile].astype(np.float32)
        mt = st.max(1)
        need = mt > m + np.float32(lazy)
        if need.any():
            rescales += 1
            m_new = np.where(need, mt, m).astype(np.float32)
            alpha = np.exp2((m - m_new).astype(np.float32)).astype(np.float32)
            assert np.all(alpha[~need] == 1.0)
            l = (l * alpha).astype(np.float32)
            o = (o * alpha[:, None]).astype(np.float32)
            m = m_new
        p = np.exp2((st - m[:, None]).astype(np.float32)).astype(np.float32)
        assert np.isfinite(p).all() and p.max() <= 2.0 ** (lazy + 1e-3)
        l = (l + p.sum(1, dtype=np.float32)).astype(np.float32)
        o = (o + p @ v[t:t + tile].astype(np.float32)).astype(np.float32)
    return o / l[:, None], rescales


@pytest.mark.parametrize("pattern", ["ramp_up", "ramp_down", "spike_late", "flat", "huge_first", "gauss"])
def test_lazy_softmax_reference_is_softmax(pattern):
    """The recurrence keeps  O / l == softmax(s) V  whatever the reference does: the factor 2^(m_true - m) is common to O and l.  The
    global maximum's own term is >= 1 (the reference never exceeds a tile maximum), so l cannot underflow; probabilities stay below
    2^40 (finite in fp32, exactly splittable into three bf16 pieces); and the rescale fires on the first tile and then only when the
    scores really move (ramp: every second tile; Gaussian scores: once)."""
    rng = np.random.default_rng(5)
    nq, nk, D = 32, 1024, 16
    tile = np.arange(nk) // 32
    if pattern == "ramp_up":
        prof = 30.3 * tile + rng.standard_normal(nk)
    elif pattern == "ramp_down":
        prof = -30.3 * tile + rng.standard_normal(nk)
    elif pattern == "spike_late":
        prof = rng.standard_normal(nk); prof[nk - 3] = 430.0
    elif pattern == "flat":
        prof = np.full(nk, 7.0)
    elif pattern == "huge_first":
        prof = rng.standard_normal(nk); prof[:32] += 290.0
    else:
        prof = 3.0 * rng.standard_normal(nk)
    s2 = (prof[None, :] + 0.3 * rng.standard_normal((nq, nk))).astype(np.float32)
    v = rng.standard_normal((nk, D)).astype(np.float32)
    got, rescales = lazy_softmax_rows(s2, v)
    s64 = s2.astype(np.float64) * np.log(2.0)
    w = np.exp(s64 - s64.max(1, keepdims=True)); w /= w.sum(1, keepdims=True)
    want = w @ v.astype(np.float64)
    assert np.isfinite(got).all()
    np.testing.assert_allclose(got, want, rtol=2e-5, atol=2e-6)
    expect = {"ramp_up": (12, 32), "ramp_down": (1, 1), "spike_late": (2, 2), "flat": (1, 1), "huge_first": (1, 1), "gauss": (1, 1)}[pattern]
    assert expect[0] <= rescales <= expect[1], rescales

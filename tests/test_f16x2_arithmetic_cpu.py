"""CPU: the arithmetic of the two-piece fp16 kernels (conv_wino2h.cpp / conv1x1_h2.cpp / attention_h2.cpp) restated in numpy.

Operands are split  v ~= v1 + v2,  v1 = fp16(v), v2 = fp16(v - v1)  (round to nearest even at both levels), the product is
u1 v2 + u2 v1 + u1 v1 (u2 v2 dropped), every piece product is exact in fp32 and the sum is accumulated in fp32.  The claims the
kernels' headers make, checked here without a GPU:
  * the split leaves at most 2^-22 of an operand behind (after the power-of-two scaling that keeps the pieces in the fp16 range);
  * a K-deep dot product computed this way is as close to the fp64 result as an fp32 dot product is (the operand error is random in
    sign and averages out, the fp32 accumulation error is common to both);
  * the per-layer weight scale 2^e (max |U| at 2^13..2^14) and the activation scale 2^4 are exact and make the result independent of
    the magnitudes of weights and inputs.
"""
import numpy as np
import pytest


def split2(v):
    """v (fp32 array) -> (v1, v2) as fp32 arrays holding fp16-representable values."""
    v = np.clip(v.astype(np.float32), -65504.0, 65504.0)
    v1 = v.astype(np.float16).astype(np.float32)
    v2 = (v - v1).astype(np.float16).astype(np.float32)          # v - v1 is exact in fp32
    return v1, v2


def dot_f16x2(u, v, step=16):
    """sum_k u_k v_k along the last axis with two-piece operands, fp32 accumulation in chunks of `step` (one MFMA K step: its
    16 products are summed exactly, then added to the fp32 accumulator once)."""
    u1, u2 = split2(u)
    v1, v2 = split2(v)
    acc = np.zeros(u.shape[:-1], np.float32)
    for k0 in range(0, u.shape[-1], step):
        sl = slice(k0, k0 + step)
        for a, b in ((u1, v2), (u2, v1), (u1, v1)):
            acc = (acc + np.sum(a[..., sl].astype(np.float64) * b[..., sl].astype(np.float64), axis=-1).astype(np.float32)).astype(np.float32)
    return acc


def dot_f32(u, v):
    acc = np.zeros(u.shape[:-1], np.float32)
    for k in range(u.shape[-1]):
        acc = (acc + (u[..., k] * v[..., k]).astype(np.float32)).astype(np.float32)      # one rounding per product and per add
    return acc


def weight_scale(w):
    """The pack kernels' rule: e with max|w| * 2^e in [2^13, 2^14)."""
    m, k = np.frexp(np.float32(np.abs(w).max()))
    return np.float32(2.0) ** (14 - int(k))


def test_split_residual_is_at_most_2_to_minus_22():
    rng = np.random.default_rng(0)
    v = (rng.standard_normal(200000) * 10.0 ** rng.uniform(-2, 3, 200000)).astype(np.float32)
    v = v[np.abs(v) > 2.0 ** -3]              # second piece in the normal fp16 range (the kernels scale their operands into it)
    v1, v2 = split2(v)
    res = np.abs(v.astype(np.float64) - v1 - v2)
    assert (res <= 2.0 ** -22 * np.abs(v)).all()
    assert np.array_equal((v - v1).astype(np.float64), v.astype(np.float64) - v1.astype(np.float64))      # the remainder is exact


@pytest.mark.parametrize("K", [96 * 9, 480 * 9, 192])
@pytest.mark.parametrize("wmag,xmag", [(1.0, 1.0), (1e-3, 30.0), (40.0, 0.02)])
def test_dot_product_error_matches_fp32(K, wmag, xmag):
    rng = np.random.default_rng(1)
    n = 4096
    w = (rng.standard_normal((n, K)) / np.sqrt(K) * wmag).astype(np.float32)
    x = (rng.standard_normal((n, K)) * xmag).astype(np.float32)
    exact = np.sum(w.astype(np.float64) * x.astype(np.float64), axis=-1)
    sw, sx = weight_scale(w), np.float32(16.0)
    got = dot_f16x2(w * sw, x * sx).astype(np.float64) / (float(sw) * float(sx))
    ref32 = dot_f32(w, x).astype(np.float64)
    scale = np.abs(exact).max()
    e_h2, e_32 = np.abs(got - exact).max() / scale, np.abs(ref32 - exact).max() / scale
    rms_h2, rms_32 = np.sqrt(np.mean((got - exact) ** 2)) / scale, np.sqrt(np.mean((ref32 - exact) ** 2)) / scale
    assert e_h2 <= 1.5 * e_32 + 1e-7, (e_h2, e_32)
    assert rms_h2 <= 1.2 * rms_32 + 2e-8, (rms_h2, rms_32)


def test_scales_are_exact_powers_of_two():
    rng = np.random.default_rng(2)
    w = (rng.standard_normal(1000) * 3e-3).astype(np.float32)
    s = weight_scale(w)
    assert np.log2(float(s)) == int(np.log2(float(s)))
    assert 2.0 ** 13 <= np.abs(w * s).max() < 2.0 ** 14
    assert np.array_equal((w * s) / s, w)                       # scaling and unscaling by a power of two is exact

"""CPU ORACLE -- TEST INFRASTRUCTURE ONLY.

A functional, state_dict-driven restatement (torch CPU, fp32) of the MCVD
`unetmore` score network forward.  Nothing in the product path
(`mcvd_pytorch_amd/`) may import this file; only `tests/`,
`__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg do.

Parity status: PINNED.  `oracle/gen_golden.py` runs the real reference
(`/root/reference`, imported read-only in the build container) on seeded
synthetic weights/inputs and commits the outputs under `tests/golden/`;
`tests/test_oracle_golden.py` checks this restatement against them.

Third-party arithmetic: the reference's heavy math is PyTorch ATen (torch
2.10.0+rocm7.0 CPU build: oneDNN conv, MKL bmm).  Here convolution goes through
`torch.nn.functional.conv2d` (same library, same published semantics:
cross-correlation, zero padding); everything else is spelt out with elementary
tensor ops.  `conv2d_unfold` restates conv as im2col + matmul and is checked
against F.conv2d in the tests.

Each function cites the reference lines it follows (paths relative to
/root/reference).
"""
import math
from types import SimpleNamespace

import numpy as np
import torch
import torch.nn.functional as F

RSQRT2_DIV = float(np.sqrt(2.0))  # reference divides by np.sqrt(2.) (layerspp.py:249,624)


# --------------------------------------------------------------------------- config
def cfg_get(ns, name, default=None):
    return getattr(ns, name, default)


def hot_cfg(config):
    """Pull the hot-path keys out of a reference-style nested Namespace
    (schema: main.py:359-367 dict2namespace of configs/*.yml)."""
    d, m = config.data, config.model
    c = SimpleNamespace()
    c.image_size = int(d.image_size)
    c.channels = int(d.channels)
    c.num_frames = int(d.num_frames)
    # ncsnpp_more.py:47 -- cond frames = past + future
    c.num_frames_cond = int(d.num_frames_cond) + int(cfg_get(d, "num_frames_future", 0))
    c.ngf = int(m.ngf)
    c.ch_mult = [int(v) for v in m.ch_mult]
    c.num_res_blocks = int(m.num_res_blocks)
    c.attn_resolutions = [int(v) for v in m.attn_resolutions]
    c.n_head_channels = int(cfg_get(m, "n_head_channels", -1))
    c.spade = bool(cfg_get(m, "spade", False))
    c.spade_dim = int(cfg_get(m, "spade_dim", 128))
    c.num_classes = int(m.num_classes)
    c.sigma_dist = cfg_get(m, "sigma_dist", "linear")
    c.sigma_begin = float(m.sigma_begin)
    c.sigma_end = float(m.sigma_end)
    # SURVEY 8f rank 4 flags
    c.cond_emb = bool(cfg_get(m, "cond_emb", False))                  # ncsnpp_more.py:61, :97-99, :282-286
    c.noise_in_cond = bool(cfg_get(m, "noise_in_cond", False))        # :751, :755-768
    c.gamma = bool(cfg_get(m, "gamma", False))                        # :744-749
    c.output_all_frames = bool(cfg_get(m, "output_all_frames", False))   # :384-385
    if cfg_get(m, "arch", "unetmore") != "unetmore":
        raise NotImplementedError("only arch=unetmore is on the hot path")
    return c


# --------------------------------------------------------------------------- topology
def gn_groups(ch):
    """layerspp.py:474-476 / :212-214 / :128-130."""
    g = min(ch // 4, 32)
    while ch % g != 0:
        g -= 1
    return g


def module_plan(c):
    """Module list in `all_modules` order (ncsnpp_more.py:70-249 concat mode,
    :435-588 SPADE mode).  Returns list of dict specs."""
    nf = c.ngf
    C = c.channels
    in0 = C * c.num_frames if c.spade else C * (c.num_frames + c.num_frames_cond)
    res = [c.image_size // (2 ** i) for i in range(len(c.ch_mult))]
    mods = [dict(kind="linear", cin=nf, cout=4 * nf),
            dict(kind="linear", cin=4 * nf, cout=4 * nf)]
    if getattr(c, "cond_emb", False):
        mods.append(dict(kind="embed", n=2, dim=nf // 2))          # torch.nn.Embedding(2, nf // 2), ncsnpp_more.py:98
    mods.append(dict(kind="conv3", cin=in0, cout=nf))
    hs_c = [nf]
    in_ch = nf
    L = len(c.ch_mult)
    for lv in range(L):
        for _ in range(c.num_res_blocks):
            out_ch = nf * c.ch_mult[lv]
            mods.append(dict(kind="res", cin=in_ch, cout=out_ch, up=False, down=False))
            in_ch = out_ch
            if res[lv] in c.attn_resolutions:
                mods.append(dict(kind="attn", ch=in_ch))
            hs_c.append(in_ch)
        if lv != L - 1:
            mods.append(dict(kind="res", cin=in_ch, cout=in_ch, up=False, down=True))
            hs_c.append(in_ch)
    in_ch = hs_c[-1]
    mods.append(dict(kind="res", cin=in_ch, cout=in_ch, up=False, down=False))
    mods.append(dict(kind="attn", ch=in_ch))
    mods.append(dict(kind="res", cin=in_ch, cout=in_ch, up=False, down=False))
    for lv in reversed(range(L)):
        for _ in range(c.num_res_blocks + 1):
            out_ch = nf * c.ch_mult[lv]
            mods.append(dict(kind="res", cin=in_ch + hs_c.pop(), cout=out_ch, up=False, down=False))
            in_ch = out_ch
        if res[lv] in c.attn_resolutions:
            mods.append(dict(kind="attn", ch=in_ch))
        if lv != 0:
            mods.append(dict(kind="res", cin=in_ch, cout=in_ch, up=True, down=False))
    assert not hs_c
    mods.append(dict(kind="norm", ch=in_ch))
    mods.append(dict(kind="conv3", cin=in_ch, cout=C * c.num_frames))
    return mods


def param_shapes(c):
    """name -> shape for every parameter of UNetMore_DDPM.state_dict()
    (checked against the live reference by gen_golden.py)."""
    temb = 4 * c.ngf + (c.ngf // 2 if getattr(c, "cond_emb", False) else 0)      # temb_dim, ncsnpp_more.py:95-99
    cond_ch = c.num_frames_cond * c.channels
    out = {}

    def actnorm(prefix, ch, emb):
        if emb:
            out[prefix + ".Dense_0.weight"] = (2 * ch, temb)
            out[prefix + ".Dense_0.bias"] = (2 * ch,)
        if c.spade:
            sd = c.spade_dim
            out[prefix + ".Norm_0.mlp_shared.0.weight"] = (sd, cond_ch, 3, 3)
            out[prefix + ".Norm_0.mlp_shared.0.bias"] = (sd,)
            out[prefix + ".Norm_0.mlp_gamma.weight"] = (ch, sd, 3, 3)
            out[prefix + ".Norm_0.mlp_gamma.bias"] = (ch,)
            out[prefix + ".Norm_0.mlp_beta.weight"] = (ch, sd, 3, 3)
            out[prefix + ".Norm_0.mlp_beta.bias"] = (ch,)
        elif not emb:
            out[prefix + ".Norm_0.weight"] = (ch,)
            out[prefix + ".Norm_0.bias"] = (ch,)

    for i, m in enumerate(module_plan(c)):
        p = f"unet.all_modules.{i}"
        k = m["kind"]
        if k == "linear":
            out[p + ".weight"] = (m["cout"], m["cin"])
            out[p + ".bias"] = (m["cout"],)
        elif k == "embed":
            out[p + ".weight"] = (m["n"], m["dim"])
        elif k == "conv3":
            out[p + ".weight"] = (m["cout"], m["cin"], 3, 3)
            out[p + ".bias"] = (m["cout"],)
        elif k == "res":
            ci, co = m["cin"], m["cout"]
            actnorm(p + ".actnorm0", ci, True)
            out[p + ".Conv_0.weight"] = (co, ci, 3, 3)
            out[p + ".Conv_0.bias"] = (co,)
            actnorm(p + ".actnorm1", co, True)
            out[p + ".Conv_1.weight"] = (co, co, 3, 3)
            out[p + ".Conv_1.bias"] = (co,)
            if ci != co or m["up"] or m["down"]:
                out[p + ".Conv_2.weight"] = (co, ci, 1, 1)
                out[p + ".Conv_2.bias"] = (co,)
        elif k == "attn":
            ch = m["ch"]
            out[p + ".GroupNorm_0.weight"] = (ch,)
            out[p + ".GroupNorm_0.bias"] = (ch,)
            for j in range(4):
                out[p + f".NIN_{j}.W"] = (ch, ch)
                out[p + f".NIN_{j}.b"] = (ch,)
        elif k == "norm":
            actnorm(p, m["ch"], False)
    return out


# --------------------------------------------------------------------------- schedule
def make_schedule(c):
    """models/__init__.py:16-35 + ncsnpp_more.py:735-743.  Returns (betas, alphas, alphas_prev)."""
    T = c.num_classes
    if c.sigma_dist == "linear":
        betas = torch.linspace(c.sigma_begin, c.sigma_end, T)
        alphas = torch.cumprod(1 - betas.flip(0), 0).flip(0)
        alphas_prev = torch.cat([alphas[1:], torch.tensor([1.0]).to(alphas)])
    elif c.sigma_dist == "cosine":
        t = torch.linspace(T, 0, T + 1) / T
        s = 0.008
        f = torch.cos((t + s) / (1 + s) * np.pi / 2) ** 2
        alphas = f[:-1] / f[-1]
        alphas_prev = torch.cat([alphas[1:], torch.tensor([1.0]).to(alphas)])
        betas = 1 - alphas / alphas_prev
    else:
        raise NotImplementedError(c.sigma_dist)
    return betas, alphas, alphas_prev


# --------------------------------------------------------------------------- primitives
def silu(x):
    """layers.py:29-31 (nn.SiLU): x * sigmoid(x)."""
    return x * torch.sigmoid(x)


def group_norm_plain(x, groups, eps):
    """torch.nn.GroupNorm(affine=False): biased variance per (sample, group)."""
    B, C, H, W = x.shape
    xg = x.reshape(B, groups, -1)
    mean = xg.mean(dim=2, keepdim=True)
    var = ((xg - mean) ** 2).mean(dim=2, keepdim=True)
    return ((xg - mean) / torch.sqrt(var + eps)).reshape(B, C, H, W)


def conv2d(x, w, b):
    """layers.py:89-113: nn.Conv2d(stride 1, padding k//2, bias)."""
    return F.conv2d(x, w, b, stride=1, padding=w.shape[-1] // 2)


def conv2d_unfold(x, w, b):
    """Same op as im2col + matmul (used to pin F.conv2d's semantics in tests)."""
    B, C, H, W = x.shape
    co, ci, kh, kw = w.shape
    cols = F.unfold(x, (kh, kw), padding=kh // 2)            # [B, ci*kh*kw, H*W]
    y = torch.matmul(w.reshape(co, -1), cols) + b.reshape(1, co, 1)
    return y.reshape(B, co, H, W)


def nin(x, W, b):
    """layers.py:535-544: y[b,o,h,w] = sum_i x[b,i,h,w] W[i,o] + b[o]."""
    B, C, H, Wd = x.shape
    y = torch.matmul(W.t(), x.reshape(B, C, H * Wd)) + b.reshape(1, -1, 1)
    return y.reshape(B, -1, H, Wd)


def timestep_embedding(t, dim):
    """layers.py:504-518 (fp32 hard-wired, sin block first)."""
    half = dim // 2
    e = math.log(10000) / (half - 1)
    freqs = torch.exp(torch.arange(half, dtype=torch.float32) * -e)
    arg = t.float()[:, None] * freqs[None, :]
    emb = torch.cat([torch.sin(arg), torch.cos(arg)], dim=1)
    if dim % 2 == 1:
        emb = F.pad(emb, (0, 1))
    return emb


def _shift(x, dim, k):
    """y[i] = x[i+k] along dim with zeros outside."""
    if k == 0:
        return x
    z = torch.zeros_like(x)
    n = x.shape[dim]
    src = [slice(None)] * x.dim()
    dst = [slice(None)] * x.dim()
    if k > 0:
        src[dim] = slice(k, n)
        dst[dim] = slice(0, n - k)
    else:
        src[dim] = slice(0, n + k)
        dst[dim] = slice(-k, n)
    z[tuple(dst)] = x[tuple(src)]
    return z


def fir_down2(x):
    """up_or_down_sampling.py:228-258 with k=[1,3,3,1], factor 2:
    per axis y[m] = (x[2m-1] + 3x[2m] + 3x[2m+1] + x[2m+2]) / 8, zero boundary."""
    for dim in (2, 3):
        a = (_shift(x, dim, -1) + 3 * x + 3 * _shift(x, dim, 1) + _shift(x, dim, 2)) / 8
        idx = torch.arange(0, x.shape[dim], 2)
        x = a.index_select(dim, idx)
    return x


def fir_up2(x):
    """up_or_down_sampling.py:196-225 with k=[1,3,3,1], factor 2, gain 4 folded in:
    per axis y[2n] = x[n-1]/4 + 3x[n]/4 ; y[2n+1] = 3x[n]/4 + x[n+1]/4."""
    for dim in (2, 3):
        even = 0.25 * _shift(x, dim, -1) + 0.75 * x
        odd = 0.75 * x + 0.25 * _shift(x, dim, 1)
        shape = list(x.shape)
        shape[dim] *= 2
        x = torch.stack([even, odd], dim=dim + 1).reshape(shape)
    return x


def upfirdn2d_generic(x, kernel, up, down, pad0, pad1):
    """op/upfirdn2d.py:163-204 restated with explicit loops over the taps
    (zero-insert upsample, pad, correlate with the flipped kernel, decimate)."""
    B, C, H, W = x.shape
    kh, kw = kernel.shape
    u = torch.zeros(B, C, H * up, W * up, dtype=x.dtype)
    u[:, :, ::up, ::up] = x
    u = F.pad(u, [max(pad0, 0), max(pad1, 0), max(pad0, 0), max(pad1, 0)])
    u = u[:, :, max(-pad0, 0):u.shape[2] - max(-pad1, 0), max(-pad0, 0):u.shape[3] - max(-pad1, 0)]
    oh = u.shape[2] - kh + 1
    ow = u.shape[3] - kw + 1
    out = torch.zeros(B, C, oh, ow, dtype=x.dtype)
    kf = torch.flip(kernel, [0, 1])
    for i in range(kh):
        for j in range(kw):
            out += kf[i, j] * u[:, :, i:i + oh, j:j + ow]
    return out[:, :, ::down, ::down]


# --------------------------------------------------------------------------- blocks
def spade_modulation(sd, prefix, cond, hw):
    """layerspp.py:164-168: gamma/beta maps from the conditioning frames."""
    seg = F.interpolate(cond, size=hw, mode="nearest")
    a = silu(conv2d(seg, sd[prefix + ".mlp_shared.0.weight"], sd[prefix + ".mlp_shared.0.bias"]))
    gamma = conv2d(a, sd[prefix + ".mlp_gamma.weight"], sd[prefix + ".mlp_gamma.bias"])
    beta = conv2d(a, sd[prefix + ".mlp_beta.weight"], sd[prefix + ".mlp_beta.bias"])
    return gamma, beta


def act_norm(sd, prefix, x, temb_act, c, cond):
    """layerspp.py:518-549 (get_act_norm.forward), 2-D path."""
    ch = x.shape[1]
    if c.spade:
        h = group_norm_plain(x, gn_groups(ch), 1e-6)             # layerspp.py:131
        gamma, beta = spade_modulation(sd, prefix + ".Norm_0", cond, x.shape[-2:])
        h = h * (1 + gamma) + beta                                # layerspp.py:171
    else:
        h = group_norm_plain(x, gn_groups(ch), 1e-5)             # layerspp.py:477
        if temb_act is None:                                      # affine=True when no emb (:509-510)
            h = h * sd[prefix + ".Norm_0.weight"].reshape(1, -1, 1, 1) \
                + sd[prefix + ".Norm_0.bias"].reshape(1, -1, 1, 1)
    if temb_act is not None:
        e = temb_act @ sd[prefix + ".Dense_0.weight"].t() + sd[prefix + ".Dense_0.bias"]
        scale, shift = e[:, :ch, None, None], e[:, ch:, None, None]   # torch.chunk(.,2,dim=1) :523
        h = h * (1 + scale) + shift                               # :535
    return silu(h)


def res_block(sd, p, m, x, temb_act, c, cond):
    """layerspp.py:595-624 (GN) / :675-705 (SPADE)."""
    h = act_norm(sd, p + ".actnorm0", x, temb_act, c, cond)
    if m["up"]:
        h, x = fir_up2(h), fir_up2(x)
    elif m["down"]:
        h, x = fir_down2(h), fir_down2(x)
    h = conv2d(h, sd[p + ".Conv_0.weight"], sd[p + ".Conv_0.bias"])
    h = act_norm(sd, p + ".actnorm1", h, temb_act, c, cond)
    h = conv2d(h, sd[p + ".Conv_1.weight"], sd[p + ".Conv_1.bias"])   # dropout = identity in eval
    if m["cin"] != m["cout"] or m["up"] or m["down"]:
        x = conv2d(x, sd[p + ".Conv_2.weight"], sd[p + ".Conv_2.bias"])
    return (x + h) / RSQRT2_DIV


def attn_block(sd, p, x, c):
    """layerspp.py:230-249."""
    B, C, H, W = x.shape
    h = group_norm_plain(x, gn_groups(C), 1e-6)
    h = h * sd[p + ".GroupNorm_0.weight"].reshape(1, -1, 1, 1) + sd[p + ".GroupNorm_0.bias"].reshape(1, -1, 1, 1)
    q = nin(h, sd[p + ".NIN_0.W"], sd[p + ".NIN_0.b"])
    k = nin(h, sd[p + ".NIN_1.W"], sd[p + ".NIN_1.b"])
    v = nin(h, sd[p + ".NIN_2.W"], sd[p + ".NIN_2.b"])
    nh = c.n_head_channels
    if nh == -1:
        heads = 1
    elif C < nh:
        heads = 1
    else:
        assert C % nh == 0
        heads = C // nh
    D = C // heads
    q = q.reshape(B * heads, D, H * W)
    k = k.reshape(B * heads, D, H * W)
    v = v.reshape(B * heads, D, H * W)
    w = torch.matmul(q.transpose(1, 2), k) * (int(D) ** (-0.5))     # [BH, HW(query), HW(key)]
    w = w - w.max(dim=-1, keepdim=True).values
    w = torch.exp(w)
    w = w / w.sum(dim=-1, keepdim=True)
    o = torch.matmul(v, w.transpose(1, 2)).reshape(B, C, H, W)      # o[c,q] = sum_k w[q,k] v[c,k]
    o = nin(o, sd[p + ".NIN_3.W"], sd[p + ".NIN_3.b"])
    return (x + o) / RSQRT2_DIV


def unet_forward(sd, config, x, t, cond=None, taps=None, cond_mask=None):
    """UNetMore_DDPM.forward (ncsnpp_more.py:753-770) -> NCSNpp.forward (:251-392)
    or SPADE_NCSNpp.forward (:590-718).  `taps`, if a dict, receives the output of
    every module index (for per-module golden checks)."""
    c = hot_cfg(config) if hasattr(config, "model") else config
    mods = module_plan(c)
    P = "unet.all_modules."

    def tap(i, v):
        if taps is not None:
            taps[i] = v
        return v

    if cond is not None and not c.spade:
        x = torch.cat([x, cond], dim=1)                           # :257
    temb = timestep_embedding(t, c.ngf).to(x.dtype)               # :273 (fp32 there; cast only matters for fp64 noise-floor runs)
    temb = tap(0, temb @ sd[P + "0.weight"].t() + sd[P + "0.bias"])
    temb = tap(1, silu(temb) @ sd[P + "1.weight"].t() + sd[P + "1.bias"])   # :278-280
    i = 2
    if c.cond_emb:                                                # :282-286
        if cond_mask is None:
            cond_mask = torch.ones(x.shape[0], dtype=torch.int32)
        e = tap(i, sd[P + "2.weight"][cond_mask.long()].to(x.dtype))          # nn.Embedding lookup
        temb = torch.cat([temb, e], dim=1)
        i += 1
    temb_act = silu(temb)                                         # act_emb(emb), layerspp.py:521

    hs = [tap(i, conv2d(x, sd[P + f"{i}.weight"], sd[P + f"{i}.bias"]))]
    i += 1
    L = len(c.ch_mult)
    for lv in range(L):
        for _ in range(c.num_res_blocks):
            h = tap(i, res_block(sd, P + str(i), mods[i], hs[-1], temb_act, c, cond)); i += 1
            if h.shape[-1] in c.attn_resolutions:
                h = tap(i, attn_block(sd, P + str(i), h, c)); i += 1
            hs.append(h)
        if lv != L - 1:
            h = tap(i, res_block(sd, P + str(i), mods[i], hs[-1], temb_act, c, cond)); i += 1
            hs.append(h)
    h = hs[-1]
    h = tap(i, res_block(sd, P + str(i), mods[i], h, temb_act, c, cond)); i += 1
    h = tap(i, attn_block(sd, P + str(i), h, c)); i += 1
    h = tap(i, res_block(sd, P + str(i), mods[i], h, temb_act, c, cond)); i += 1
    for lv in reversed(range(L)):
        for _ in range(c.num_res_blocks + 1):
            h = torch.cat([h, hs.pop()], dim=1)                   # :356-357  [h, skip]
            h = tap(i, res_block(sd, P + str(i), mods[i], h, temb_act, c, cond)); i += 1
        if h.shape[-1] in c.attn_resolutions:
            h = tap(i, attn_block(sd, P + str(i), h, c)); i += 1
        if lv != 0:
            h = tap(i, res_block(sd, P + str(i), mods[i], h, temb_act, c, cond)); i += 1
    assert not hs
    h = tap(i, act_norm(sd, P + str(i), h, None, c, cond)); i += 1    # :375 / :704
    h = tap(i, conv2d(h, sd[P + str(i) + ".weight"], sd[P + str(i) + ".bias"])); i += 1
    assert i == len(mods)
    if c.output_all_frames and cond is not None:                  # :384-385 -- the split sizes cannot match conv3x3_last's C*nf channels
        _, h = torch.split(h, [c.num_frames_cond * c.channels, c.num_frames * c.channels], dim=1)
    return h


def gamma_tables(betas, alphas, theta_0=0.001):
    """ncsnpp_more.py:745-749: k, k_cum, theta_t of a model.gamma net."""
    k = betas / (alphas * (theta_0 ** 2))
    k_cum = torch.cumsum(k.flip(0), 0).flip(0)
    theta_t = torch.sqrt(alphas) * theta_0
    return k, k_cum, theta_t


class OracleScoreNet:
    """Callable with the scorenet protocol the reference samplers need
    (models/__init__.py:211,221,226): .alphas/.alphas_prev/.betas, __call__(x, labels, cond=)."""

    type = None

    def __init__(self, config, sd, dtype=torch.float32):
        self.config = config
        self.c = hot_cfg(config)
        self.sd = {k: v.to(dtype) for k, v in sd.items()}
        # dtype=float64: same fp32-rounded tables and weights, all arithmetic in double (noise-floor measurements)
        self.betas, self.alphas, self.alphas_prev = (t.to(dtype) for t in make_schedule(self.c))
        if self.c.gamma:
            self.k, self.k_cum, self.theta_t = gamma_tables(self.betas, self.alphas)
        self.cond_noise_fn = None          # noise_in_cond: callable(cond) -> z, else torch.randn_like

    @torch.no_grad()
    def __call__(self, x, y, cond=None, cond_mask=None):
        """UNetMore_DDPM.forward, ncsnpp_more.py:753-770 (the gamma branch of noise_in_cond takes its z from cond_noise_fn too:
        the caller standardises, :761-765)."""
        if self.c.noise_in_cond and cond is not None:
            ua = self.alphas[y].reshape(cond.shape[0], 1, 1, 1)
            z = self.cond_noise_fn(cond) if self.cond_noise_fn is not None else torch.randn_like(cond)
            cond = ua.sqrt() * cond + (1 - ua).sqrt() * z
        return unet_forward(self.sd, self.c, x, y, cond, cond_mask=cond_mask)

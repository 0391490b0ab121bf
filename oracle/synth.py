"""CPU ORACLE -- TEST INFRASTRUCTURE ONLY (see oracle/unet_ref.py header).

Synthetic configs / weights / inputs / noise shared by the golden generator,
the parity tests and bench.py's cpu_baseline leg.  Weights are
drawn per parameter from a generator keyed by (seed, crc32(name)) so the set is
independent of enumeration order, and are NEVER the reference default init
(init_scale=0 zeroes 217/376 tensors).
"""
import argparse
import zlib

import torch

from . import unet_ref


def dict2namespace(d):
    """main.py:359-367."""
    ns = argparse.Namespace()
    for k, v in d.items():
        setattr(ns, k, dict2namespace(v) if isinstance(v, dict) else v)
    return ns


def make_config(name):
    """Hot-path subset of the reference YAMLs with BASELINE.json's overrides
    (SURVEY section 0 table).  `tiny*` are small shapes for fast CPU/GPU unit parity."""
    data = dict(image_size=64, channels=1, num_frames=5, num_frames_cond=5, num_frames_future=0,
                logit_transform=False, uniform_dequantization=False, gaussian_dequantization=False,
                rescaled=True)
    model = dict(version="DDPM", arch="unetmore", type="v1", time_conditional=True, dropout=0.1,
                 sigma_dist="linear", sigma_begin=0.02, sigma_end=0.0001, num_classes=1000,
                 ngf=64, ch_mult=[1, 2, 3, 4], num_res_blocks=2, attn_resolutions=[8, 16, 32],
                 n_head_channels=64, noise_in_cond=False, output_all_frames=False, cond_emb=False,
                 spade=False, spade_dim=128, gamma=False)
    sampling = dict(subsample=100, denoise=True, clip_before=True, final_only=True, init_prev_t=-1.0,
                    num_frames_pred=20, one_frame_at_a_time=False, n_steps_each=0, step_lr=0.0)
    if name == "smmnist_big5":               # configs/smmnist_DDPM_big5.yml as shipped (BASELINE cfg 1)
        pass
    elif name == "smmnist_big5_ngf96":       # BASELINE cfg 2
        model.update(ngf=96, n_head_channels=96)
    elif name == "kth64_big_ngf128":         # BASELINE cfg 3 (configs/kth64_big.yml + overrides)
        data.update(num_frames_cond=10)
        model.update(ngf=128, n_head_channels=128)
    elif name == "bair_big_spade":           # BASELINE cfg 4 (configs/bair_big_spade.yml)
        data.update(channels=3, num_frames_cond=2)
        model.update(ngf=96, n_head_channels=96, spade=True, spade_dim=128)
        sampling.update(subsample=1000)
    elif name == "cityscapes_big":           # BASELINE cfg 5, shipped ch_mult (configs/cityscapes_big.yml)
        data.update(image_size=128, channels=3, num_frames_cond=2)
        model.update(ngf=128, n_head_channels=128, ch_mult=[1, 1, 2, 3, 4])
        sampling.update(num_frames_pred=28)
    elif name == "cityscapes_big_spade":     # configs/cityscapes_big_spade.yml as shipped: ngf 192, 2/3/4 heads of 192 channels, SPADE dim 256
        data.update(image_size=128, channels=3, num_frames_cond=2)
        model.update(ngf=192, n_head_channels=192, ch_mult=[1, 1, 2, 3, 4], spade=True, spade_dim=256)
        sampling.update(num_frames_pred=28)
    elif name == "cityscapes_big_variant":   # BASELINE.json wording: ch_mult [1,2,3,4,4], attention at 16 only
        data.update(image_size=128, channels=3, num_frames_cond=2)
        model.update(ngf=128, n_head_channels=128, ch_mult=[1, 2, 3, 4, 4], attn_resolutions=[16])
        sampling.update(num_frames_pred=28)
    elif name == "tiny":                     # 32x32, 3 levels, attention at 16 and 8
        data.update(image_size=32, num_frames=2, num_frames_cond=2)
        model.update(ngf=32, n_head_channels=32, ch_mult=[1, 2, 2], num_res_blocks=1, attn_resolutions=[8, 16])
        sampling.update(subsample=10)
    elif name == "tiny_uncond":              # no conditioning frames at all (conditioning_fn(..., conditional=False), ncsn_runner.py:109-110)
        data.update(image_size=32, num_frames=2, num_frames_cond=0)
        model.update(ngf=32, n_head_channels=32, ch_mult=[1, 2, 2], num_res_blocks=1, attn_resolutions=[8, 16])
        sampling.update(subsample=10)
    elif name == "tiny_cosine":              # `tiny` with the cosine alpha-bar schedule (models/__init__.py:28-32)
        data.update(image_size=32, num_frames=2, num_frames_cond=2)
        model.update(ngf=32, n_head_channels=32, ch_mult=[1, 2, 2], num_res_blocks=1, attn_resolutions=[8, 16],
                     sigma_dist="cosine")
        sampling.update(subsample=10)
    elif name in ("tiny_condemb", "tiny_noisecond", "tiny_gamma", "tiny_allframes"):   # SURVEY 8f rank 4 flags on the `tiny` net
        data.update(image_size=32, num_frames=2, num_frames_cond=2)
        model.update(ngf=32, n_head_channels=32, ch_mult=[1, 2, 2], num_res_blocks=1, attn_resolutions=[8, 16])
        model.update(dict(tiny_condemb=dict(cond_emb=True), tiny_noisecond=dict(noise_in_cond=True),
                          tiny_gamma=dict(gamma=True, noise_in_cond=True), tiny_allframes=dict(output_all_frames=True))[name])
        sampling.update(subsample=10)
    elif name == "tiny_spade_noisecond":     # SPADE + noise_in_cond: gamma/beta cannot be hoisted out of the step
        data.update(image_size=32, channels=3, num_frames=2, num_frames_cond=1, num_frames_future=1)
        model.update(ngf=32, n_head_channels=32, ch_mult=[1, 2, 2], num_res_blocks=1, attn_resolutions=[8, 16],
                     spade=True, spade_dim=32, noise_in_cond=True)
        sampling.update(subsample=10)
    elif name == "tiny_spade":
        data.update(image_size=32, channels=3, num_frames=2, num_frames_cond=1, num_frames_future=1)
        model.update(ngf=32, n_head_channels=32, ch_mult=[1, 2, 2], num_res_blocks=1, attn_resolutions=[8, 16],
                     spade=True, spade_dim=32)
        sampling.update(subsample=10)
    else:
        raise KeyError(name)
    return dict2namespace(dict(data=data, model=model, sampling=sampling))


def _gen(seed, name):
    g = torch.Generator()
    g.manual_seed((seed * 1000003 + zlib.crc32(name.encode())) & 0x7FFFFFFF)
    return g


def make_state_dict(config, seed=123):
    """name -> fp32 CPU tensor for every parameter (SURVEY 8d recipe)."""
    c = unet_ref.hot_cfg(config)
    sd = {}
    for name, shape in unet_ref.param_shapes(c).items():
        g = _gen(seed, name)
        if len(shape) >= 2:
            if name.endswith(".W"):                       # NIN: [in, out] (layers.py:538)
                fan_in = shape[0]
            else:                                         # Linear [out,in], Conv [out,in,kh,kw]
                fan_in = 1
                for s in shape[1:]:
                    fan_in *= s
            t = torch.randn(shape, generator=g) * (1.0 / fan_in) ** 0.5
        elif name.endswith(".weight"):                    # GroupNorm gains
            t = 1.0 + 0.1 * torch.randn(shape, generator=g)
        else:                                             # biases, NIN.b
            t = 0.1 * torch.randn(shape, generator=g)
        sd[name] = t.float().contiguous()
    return sd


def make_inputs(config, batch, seed=0):
    """x_init ~ N(0,1) (seed), cond ~ clamp(N(0,1),-1,1) (seed+1).  Rows are keyed by
    GLOBAL sample index so a rank's shard equals the same rows of the full batch."""
    c = unet_ref.hot_cfg(config)
    S = c.image_size
    xs, cs = [], []
    for b in range(batch):
        gx = _gen(seed, f"x{b}")
        gc = _gen(seed + 1, f"c{b}")
        xs.append(torch.randn(c.channels * c.num_frames, S, S, generator=gx))
        cs.append(torch.randn(c.channels * c.num_frames_cond, S, S, generator=gc).clamp(-1, 1))
    return torch.stack(xs), torch.stack(cs)


def make_noise(config, batch, n_steps, seed=2):
    """[n_steps, B, C*nf, S, S] host noise sequence for injected-noise parity runs."""
    c = unet_ref.hot_cfg(config)
    S = c.image_size
    g = _gen(seed, "noise")
    return torch.randn(n_steps, batch, c.channels * c.num_frames, S, S, generator=g)

"""Caller-side glue of the sampling path, restated so the reference's sampling scripts have everything they use around
the sampler: conditioning layout, data transforms and the autoregressive block driver of `NCSNRunner.video_gen`.
Everything stays on the GPU between blocks (the reference moves each block to the CPU and back,
runners/ncsn_runner.py:1521-1539).
"""
from math import ceil

import torch

from .samplers import get_sampler


def conditioning_fn(config, X, num_frames_pred=0, prob_mask_cond=0.0, prob_mask_future=0.0, conditional=True):
    """Frames -> (pred, cond, cond_mask): frame-major channel stacking `t*C + c`, cond = [past..., future...]
    (reference: runners/ncsn_runner.py:104-147)."""
    imsize = config.data.image_size
    if not conditional:
        return X.reshape(len(X), -1, imsize, imsize), None, None
    cond = config.data.num_frames_cond
    train = config.data.num_frames
    pred = num_frames_pred
    future = getattr(config.data, "num_frames_future", 0)
    pred_frames = X[:, cond:cond + pred].reshape(len(X), -1, imsize, imsize)
    cond_frames = X[:, :cond].reshape(len(X), -1, imsize, imsize)
    if prob_mask_cond > 0.0:
        cond_mask = (torch.rand(X.shape[0], device=X.device) > prob_mask_cond)
        cond_frames = cond_mask.reshape(-1, 1, 1, 1) * cond_frames
        cond_mask = cond_mask.to(torch.int32)
    else:
        cond_mask = None
    if future > 0:
        if prob_mask_future == 1.0:
            future_frames = torch.zeros(len(X), config.data.channels * future, imsize, imsize, device=X.device)
        else:
            future_frames = X[:, cond + train:cond + train + future].reshape(len(X), -1, imsize, imsize)
            if prob_mask_future > 0.0:
                if getattr(config.data, "prob_mask_sync", False):
                    future_mask = cond_mask
                else:
                    future_mask = (torch.rand(X.shape[0], device=X.device) > prob_mask_future)
                future_frames = future_mask.reshape(-1, 1, 1, 1) * future_frames
        cond_frames = torch.cat([cond_frames, future_frames], dim=1)
    return pred_frames, cond_frames, cond_mask


def data_transform(config, X):
    """[0,1] frames -> network range (reference: datasets/__init__.py:235-249)."""
    d = config.data
    if getattr(d, "uniform_dequantization", False):
        X = X / 256. * 255. + torch.rand_like(X) / 256.
    if getattr(d, "gaussian_dequantization", False):
        X = X + torch.randn_like(X) * 0.01
    if getattr(d, "rescaled", False):
        X = 2 * X - 1.
    elif getattr(d, "logit_transform", False):
        lam = 1e-6
        X = lam + (1 - 2 * lam) * X
        X = torch.log(X) - torch.log1p(-X)
    if hasattr(config, "image_mean"):
        return X - config.image_mean.to(X.device)[None, ...]
    return X


def inverse_data_transform(config, X):
    """Network range -> [0,1] (reference: datasets/__init__.py:252-261)."""
    d = config.data
    if hasattr(config, "image_mean"):
        X = X + config.image_mean.to(X.device)[None, ...]
    if getattr(d, "logit_transform", False):
        X = torch.sigmoid(X)
    elif getattr(d, "rescaled", False):
        X = (X + 1.) / 2.
    return torch.clamp(X, 0.0, 1.0)


@torch.no_grad()
def video_gen(config, scorenet, cond, num_frames_pred=None, init_noise_fn=None, sampler=None, **sampler_kwargs):
    """Autoregressive block loop of NCSNRunner.video_gen (runners/ncsn_runner.py:1504-1569, future == 0 path):
    generate ceil(num_frames_pred / num_frames) blocks, sliding the conditioning window on the device, and return
    [B, C*num_frames_pred, S, S].  `init_noise_fn(block_index, shape, device)` supplies z for each block
    (default torch.randn on the device, like :1476/:1551)."""
    d, s = config.data, config.sampling
    C, nf, nc, S = d.channels, d.num_frames, d.num_frames_cond, d.image_size
    nfp = int(num_frames_pred if num_frames_pred is not None else s.num_frames_pred)
    if getattr(s, "one_frame_at_a_time", False):
        raise NotImplementedError("sampling.one_frame_at_a_time")
    sampler = sampler or get_sampler(config)
    dev = scorenet.device
    cond = cond.to(dev).float().contiguous()
    B = cond.shape[0]
    shape = (B, C * nf, S, S)
    init_noise_fn = init_noise_fn or (lambda i, shp, dv: torch.randn(shp, device=dv))
    t_min = getattr(s, "init_prev_t", -1)
    n_iter = ceil(nfp / nf)
    preds, gen = [], None
    for i in range(n_iter):
        init = init_noise_fn(i, shape, dev) if (i == 0 or t_min <= 0) else gen          # :1513
        out = sampler(init, scorenet, cond=cond, cond_mask=None, final_only=True, denoise=getattr(s, "denoise", True),
                      subsample_steps=getattr(s, "subsample", None), clip_before=getattr(s, "clip_before", True),
                      t_min=t_min, verbose=False, log=False, **sampler_kwargs)
        gen = out[-1].reshape(B, C * nf, S, S)                                          # :1521-1522
        preds.append(gen)
        if i == n_iter - 1:
            continue
        # cond <- [cond[:, C*nf:], gen[:, C*max(0, nf - nc):]]                            :1537-1539
        cond = torch.cat([cond[:, C * nf:], gen[:, C * max(0, nf - nc):]], dim=1).contiguous()
    return torch.cat(preds, dim=1)[:, :C * nfp]                                         # :1569


def save_video_pred(path, cond, pred, real):
    """The on-disk result of NCSNRunner.video_gen: torch.save({"cond", "pred", "real"}) of the [0,1]-range CPU tensors
    (runners/ncsn_runner.py:2106-2112, `videos_pred_<ckpt>.pt`), so downstream metric scripts read our output unchanged."""
    to_cpu = lambda t: None if t is None else t.detach().to("cpu")
    torch.save({"cond": to_cpu(cond), "pred": to_cpu(pred), "real": to_cpu(real)}, path)
    return path

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
def video_gen(config, scorenet, cond, num_frames_pred=None, init_noise_fn=None, sampler=None, data_init=None, verbose=False,
              log=False, **sampler_kwargs):
    """Autoregressive block loop of NCSNRunner.video_gen (runners/ncsn_runner.py:1476-1569, prediction path: future == 0), kept on
    the device between blocks (the reference moves every block to the CPU and back, :1521-1539).  Returns [B, C*num_frames_pred, S, S].

      * blocks: ceil(num_frames_pred / num_frames), or num_frames_pred when `sampling.one_frame_at_a_time` (:1501-1504);
      * block input: fresh z for every block (:1476, :1551) unless `sampling.init_prev_t` > 0, where block i > 0 restarts from the
        previous block's output and the sampler re-noises it (:1513, models/__init__.py:269-280);
      * cond update (:1528-1539): `cond is None` (unconditional bootstrap) -> cond = gen; one_frame_at_a_time -> drop the oldest
        cond frame, append the first generated frame; else drop the oldest num_frames cond frames, append the newest
        min(num_frames, num_frames_cond) generated frames;
      * `data_init` (sampling.data_init, :1479-1498, :1553-1565): frames `real_init` already in network range, flattened to
        channels as the reference does; block input = sqrt(alpha_0) * real_init1 + sqrt(1 - alpha_0) * z.  The later-block slices
        follow the reference's expressions literally (they index dim 0 there);
      * result: cat(blocks, dim=1)[:, :C*num_frames_pred] (:1569).

    `init_noise_fn(block_index, shape, device)` supplies z (default torch.randn on the device).  A `seed=` kwarg (on-device Philox
    step noise) is advanced by one per block, so blocks never share a noise stream."""
    d, s = config.data, config.sampling
    C, nf, nc, S = d.channels, d.num_frames, d.num_frames_cond, d.image_size
    nfp = int(num_frames_pred if num_frames_pred is not None else s.num_frames_pred)
    one_at_a_time = bool(getattr(s, "one_frame_at_a_time", False))
    sampler = sampler or get_sampler(config)
    dev = scorenet.device
    if cond is not None:
        cond = cond.to(dev).float().contiguous()
        B = cond.shape[0]
    else:
        B = int(sampler_kwargs.pop("batch_size", getattr(s, "batch_size", 1)))
    shape = (B, C * nf, S, S)
    init_noise_fn = init_noise_fn or (lambda i, shp, dv: torch.randn(shp, device=dv))
    t_min = getattr(s, "init_prev_t", -1)
    n_iter = nfp if one_at_a_time else ceil(nfp / nf)                                  # :1501-1504
    seed = sampler_kwargs.pop("seed", None)
    real_init = None
    if data_init is not None:
        real_init = data_init.to(dev).float().reshape(len(data_init), -1, S, S)        # conditioning_fn(..., conditional=False) :109-110
        alpha0 = scorenet.alphas[0]

    def init_for(i, real_init1):
        z = init_noise_fn(i, shape, dev)
        if real_init is None:
            return z
        return alpha0.sqrt() * real_init1 + (1 - alpha0).sqrt() * z                   # :1495-1496, :1563-1564

    init = init_for(0, real_init[:, :C * nf] if real_init is not None else None)       # :1488
    preds, gen = [], None
    for i in range(n_iter):
        x0 = init if (i == 0 or t_min <= 0) else gen                                    # :1513
        kw = dict(sampler_kwargs)
        if seed is not None:
            kw["seed"] = int(seed) + i
        out = sampler(x0, scorenet, cond=cond, cond_mask=None, final_only=True, denoise=getattr(s, "denoise", True),
                      subsample_steps=getattr(s, "subsample", None), clip_before=getattr(s, "clip_before", True),
                      t_min=t_min, verbose=verbose, log=log, **kw)
        gen = out[-1].reshape(B, C * nf, S, S)                                          # :1521-1522
        preds.append(gen)
        if i == n_iter - 1:
            continue
        if cond is None:                                                                # :1528-1529
            cond = gen
        elif one_at_a_time:                                                             # :1530-1531
            cond = torch.cat([cond[:, C:], gen[:, :C]], dim=1).contiguous()
        else:                                                                           # :1532-1535
            cond = torch.cat([cond[:, C * nf:], gen[:, C * max(0, nf - nc):]], dim=1).contiguous()
        real_init1 = None
        if real_init is not None:                                                       # :1553-1557 (dim-0 slices, sic)
            real_init1 = real_init[C * (i + 1):C * (i + 1 + nf)] if one_at_a_time else \
                real_init[(i + 1) * C * nf:(i + 2) * C * nf]
        init = init_for(i + 1, real_init1)
    return torch.cat(preds, dim=1)[:, :C * nfp]                                         # :1569


def frames_to_uint8(scorenet, frames01, channels):
    """[B, T*C, H, W] frames in [0, 1] (after `inverse_data_transform`) -> uint8 [B, T, H, W, C] on the device: the packing the
    reference applies to every frame before it writes GIFs / PNGs (`(frame * 255).astype('uint8')` on the HWC view,
    runners/ncsn_runner.py:2019-2062).  The grid / caption drawing around it (torchvision make_grid, cv2.putText) is host-side
    presentation and stays with the caller."""
    import ctypes as C
    from . import _lib
    f = frames01.to(device=scorenet.device, dtype=torch.float32).contiguous()
    B, TC, H, W = f.shape
    if TC % channels:
        raise ValueError(f"{TC} channels is not a multiple of {channels}")
    out = torch.empty((B, TC // channels, H, W, channels), dtype=torch.uint8, device=f.device)
    with torch.cuda.device(f.device):
        scorenet._bind_stream()
        _lib.check(_lib.lib.mcvd_pack_frames_u8(scorenet._ctx, C.c_void_p(f.data_ptr()), C.c_void_p(out.data_ptr()), B, TC // channels,
                                                channels, H, W), "pack_frames_u8")
    return out


def save_video_pred(path, cond, pred, real):
    """The on-disk result of NCSNRunner.video_gen: torch.save({"cond", "pred", "real"}) of the [0,1]-range CPU tensors
    (runners/ncsn_runner.py:2106-2112, `videos_pred_<ckpt>.pt`), so downstream metric scripts read our output unchanged."""
    to_cpu = lambda t: None if t is None else t.detach().to("cpu")
    torch.save({"cond": to_cpu(cond), "pred": to_cpu(pred), "real": to_cpu(real)}, path)
    return path

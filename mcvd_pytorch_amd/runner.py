"""Caller-side glue of the sampling path, restated so the reference's sampling scripts have everything they use around
the sampler: conditioning layout, data transforms and the autoregressive block driver of `NCSNRunner.video_gen`.
Everything stays on the GPU between blocks (the reference moves each block to the CPU and back,
runners/ncsn_runner.py:1521-1539).
"""
from math import ceil

import torch

from .samplers import get_sampler


def _frames_to_channels(frames):
    """[B, T, C, H, W] -> [B, T*C, H, W]: frame-major channel stacking (channel index t*C + c), the layout every tensor of the
    sampling path uses."""
    return frames.flatten(1, 2)


def _keep_mask(n, p_drop, device):
    """Per-sample Bernoulli keep mask (True with probability 1 - p_drop), one torch.rand draw per call as the reference's masks."""
    return torch.rand(n, device=device) > p_drop


def conditioning_fn(config, X, num_frames_pred=0, prob_mask_cond=0.0, prob_mask_future=0.0, conditional=True):
    """Clip [B, T, C, H, W] -> (pred, cond, cond_mask) in the channel layout of the network: pred = the `num_frames_pred` frames after
    the past, cond = [past frames | future frames] with whole samples zeroed by the conditioning masks.  Behaviour of
    NCSNRunner's conditioning_fn (runners/ncsn_runner.py:104-147), including its draw order (cond mask, then future mask), so a
    seeded reference script sees the same masks; `conditional=False` returns the flattened clip alone."""
    d = config.data
    if not conditional:
        return _frames_to_channels(X), None, None
    n_past, n_train, n_future = d.num_frames_cond, d.num_frames, getattr(d, "num_frames_future", 0)
    B, dev = X.shape[0], X.device
    pred = _frames_to_channels(X[:, n_past:n_past + num_frames_pred])
    blocks = [_frames_to_channels(X[:, :n_past])]
    cond_mask = None
    if prob_mask_cond > 0.0:
        keep = _keep_mask(B, prob_mask_cond, dev)
        blocks[0] = blocks[0] * keep.view(B, 1, 1, 1)
        cond_mask = keep.to(torch.int32)
    if n_future > 0:
        if prob_mask_future == 1.0:                   # the future block is dropped for everybody: zeros, whatever the clip holds
            fut = torch.zeros(B, d.channels * n_future, d.image_size, d.image_size, device=dev)
        else:
            start = n_past + n_train
            fut = _frames_to_channels(X[:, start:start + n_future])
            if prob_mask_future > 0.0:
                if getattr(d, "prob_mask_sync", False):
                    if cond_mask is None:
                        raise AttributeError("prob_mask_sync needs prob_mask_cond > 0 (the reference reuses the cond mask there)")
                    keep_f = cond_mask
                else:
                    keep_f = _keep_mask(B, prob_mask_future, dev)
                fut = fut * keep_f.view(B, 1, 1, 1)
        blocks.append(fut)
    return pred, torch.cat(blocks, dim=1) if len(blocks) > 1 else blocks[0], cond_mask


def _mean_image(config, like):
    m = getattr(config, "image_mean", None)
    return None if m is None else m.to(like.device)[None, ...]


def data_transform(config, X):
    """[0, 1] frames -> the range the network was trained on: the `config.data` switches of datasets/__init__.py:235-249
    (dequantisation noise, `rescaled` to [-1, 1] or the logit transform, minus the dataset's mean image when one is configured)."""
    d = config.data
    if getattr(d, "uniform_dequantization", False):
        X = X / 256.0 * 255.0 + torch.rand_like(X) / 256.0
    if getattr(d, "gaussian_dequantization", False):
        X = X + 0.01 * torch.randn_like(X)
    if getattr(d, "rescaled", False):
        X = 2.0 * X - 1.0
    elif getattr(d, "logit_transform", False):
        Y = 1e-6 + (1.0 - 2.0 * 1e-6) * X
        X = torch.log(Y) - torch.log1p(-Y)
    mean = _mean_image(config, X)
    return X if mean is None else X - mean


def inverse_data_transform(config, X):
    """The inverse map back to [0, 1], clamped (datasets/__init__.py:252-261)."""
    d = config.data
    mean = _mean_image(config, X)
    if mean is not None:
        X = X + mean
    if getattr(d, "logit_transform", False):
        X = torch.sigmoid(X)
    elif getattr(d, "rescaled", False):
        X = 0.5 * (X + 1.0)
    return X.clamp(0.0, 1.0)


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

      * `config.model.gamma` (:1470-1474, :1518, :1545-1549): every sampler call gets `gamma=True` and the initial z of every block is the
        centred gamma variate Gamma(k_cum[0], rate 1 / theta_t[0]) - k_cum[0] * theta_t[0].  With `data_init` as well the reference
        divides by an undefined `used_alphas` (:1494, NameError): that combination is refused here too.

    `init_noise_fn(block_index, shape, device)` supplies z (default: torch.randn on the device, or the centred gamma variate for a
    `model.gamma` config).  A `seed=` kwarg (on-device Philox step noise) is advanced by one per block, so blocks never share a
    noise stream."""
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
    gamma = bool(getattr(config.model, "gamma", False))
    if gamma and data_init is not None:
        raise NameError("name 'used_alphas' is not defined (runners/ncsn_runner.py:1494: the reference cannot run model.gamma with "
                        "sampling.data_init; refused here as well)")
    if init_noise_fn is None:
        if gamma:
            k0, th0 = float(scorenet.k_cum[0]), float(scorenet.theta_t[0])

            def init_noise_fn(i, shp, dv):
                # drawn on the CPU and then moved, as the reference does (:1470-1474, :1545-1549: `Gamma(full(...), full(...)).sample().to(device)`):
                # under torch.manual_seed a reference script and this loop consume the same CPU generator stream
                g = torch.distributions.gamma.Gamma(torch.full(shp, k0), torch.full(shp, 1.0 / th0)).sample().to(dv)
                return g - k0 * th0
        else:
            def init_noise_fn(i, shp, dv):
                return torch.randn(shp, device=dv)
    t_min = getattr(s, "init_prev_t", -1)
    n_iter = nfp if one_at_a_time else ceil(nfp / nf)                                  # :1501-1504
    seed = sampler_kwargs.pop("seed", None)
    sampler_kwargs.pop("gamma", None)                 # decided by config.model.gamma (:1518); a duplicate keyword would be a TypeError below
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
                      t_min=t_min, gamma=gamma, verbose=verbose, log=log, **kw)
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

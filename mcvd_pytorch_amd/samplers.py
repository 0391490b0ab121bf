"""DDPM / DDIM / F-PNDM samplers with the reference's call surface (models/__init__.py:206-209, :102-104, :37-38).

    sampler(x_mod, scorenet, cond=None, final_only=..., denoise=..., subsample_steps=..., clip_before=...,
            t_min=..., verbose=..., log=..., **kwargs) -> Tensor [(1 | n_saved), B, C*nf, H, W]

Unknown kwargs (`cond_mask`, `n_steps_each`, `step_lr`, `config`, ... passed by runners/ncsn_runner.py:1513-1520)
are accepted and ignored, as in the reference.  Two execution paths, same algebra:

  * device loop  (`final_only=True`, no verbose/log, no same_noise/frac_steps): the whole L-step loop runs inside
    `mcvd_sampler_run` -- labels, UNet forward, fused x0/clip/posterior/noise update, denoise pass -- with either an
    injected noise sequence (`noise=` kwarg, for parity runs) or the on-device counter-based Philox stream keyed by
    (seed, global sample index, draw).
  * host loop: mirrors the reference loop statement by statement (images list, 10x logging of norms, same_noise,
    frac_steps, just_beta) and calls the HIP forward + the fused update kernel once per step.

`scorenet` must be a HipScoreNet (optionally wrapped in something exposing `.module`); anything else raises --
this package has no eager fallback.
"""
import ctypes as C
import logging
from functools import partial

import numpy as np
import torch

from . import _lib
from .scorenet import HipScoreNet


def _unwrap(scorenet):
    net = scorenet.module if hasattr(scorenet, "module") else scorenet      # models/__init__.py:211
    if not isinstance(net, HipScoreNet):
        raise TypeError(f"mcvd_pytorch_amd samplers need a HipScoreNet, got {type(net).__name__} (no eager fallback)")
    return net


def _subsample(net, subsample_steps):
    """Schedule subsampling, models/__init__.py:229-237 (fp32, on CPU copies of the buffers)."""
    alphas, alphas_prev, betas = net.alphas.cpu(), net.alphas_prev.cpu(), net.betas.cpu()
    steps = np.arange(len(betas))
    if subsample_steps is not None and subsample_steps < len(alphas):
        skip = len(alphas) // subsample_steps
        steps = torch.tensor(list(range(0, len(alphas), skip)))
        alphas = alphas.index_select(0, steps)
        alphas_prev = torch.cat([alphas[1:], torch.tensor([1.0]).to(alphas)])
        betas = 1.0 - torch.div(alphas, alphas_prev)
    return steps, alphas, alphas_prev, betas


def _skipped(step, t_min, n, subsampled):
    """`step < t_min*len(alphas)` (models/__init__.py:269) as the reference evaluates it: on a subsampled schedule `step` is a 0-dim
    int64 tensor and torch compares it with the Python float in float32; otherwise it is a numpy int64 and the comparison is in double."""
    thr = t_min * n
    if subsampled:
        return bool(np.float32(step) < np.float32(thr))
    return step < thr


def _draw_seed():
    """A 63-bit seed from torch's default CPU generator, so torch.manual_seed() controls the device stream."""
    return int(torch.randint(0, 2 ** 62, (1,), dtype=torch.int64).item())


def _f(t):
    return float(t)          # fp32 0-dim tensor -> python float (exact)


@torch.no_grad()
def _sample(kind, x_mod, scorenet, cond=None, just_beta=False, final_only=False, denoise=True, subsample_steps=None,
            same_noise=False, noise_val=None, frac_steps=None, verbose=False, log=False, clip_before=True, t_min=-1,
            gamma=False, noise=None, seed=None, sample_offset=0, cond_noise=None, **kwargs):
    net = _unwrap(scorenet)
    if gamma and not getattr(net, "gamma", False):
        raise AttributeError("'HipScoreNet' object has no attribute 'k_cum' (gamma=True needs a model.gamma net, models/__init__.py:224, :118)")
    # ddim_sampler reads `gamma` in ONE place: the t_min re-noise of the first executed step draws a standardised Gamma variate (:144-151);
    # it adds no step noise, so without t_min > 0 the kwarg changes nothing but the log prefix ("DDIM gamma")
    gamma_device = gamma and kind == _lib.SAMPLER_DDPM
    net.sync_parameters(force=True)
    dev = net.device
    x = x_mod.to(device=dev, dtype=torch.float32).contiguous().clone()
    if cond is not None:
        cond = cond.to(device=dev, dtype=torch.float32).contiguous()
    B = x.shape[0]
    per = x[0].numel()
    if noise is not None:
        noise = noise.to(device=dev, dtype=torch.float32).contiguous()
    if noise_val is not None:
        noise_val = noise_val.to(device=dev, dtype=torch.float32).contiguous()
    name = ("DDPM" if kind == _lib.SAMPLER_DDPM else "DDIM") + (" gamma" if gamma else "")
    # The device loop hands raw pointers to the library, which sizes everything from the model description: check every
    # shape here (the same conditions HipScoreNet.__call__ enforces per forward), so a wrong-shaped tensor raises instead of
    # being read / written out of bounds.
    d = net._desc
    want = (d.channels * d.num_frames, d.image_size, d.image_size)
    if x.dim() != 4 or tuple(x.shape[1:]) != want:
        raise RuntimeError(f"x_mod has shape {tuple(x.shape)}, the model expects [B, {want[0]}, {want[1]}, {want[2]}]")
    if d.num_frames_cond > 0:
        cwant = (B, d.channels * d.num_frames_cond, d.image_size, d.image_size)
        if cond is None or tuple(cond.shape) != cwant:
            raise RuntimeError(f"cond missing or mis-shaped: got {None if cond is None else tuple(cond.shape)}, expected {cwant}")
    elif cond is not None:
        raise RuntimeError("this model takes no conditioning frames (num_frames_cond == 0) but cond was passed")
    # the steps the loop will run: schedule subsampling (:229-237), then the frac_steps cut (:249-256: the tail of the step list; the
    # t_min test below compares against the CUT length, as the reference's `len(alphas)` does), then the t_min skip (:269-270)
    skip_all = d.num_classes // int(subsample_steps) if (subsample_steps is not None and subsample_steps < d.num_classes) else 1
    step_list = list(range(0, d.num_classes, skip_all))
    if frac_steps is not None and kind == _lib.SAMPLER_DDPM:
        step_list = step_list[int((1 - frac_steps) * len(step_list)):]
    subsampled = subsample_steps is not None and subsample_steps < d.num_classes
    n_exec = sum(1 for st in step_list if not _skipped(st, t_min, len(step_list), subsampled))
    if noise is not None:
        # draws of the call: one per executed DDPM step but the last -- none with same_noise, where every step adds noise_val (:316-317) --
        # plus the t_min re-noise of the first executed step, which always draws (:272-279)
        step_draws = max(n_exec - 1, 0) if (kind == _lib.SAMPLER_DDPM and not same_noise) else 0
        need = step_draws + (1 if (t_min > 0 and n_exec > 0) else 0)
        if noise.dim() != 5 or tuple(noise.shape[1:]) != tuple(x.shape) or noise.shape[0] < need:
            raise RuntimeError(f"injected noise has shape {tuple(noise.shape)}; need at least [{need}, {', '.join(map(str, x.shape))}]")
    if noise_val is not None and tuple(noise_val.shape) != tuple(x.shape):
        raise RuntimeError(f"noise_val has shape {tuple(noise_val.shape)}, expected {tuple(x.shape)}")
    nic = bool(getattr(net, "noise_in_cond", False)) and cond is not None
    if cond_noise is not None:
        if not nic:
            raise RuntimeError("cond_noise was passed but the model has no noise_in_cond")
        cond_noise = cond_noise.to(device=dev, dtype=torch.float32).contiguous()
        n_fwd = n_exec + (1 if denoise else 0)
        if cond_noise.dim() != 5 or tuple(cond_noise.shape[1:]) != tuple(cond.shape) or cond_noise.shape[0] < n_fwd:
            raise RuntimeError(f"cond_noise has shape {tuple(cond_noise.shape)}; need at least [{n_fwd}, {', '.join(map(str, cond.shape))}]")

    _lib.check(_lib.lib.mcvd_ctx_clear_range(net._ctx), "clear_range")      # the f16x2 range verdict below is about THIS call's forwards
    fast = final_only and not verbose and not log and not same_noise and noise_val is None and frac_steps is None
    if gamma and not gamma_device and t_min > 0 and n_exec > 0:
        fast = False                   # DDIM with a gamma re-noise draw: the host loop (the device loop's gamma stream belongs to the DDPM sampler)
    if fast:
        flags = (_lib.FLAG_DENOISE if denoise else 0) | (_lib.FLAG_CLIP_BEFORE if clip_before else 0) \
            | (_lib.FLAG_JUST_BETA if just_beta else 0) | (_lib.FLAG_GAMMA if gamma_device else 0)
        if seed is None and (noise is None or (nic and cond_noise is None)):
            seed = _draw_seed()
        with torch.cuda.device(dev):
            net._bind_stream()
            if nic:                        # conditioning noise: the injected sequence, else the library's Philox stream of this call
                _lib.check(_lib.lib.mcvd_model_set_cond_noise(
                    net._model, C.c_void_p(cond_noise.data_ptr()) if cond_noise is not None else None, 0, 0, 0), "set_cond_noise")
            try:
                rc = _lib.lib.mcvd_sampler_run(
                    net._model, kind, C.c_void_p(x.data_ptr()), C.c_void_p(cond.data_ptr()) if cond is not None else None,
                    C.c_void_p(noise.data_ptr()) if noise is not None else None, C.c_uint64(seed or 0),
                    C.c_uint64(sample_offset), int(subsample_steps) if subsample_steps is not None else 0, flags,
                    float(t_min), B)
            finally:
                if nic:
                    _lib.lib.mcvd_model_set_cond_noise(net._model, None, 0, 0, 0)
            _lib.check(rc, "sampler_run")
        net._cond_key = None          # the device loop prepared (and then dropped) its own SPADE cache
        return x.unsqueeze(0)

    # ------------------------------------------------------------------ host loop (reference :262-340 / :138-203)
    steps, alphas, alphas_prev, betas = _subsample(net, subsample_steps)
    if frac_steps is not None and kind == _lib.SAMPLER_DDPM:                       # :250-254
        steps = steps[int((1 - frac_steps) * len(steps)):]
        alphas, alphas_prev, betas = alphas[steps], alphas_prev[steps], betas[steps]
    if same_noise and noise_val is None:
        noise_val = x.detach().clone()                                              # :259-260
    draw = [0]
    fwd_no = [0]
    if gamma:                                                                       # :224-225, :238-240, :255-257
        ks_cum, thetas = net.k_cum.cpu(), net.theta_t.cpu()
        if subsample_steps is not None and subsample_steps < len(net.alphas):
            ks_cum, thetas = ks_cum.index_select(0, steps), thetas.index_select(0, steps)
        if frac_steps is not None:
            ks_cum, thetas = ks_cum[steps], thetas[steps]

    def call_net(xx, labels):
        if nic and cond_noise is not None:
            net.set_next_cond_noise(cond_noise[fwd_no[0]])
        fwd_no[0] += 1
        return net(xx, labels, cond=cond)

    def next_noise(i=None):
        if gamma:           # z = (Gamma(k_i, rate 1/theta_i).sample() - k_i theta_i) / sqrt(1 - alpha_i)       :273-276, :319-322
            k, th = ks_cum[i], thetas[i]
            kt, sd = _f(k * th), _f((1 - alphas[i]).sqrt())
            z = torch.empty_like(x)
            raw = None
            if noise is not None:
                raw = noise[draw[0]].contiguous()
            elif seed is None:                                                      # torch's device sampler, as the reference
                raw = torch.distributions.gamma.Gamma(torch.full(tuple(x.shape[1:]), _f(k), device=dev),
                                                      torch.full(tuple(x.shape[1:]), _f(1 / th), device=dev)).sample((B,)).contiguous()
            with torch.cuda.device(dev):
                net._bind_stream()
                _lib.check(_lib.lib.mcvd_gamma_noise(net._ctx, C.c_void_p(z.data_ptr()), C.c_void_p(raw.data_ptr()) if raw is not None else None,
                                                     _f(k), _f(th), kt, sd, C.c_uint64(seed or 0), C.c_uint64(sample_offset),
                                                     C.c_uint64(draw[0]), B, per), "gamma_noise")
            draw[0] += 1
            return z
        if noise is not None:
            z = noise[draw[0]]
        elif seed is not None:
            z = torch.empty_like(x)
            _lib.check(_lib.lib.mcvd_randn(net._ctx, C.c_void_p(z.data_ptr()), C.c_uint64(seed), C.c_uint64(sample_offset),
                                           C.c_uint64(draw[0]), B, per), "randn")
        else:
            z = torch.randn_like(x)                                                 # device RNG, as the reference
        draw[0] += 1
        return z.contiguous()

    def update(eps, z, c_x0a, c_x0b, c0, c1, cn):
        with torch.cuda.device(dev):
            net._bind_stream()
            _lib.check(_lib.lib.mcvd_sampler_update(
                net._ctx, kind, C.c_void_p(x.data_ptr()), C.c_void_p(eps.data_ptr()),
                C.c_void_p(z.data_ptr()) if z is not None else None, c_x0a, c_x0b, c0, c1, cn, 1 if clip_before else 0,
                x.numel()), "sampler_update")

    images = []
    x_transf = False
    L = len(steps)
    for i, step in enumerate(steps):
        if step < t_min * len(alphas):                                              # :269-270
            continue
        c_beta, c_alpha, c_alpha_prev = betas[i], alphas[i], alphas_prev[i]
        if not x_transf and t_min > 0:                                              # :272-279
            z = next_noise(i)
            x.mul_(_f(c_alpha.sqrt())).add_(z, alpha=_f((1 - c_alpha).sqrt()))
        x_transf = True
        labels = (int(step) * torch.ones(B, device=dev)).long()                     # :283
        grad = call_net(x, labels)                                                  # :284
        c_x0a, c_x0b = _f(1 / c_alpha.sqrt()), _f((1 - c_alpha).sqrt())             # :287
        last_step = i + 1 == L
        if kind == _lib.SAMPLER_DDPM:
            c0 = _f(c_alpha_prev.sqrt() * c_beta / (1 - c_alpha))                   # :290
            c1 = _f((1 - c_beta).sqrt() * (1 - c_alpha_prev) / (1 - c_alpha))
        else:
            c0, c1 = _f(c_alpha_prev.sqrt()), _f((1 - c_alpha_prev).sqrt())         # :168
        need_log = (i == 0 or (i + 1) % max(L // 10, 1) == 0) and (verbose or log)
        add_noise = kind == _lib.SAMPLER_DDPM and not last_step
        # The reference appends/logs x_mod BEFORE adding the step noise (:292-308 precede :324-328), so when something
        # must observe the pre-noise state the update is issued without noise and the noise is added afterwards.
        split = add_noise and (not final_only or need_log)
        z, cn = None, 0.0
        if add_noise:
            z = noise_val if same_noise else next_noise(i)
            cn = _f(c_beta.sqrt()) if just_beta else _f(((1 - c_alpha_prev) / (1 - c_alpha) * c_beta).sqrt())   # :326/:328
        update(grad, None if split else z, c_x0a, c_x0b, c0, c1, 0.0 if split else cn)
        if not final_only:
            images.append(x.to("cpu"))                                              # :292-293
        if need_log:                                                                # :295-308
            g = -1 / (1 - c_alpha).sqrt().item() * grad
            grad_norm = torch.norm(g.reshape(B, -1), dim=-1).mean()
            image_norm = torch.norm(x.reshape(B, -1), dim=-1).mean()
            grad_mean_norm = torch.norm(g.mean(dim=0).reshape(-1)) ** 2 * (1 - c_alpha).item()
            msg = "{}: {}/{}, grad_norm: {}, image_norm: {}, grad_mean_norm: {}".format(
                name, i + 1, L, grad_norm.item(), image_norm.item(), grad_mean_norm.item())
            if verbose:
                print(msg)
            if log:
                logging.info(msg)
        if split:
            x.add_(z, alpha=cn)

    if denoise:                                                                     # :331-335 (label L-1, sic)
        last_noise = ((L - 1) * torch.ones(B, device=dev)).long()
        x = x - _f((1 - alphas[-1]).sqrt()) * call_net(x, last_noise)
        if not final_only:
            images.append(x.to("cpu"))
    _lib.check(_lib.lib.mcvd_ctx_check_range(net._ctx), "sampler (f16x2 range guard)")      # no-op unless the option f16x2 is on
    if final_only:
        return x.unsqueeze(0)
    return torch.stack(images)


def ddpm_sampler(x_mod, scorenet, cond=None, just_beta=False, final_only=False, denoise=True, subsample_steps=None,
                 same_noise=False, noise_val=None, frac_steps=None, verbose=False, log=False, clip_before=True,
                 t_min=-1, gamma=False, **kwargs):
    """Reference: models/__init__.py:206-340.  Extra kwargs understood here: `noise` ([n_draws,B,C,H,W] injected
    sequence), `seed`, `sample_offset` (global index of row 0, for sharded runs)."""
    return _sample(_lib.SAMPLER_DDPM, x_mod, scorenet, cond=cond, just_beta=just_beta, final_only=final_only,
                   denoise=denoise, subsample_steps=subsample_steps, same_noise=same_noise, noise_val=noise_val,
                   frac_steps=frac_steps, verbose=verbose, log=log, clip_before=clip_before, t_min=t_min, gamma=gamma,
                   **kwargs)


def ddim_sampler(x_mod, scorenet, cond=None, final_only=False, denoise=True, subsample_steps=None, verbose=False,
                 log=True, clip_before=True, t_min=-1, gamma=False, **kwargs):
    """Reference: models/__init__.py:102-203."""
    return _sample(_lib.SAMPLER_DDIM, x_mod, scorenet, cond=cond, final_only=final_only, denoise=denoise,
                   subsample_steps=subsample_steps, verbose=verbose, log=log, clip_before=clip_before, t_min=t_min,
                   gamma=gamma, **kwargs)


@torch.no_grad()
def fpndm_sampler(x_mod, scorenet, cond=None, final_only=False, denoise=True, subsample_steps=None, verbose=False, log=True,
                  clip_before=True, t_min=-1, gamma=False, **kwargs):
    """F-PNDM: reference models/__init__.py:38-99 (FPNDM_sampler) with models/pndm.py (gen_order_4 :41-52, runge_kutta :3-17,
    transfer :19-33), behaviour mirrored statement by statement: steps 0, skip, 2*skip, ... with t_next = the previous step
    (-1 first), alpha table = flipped `alphas` indexed by t + 1, network label = t, float midpoint label (t + t_next) / 2;
    Runge-Kutta for the first three steps (4 network evaluations each), 4th-order Adams-Bashforth afterwards.  `denoise`,
    `t_min`, `gamma`, `verbose`, `log` are accepted and unused, as in the reference.  Deterministic (no noise draws).
    The network evaluations are HIP forwards; the combinations and the transfer are the library's mcvd_lincomb /
    mcvd_pndm_transfer kernels (one rounding per operation, in the reference's order)."""
    net = _unwrap(scorenet)
    if subsample_steps is None:
        raise TypeError("FPNDM_sampler needs subsample_steps (the reference divides by it, models/__init__.py:60)")
    net.sync_parameters(force=True)
    dev = net.device
    x = x_mod.to(device=dev, dtype=torch.float32).contiguous().clone()
    if cond is not None:
        cond = cond.to(device=dev, dtype=torch.float32).contiguous()
    B, n = x.shape[0], x.numel()
    d = net._desc
    if x.dim() != 4 or tuple(x.shape[1:]) != (d.channels * d.num_frames, d.image_size, d.image_size):
        raise RuntimeError(f"x_mod has shape {tuple(x.shape)}")
    if (d.num_frames_cond > 0) != (cond is not None) or (cond is not None and tuple(cond.shape) !=
                                                         (B, d.channels * d.num_frames_cond, d.image_size, d.image_size)):
        raise RuntimeError("cond missing or mis-shaped")
    T_all = int(d.num_classes)
    if (T_all - 1) // (T_all // int(subsample_steps)) * (T_all // int(subsample_steps)) + 1 >= T_all:
        # the reference builds alphas.index_select(0, steps + 1) BEFORE its loop (models/__init__.py:74): a step list whose last entry is
        # T - 1 fails there, up front, with torch's IndexError
        raise IndexError("index out of range in self")
    if final_only and not getattr(net, "noise_in_cond", False):
        # the whole loop inside the library (the reference never logs in this sampler, so verbose / log change nothing)
        with torch.cuda.device(dev):
            net._bind_stream()
            _lib.check(_lib.lib.mcvd_fpndm_run(net._model, C.c_void_p(x.data_ptr()), C.c_void_p(cond.data_ptr()) if cond is not None else None,
                                               int(subsample_steps), _lib.FLAG_CLIP_BEFORE if clip_before else 0, B), "fpndm_run")
        net._cond_key = None
        return x.unsqueeze(0)
    _lib.check(_lib.lib.mcvd_ctx_clear_range(net._ctx), "clear_range")
    alphas_old = net.alphas.cpu().flip(0)                                           # :57
    T = len(alphas_old)
    skip = T // subsample_steps                                                     # :60
    steps = list(range(0, T, skip))
    steps_next = [-1] + steps[:-1]                                                  # :62

    def model(xx, t_value, as_float):
        dtype = torch.float32 if as_float else torch.int64
        return net(xx, torch.full((B,), t_value, device=dev, dtype=dtype), cond=cond)

    def transfer(xx, t_idx, tn_idx, et):                                            # pndm.py:19-33
        at, an = alphas_old[int(t_idx) + 1], alphas_old[int(tn_idx) + 1]            # t.long() truncates toward zero, as int()
        d = _f(an - at)
        c1 = _f(1 / (at.sqrt() * (at.sqrt() + an.sqrt())))
        c2 = _f(1 / (at.sqrt() * (((1 - an) * at).sqrt() + ((1 - at) * an).sqrt())))
        out = torch.empty_like(xx)
        with torch.cuda.device(dev):
            net._bind_stream()
            _lib.check(_lib.lib.mcvd_pndm_transfer(net._ctx, C.c_void_p(out.data_ptr()), C.c_void_p(xx.data_ptr()),
                                                   C.c_void_p(et.data_ptr()), d, c1, c2, 1 if clip_before else 0, n),
                       "pndm_transfer")
        return out

    def lincomb(ins, ws, scale):
        out = torch.empty_like(ins[0])
        ptr = [C.c_void_p(t.data_ptr()) for t in ins] + [None] * (4 - len(ins))
        w = list(ws) + [0.0] * (4 - len(ws))
        with torch.cuda.device(dev):
            net._bind_stream()
            _lib.check(_lib.lib.mcvd_lincomb(net._ctx, C.c_void_p(out.data_ptr()), ptr[0], ptr[1], ptr[2], ptr[3],
                                             w[0], w[1], w[2], w[3], scale, len(ins), n), "lincomb")
        return out

    images, ets = [], []
    for i, t in enumerate(steps):
        t_next = steps_next[i]
        t_mid = (t + t_next) / 2                                                    # pndm.py:42: true division
        if len(ets) > 2:                                                            # pndm.py:44-47
            ets.append(model(x, t, False))
            noise = lincomb([ets[-1], ets[-2], ets[-3], ets[-4]], [55.0, -59.0, 37.0, -9.0], _f(torch.tensor(1 / 24)))
            ets = ets[-4:]                                                          # older estimates are never read again
        else:                                                                       # runge_kutta, pndm.py:3-17
            e_1 = model(x, t, False)
            ets.append(e_1)
            e_2 = model(transfer(x, t, t_mid, e_1), t_mid, True)
            e_3 = model(transfer(x, t, t_mid, e_2), t_mid, True)
            e_4 = model(transfer(x, t, t_next, e_3), t_next, False)
            noise = lincomb([e_1, e_2, e_3, e_4], [1.0, 2.0, 2.0, 1.0], _f(torch.tensor(1 / 6)))
        x = transfer(x, t, t_next, noise)                                           # pndm.py:51
        if not final_only:
            images.append(x.to("cpu"))
    _lib.check(_lib.lib.mcvd_ctx_check_range(net._ctx), "FPNDM sampler (f16x2 range guard)")      # no-op unless the option f16x2 is on
    if final_only:
        return x.unsqueeze(0)
    return torch.stack(images)


def get_sampler(config):
    """Reference: runners/ncsn_runner.py:2702-2714 (DDPM / DDIM versions)."""
    version = getattr(config.model, "version", "DDPM").upper()
    if version == "DDPM":
        return partial(ddpm_sampler, config=config)
    if version == "DDIM":
        return partial(ddim_sampler, config=config)
    if version == "FPNDM":
        return partial(fpndm_sampler, config=config)
    raise NotImplementedError(f"sampler version {version} is not on the HIP path (SMLD: out of scope, DESIGN.md section 8)")

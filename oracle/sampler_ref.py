"""CPU ORACLE -- TEST INFRASTRUCTURE ONLY (see oracle/unet_ref.py header).

Restatement of the reference's DDPM / DDIM sampling loops
(models/__init__.py:206-340 `ddpm_sampler`, :102-203 `ddim_sampler`) for the
non-gamma path, with the per-step Gaussian noise supplied by a callable so CPU
and GPU runs can consume the *same* noise sequence.

`fpndm_sample` restates FPNDM_sampler (models/__init__.py:38-99) with models/pndm.py (transfer :19-33, runge_kutta :3-17,
gen_order_4 :41-52), quirks included: the steps run UPWARDS (0, skip, 2*skip, ...) with t_next = the previous step (-1 first),
the alpha table is the flipped `alphas` indexed by t + 1, the network label is t itself, and the Runge-Kutta midpoint label
(t + t_next) / 2 is a float (fractional for the first step).

Parity status: PINNED against the reference samplers by oracle/gen_golden.py
(tests/golden/*_b*.pt, tests/golden/tiny_b3_fpndm.pt).
"""
import numpy as np
import torch


def subsampled_schedule(alphas, alphas_prev, betas, subsample_steps):
    """models/__init__.py:229-237.  Returns (steps, alphas, alphas_prev, betas)."""
    steps = np.arange(len(betas))
    if subsample_steps is not None and subsample_steps < len(alphas):
        skip = len(alphas) // subsample_steps
        steps = torch.tensor(list(range(0, len(alphas), skip)), device=alphas.device)     # :231 (device=alphas.device)
        alphas = alphas.index_select(0, steps)
        alphas_prev = torch.cat([alphas[1:], torch.tensor([1.0]).to(alphas)])
        betas = 1.0 - torch.div(alphas, alphas_prev)
    return steps, alphas, alphas_prev, betas


@torch.no_grad()
def sample(x_mod, scorenet, cond=None, kind="ddpm", just_beta=False, final_only=False, denoise=True,
           subsample_steps=None, clip_before=True, t_min=-1, noise_fn=None, frac_steps=None, gamma=False,
           same_noise=False, noise_val=None):
    """kind='ddpm': models/__init__.py:266-333.  kind='ddim': :142-198.
    noise_fn(i, like) -> tensor shaped like `like` (i = -1 for the t_min re-noise draw).
    gamma=True (:224-225, :238-240, :273-276, :319-322): noise_fn returns the RAW draw g ~ Gamma(k_cum[i], rate 1/theta[i]) and the
    loop standardises it, z = (g - k theta) / sqrt(1 - alpha_i), as the reference does.
    same_noise / noise_val (:259-260, :316-317): every step adds the SAME tensor -- `noise_val`, or a copy of the incoming x_mod when
    none is given -- and nothing is drawn for the steps (the t_min re-noise still draws).
    `images` holds what the reference's list holds on an accelerator, where `x_mod.to('cpu')` (:293) is a copy taken BEFORE the step
    noise is added in place (:326-328); on a CPU run of the reference `.to('cpu')` is the tensor itself and the in-place `+=` shows
    through (tests that compare against a CPU-generated `final_only=False` fixture add the step noise back)."""
    assert kind in ("ddpm", "ddim")
    if noise_fn is None:
        noise_fn = lambda i, like: torch.randn_like(like)
    steps, alphas, alphas_prev, betas = subsampled_schedule(
        scorenet.alphas, scorenet.alphas_prev, scorenet.betas, subsample_steps)
    # (ddim_sampler reads `gamma` too -- for the t_min re-noise draw only, :144-151; it adds no step noise)
    if gamma:
        ks_cum, thetas = scorenet.k_cum, scorenet.theta_t
        if subsample_steps is not None and subsample_steps < len(scorenet.alphas):
            ks_cum, thetas = ks_cum.index_select(0, steps), thetas.index_select(0, steps)
    if frac_steps is not None and kind == "ddpm":                     # :250-254
        steps = steps[int((1 - frac_steps) * len(steps)):]
        alphas, alphas_prev, betas = alphas[steps], alphas_prev[steps], betas[steps]
        if gamma:
            ks_cum, thetas = ks_cum[steps], thetas[steps]

    if same_noise and noise_val is None:                              # :259-260
        noise_val = x_mod.detach().clone()

    def std(g, i):
        return (g - ks_cum[i] * thetas[i]) / (1 - alphas[i]).sqrt() if gamma else g

    images = []
    started = False
    L = len(steps)
    for i, step in enumerate(steps):
        if step < t_min * len(alphas):                                # :269-270
            continue
        if not started and t_min > 0:                                 # :272-279
            z = std(noise_fn(-1, x_mod), i)
            x_mod = alphas[i].sqrt() * x_mod + (1 - alphas[i]).sqrt() * z
        started = True

        c_beta, c_alpha, c_alpha_prev = betas[i], alphas[i], alphas_prev[i]
        labels = (step * torch.ones(x_mod.shape[0], device=x_mod.device)).long()           # :283
        grad = scorenet(x_mod, labels, cond=cond)                     # :284

        x0 = (1 / c_alpha.sqrt()) * (x_mod - (1 - c_alpha).sqrt() * grad)   # :287
        if clip_before:
            x0 = x0.clip_(-1, 1)                                      # :288-289
        if kind == "ddpm":
            x_mod = (c_alpha_prev.sqrt() * c_beta / (1 - c_alpha)) * x0 \
                + ((1 - c_beta).sqrt() * (1 - c_alpha_prev) / (1 - c_alpha)) * x_mod   # :290
        else:
            x_mod = c_alpha_prev.sqrt() * x0 + (1 - c_alpha_prev).sqrt() * grad        # :168

        if not final_only:
            images.append(x_mod.clone())

        if kind == "ddpm" and i + 1 != L:                             # :311-328
            noise = noise_val if same_noise else std(noise_fn(i, x_mod), i)   # :316-322
            if just_beta:
                x_mod = x_mod + c_beta.sqrt() * noise
            else:
                x_mod = x_mod + ((1 - c_alpha_prev) / (1 - c_alpha) * c_beta).sqrt() * noise

    if denoise:                                                       # :331-335 (label L-1, sic)
        last = ((L - 1) * torch.ones(x_mod.shape[0], device=x_mod.device)).long()
        x_mod = x_mod - (1 - alphas[-1]).sqrt() * scorenet(x_mod, last, cond=cond)
        if not final_only:
            images.append(x_mod.clone())

    if final_only:
        return x_mod.unsqueeze(0)
    return torch.stack(images)


def pndm_transfer(x, t, t_next, et, alphas_cump, clip_before):
    """models/pndm.py:19-33 (t, t_next: [B] tensors, possibly float; the table index is t.long() + 1)."""
    at = alphas_cump[t.long() + 1].view(-1, 1, 1, 1)
    at_next = alphas_cump[t_next.long() + 1].view(-1, 1, 1, 1)
    x_delta = (at_next - at) * ((1 / (at.sqrt() * (at.sqrt() + at_next.sqrt()))) * x
                                - 1 / (at.sqrt() * (((1 - at_next) * at).sqrt() + ((1 - at) * at_next).sqrt())) * et)
    x_next = x + x_delta
    if clip_before:
        x_next = x_next.clip_(-1, 1)
    return x_next


@torch.no_grad()
def fpndm_sample(x_mod, scorenet, cond=None, final_only=False, subsample_steps=None, clip_before=True):
    """FPNDM_sampler, models/__init__.py:38-99 (denoise / t_min / gamma are accepted and ignored there)."""
    alphas = scorenet.alphas
    alphas_old = alphas.flip(0)                                       # :57
    skip = len(alphas) // subsample_steps                             # :60
    steps = list(range(0, len(alphas), skip))
    steps_next = [-1] + steps[:-1]                                    # :62
    model = lambda x, t: scorenet(x, t, cond=cond)                    # :79
    B = x_mod.shape[0]
    images, ets = [], []
    for i in range(len(steps)):
        t = (steps[i] * torch.ones(B)).long()                         # :85
        t_next = (steps_next[i] * torch.ones(B)).long()
        t_list = [t, (t + t_next) / 2, t_next]                        # pndm.py:42 (true division: float labels)
        if len(ets) > 2:                                              # pndm.py:44-47
            noise_ = model(x_mod, t)
            ets.append(noise_)
            noise = (1 / 24) * (55 * ets[-1] - 59 * ets[-2] + 37 * ets[-3] - 9 * ets[-4])
        else:                                                         # runge_kutta, pndm.py:3-17
            e_1 = model(x_mod, t_list[0])
            ets.append(e_1)
            x_2 = pndm_transfer(x_mod, t_list[0], t_list[1], e_1, alphas_old, clip_before)
            e_2 = model(x_2, t_list[1])
            x_3 = pndm_transfer(x_mod, t_list[0], t_list[1], e_2, alphas_old, clip_before)
            e_3 = model(x_3, t_list[1])
            x_4 = pndm_transfer(x_mod, t_list[0], t_list[2], e_3, alphas_old, clip_before)
            e_4 = model(x_4, t_list[2])
            noise = (1 / 6) * (e_1 + 2 * e_2 + 2 * e_3 + e_4)
        x_mod = pndm_transfer(x_mod, t, t_next, noise, alphas_old, clip_before)   # pndm.py:51
        if not final_only:
            images.append(x_mod.to("cpu"))
    if final_only:
        return x_mod.unsqueeze(0)
    return torch.stack(images)

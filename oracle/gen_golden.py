"""Golden-vector generator -- TEST INFRASTRUCTURE ONLY.

Runs the REAL reference (/root/reference, imported read-only; exists only in the
build container) on the synthetic weights/inputs of oracle/synth.py and writes
small fixtures to tests/golden/.  The GPU box has no /root/reference: tests read
only the committed fixtures.

    PYTHONDONTWRITEBYTECODE=1 python -m oracle.gen_golden

What is pinned:
  * state_dict names/shapes of UNetMore_DDPM == unet_ref.param_shapes       
  * schedule buffers betas/alphas/alphas_prev                               (ncsnpp_more.py:735-743)
  * one UNet forward: final eps + a strided probe of every module's output  (ncsnpp_more.py:251-392, 590-718)
  * ddpm_sampler / ddim_sampler end-to-end with an injected noise sequence  (models/__init__.py:102-340)
  * upfirdn2d_native for the two FIR uses + a generic case                  (op/upfirdn2d.py:163-204)
  * FPNDM_sampler end-to-end, every step (deterministic, no noise)          (models/__init__.py:38-99, models/pndm.py)
  * one forward at BASELINE configs 3-5 (probes of eps and of every module)  (ncsnpp_more.py:251-392, layerspp.py:101-173)
"""
import os
import sys

os.environ.setdefault("PYTHONDONTWRITEBYTECODE", "1")
sys.dont_write_bytecode = True

import numpy as np
import torch

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(os.path.dirname(HERE), "tests", "golden")

from oracle import synth, unet_ref  # noqa: E402


def probe(t, n=97):
    """Deterministic strided sample + moments of a tensor (keeps fixtures small)."""
    f = t.detach().reshape(-1).double()
    idx = torch.linspace(0, f.numel() - 1, min(n, f.numel())).long()
    return dict(shape=list(t.shape), mean=f.mean().item(), absmean=f.abs().mean().item(),
                sample=f[idx].float().clone(), idx=idx.clone())


def build_ref_net(config):
    sys.path.insert(0, REF)
    from models.better.ncsnpp_more import UNetMore_DDPM
    config.device = "cpu"
    net = UNetMore_DDPM(config).eval()
    return net


def check_names(net, config):
    want = unet_ref.param_shapes(unet_ref.hot_cfg(config))
    have = {k: tuple(v.shape) for k, v in net.named_parameters()}
    assert list(have.keys()) == list(want.keys()), (set(have) ^ set(want))
    assert have == want
    bufs = [k for k, _ in net.named_buffers()]
    assert sorted(bufs) == sorted(["betas", "alphas", "alphas_prev", "unet.sigmas"]), bufs


class NoiseInjector:
    """Replace torch.randn_like (as seen from the reference's `models` module) by a
    pre-drawn sequence: call k returns noise[k]."""

    def __init__(self, noise):
        self.noise, self.k = noise, 0

    def __call__(self, like, *a, **kw):
        z = self.noise[self.k].to(like)
        self.k += 1
        assert z.shape == like.shape
        return z


def _drift64(config, sd, x, cond, noise, kind, subsample, extra, res32):
    """max |reference fp32 result - the same sampler call in float64| (oracle restatement, same injected noise)."""
    from oracle import sampler_ref
    net64 = unet_ref.OracleScoreNet(config, sd, dtype=torch.float64)
    k = [0]

    def fn(i, like):
        k[0] += 1
        return noise[k[0] - 1].to(like.dtype)
    res64 = sampler_ref.sample(x.double().clone(), net64, cond=cond.double() if cond is not None else None, kind=kind, final_only=True,
                               denoise=True, subsample_steps=subsample, clip_before=True, noise_fn=fn, **extra)
    return float((res32.double() - res64).abs().max())


def gen_model_case(name, batch, steps_kinds):
    import models as ref_models
    config = synth.make_config(name)
    net = build_ref_net(config)
    check_names(net, config)
    sd = synth.make_state_dict(config, seed=123)
    missing = net.load_state_dict(sd, strict=False)
    assert not missing.unexpected_keys and set(missing.missing_keys) <= {"betas", "alphas", "alphas_prev", "unet.sigmas"}
    x, cond = synth.make_inputs(config, batch, seed=0)
    out = dict(config_name=name, batch=batch)
    out["betas"], out["alphas"], out["alphas_prev"] = net.betas.clone(), net.alphas.clone(), net.alphas_prev.clone()

    # ---- one forward with per-module probes
    taps = {}
    hooks = []
    for i, m in enumerate(net.unet.all_modules):
        hooks.append(m.register_forward_hook(lambda mod, inp, o, i=i: taps.__setitem__(i, probe(o))))
    t = torch.tensor([(37 * (b + 1)) % 1000 for b in range(batch)]).long()   # distinct labels per row
    with torch.no_grad():
        eps = net(x, t, cond=cond)
    for h in hooks:
        h.remove()
    out["fwd_t"], out["fwd_eps"], out["fwd_taps"] = t, eps.clone(), taps

    # ---- samplers with injected noise
    for kind, subsample, extra in steps_kinds:
        sampler = dict(ddpm=ref_models.ddpm_sampler, ddim=ref_models.ddim_sampler)[kind]
        noise = synth.make_noise(config, batch, subsample + 1, seed=2)
        inj = NoiseInjector(noise)
        orig = torch.randn_like
        torch.randn_like = inj
        try:
            res = sampler(x.clone(), net, cond=cond, final_only=True, denoise=True, subsample_steps=subsample,
                          clip_before=True, verbose=False, log=False, **extra)
        finally:
            torch.randn_like = orig
        key = f"{kind}_{subsample}" + ("".join(f"_{k}{v}" for k, v in extra.items()) if extra else "")
        out["sampler_" + key] = dict(result=res.clone(), n_noise=inj.k)
        if kind == "ddim":
            # round 5: the noise floor the loosened DDIM gate stands on travels WITH the fixture -- the same call evaluated in
            # float64 by the oracle restatement (pinned to the reference per forward; the reference hard-codes float32 in its embedding)
            out["sampler_" + key]["ref32_vs_ref64_max_abs"] = _drift64(config, sd, x, cond, noise, kind, subsample, extra, res)
            print(f"  {name}: {key}: reference fp32 vs fp64 evaluation {out['sampler_' + key]['ref32_vs_ref64_max_abs']:.3e}")
        print(f"  {name}: {key}: out range [{res.min():.4f}, {res.max():.4f}], noise draws {inj.k}")
    os.makedirs(OUT, exist_ok=True)
    torch.save(out, os.path.join(OUT, f"{name}_b{batch}.pt"))
    print(f"wrote {name}_b{batch}.pt  eps std {eps.std():.4f}")


def gen_forward_only(name, batch):
    """One forward of the REAL reference at a full-width BASELINE config (ngf=128 / SPADE / 128x128 five-level): the final
    eps as a strided probe + moments, and a probe of every module output.  Small file, pins the oracle at those widths."""
    config = synth.make_config(name)
    net = build_ref_net(config)
    check_names(net, config)
    net.load_state_dict(synth.make_state_dict(config, seed=123), strict=False)
    x, cond = synth.make_inputs(config, batch, seed=0)
    taps, hooks = {}, []
    for i, m in enumerate(net.unet.all_modules):
        hooks.append(m.register_forward_hook(lambda mod, inp, o, i=i: taps.__setitem__(i, probe(o, 41))))
    t = torch.tensor([(311 * (b + 1)) % 1000 for b in range(batch)]).long()
    with torch.no_grad():
        eps = net(x, t, cond=cond)
    for h in hooks:
        h.remove()
    torch.save(dict(config_name=name, batch=batch, fwd_t=t, fwd_eps_probe=probe(eps, 4001), fwd_taps=taps),
               os.path.join(OUT, f"{name}_b{batch}_fwd.pt"))
    print(f"wrote {name}_b{batch}_fwd.pt  eps std {eps.std():.4f}")


def gen_fpndm(name="tiny", batch=3, subsample=10):
    """FPNDM_sampler of the real reference (models/__init__.py:38-99) on the synthetic tiny model: every step's x."""
    sys.path.insert(0, REF)
    import models as ref_models
    config = synth.make_config(name)
    net = build_ref_net(config)
    net.load_state_dict(synth.make_state_dict(config, seed=123), strict=False)
    x, cond = synth.make_inputs(config, batch, seed=0)
    res = ref_models.FPNDM_sampler(x.clone(), net, cond=cond, final_only=False, subsample_steps=subsample, clip_before=True,
                                   verbose=False, log=False)
    fin = ref_models.FPNDM_sampler(x.clone(), net, cond=cond, final_only=True, subsample_steps=subsample, clip_before=False,
                                   verbose=False, log=False)
    torch.save(dict(config_name=name, batch=batch, subsample=subsample, all_clip=res.clone(), final_noclip=fin.clone()),
               os.path.join(OUT, f"{name}_b{batch}_fpndm.pt"))
    print(f"wrote {name}_b{batch}_fpndm.pt  steps {res.shape[0]}  range [{res.min():.4f}, {res.max():.4f}]  "
          f"noclip range [{fin.min():.4f}, {fin.max():.4f}]")


def gen_fpndm_wide(name, batch, subsample):
    """FPNDM_sampler of the REAL reference (models/__init__.py:38-99, models/pndm.py) at a full-width BASELINE config: final frames of the
    clipped run, plus the fp32-vs-fp64 distance of the same call on the oracle restatement (the sampler is deterministic and its
    linear multistep combination amplifies forward rounding: the tolerance of a test on this fixture stands on that number)."""
    import models as ref_models
    from oracle import sampler_ref
    config = synth.make_config(name)
    net = build_ref_net(config)
    net.load_state_dict(synth.make_state_dict(config, seed=123), strict=False)
    x, cond = synth.make_inputs(config, batch, seed=0)
    res = ref_models.FPNDM_sampler(x.clone(), net, cond=cond, final_only=True, subsample_steps=subsample, clip_before=True,
                                   verbose=False, log=False)
    sd = synth.make_state_dict(config, seed=123)
    o32 = sampler_ref.fpndm_sample(x.clone(), unet_ref.OracleScoreNet(config, sd), cond=cond, final_only=True,
                                   subsample_steps=subsample, clip_before=True)
    o64 = sampler_ref.fpndm_sample(x.double().clone(), unet_ref.OracleScoreNet(config, sd, dtype=torch.float64), cond=cond.double(),
                                   final_only=True, subsample_steps=subsample, clip_before=True)
    drift = float((res.double() - o64.double()).abs().max())
    print(f"  reference fp32 vs oracle fp64: {drift:.3e}; oracle fp32 vs reference fp32: {float((o32 - res).abs().max()):.3e}")
    torch.save(dict(config_name=name, batch=batch, subsample=subsample, result=res.clone(), ref32_vs_ref64_max_abs=drift),
               os.path.join(OUT, f"{name}_b{batch}_fpndm{subsample}.pt"))
    print(f"wrote {name}_b{batch}_fpndm{subsample}.pt  shape {tuple(res.shape)}  range [{res.min():.4f}, {res.max():.4f}]")


def gen_sampler_only(name, batch, subsample, kind="ddpm", measure_drift=False):
    """Full `ddpm_sampler` of the REAL reference at a full-width BASELINE config with the injected noise sequence: the final
    frames (full tensor, small) -- pins the oracle and the HIP path end-to-end at configs 3 / 4."""
    import models as ref_models
    config = synth.make_config(name)
    net = build_ref_net(config)
    net.load_state_dict(synth.make_state_dict(config, seed=123), strict=False)
    x, cond = synth.make_inputs(config, batch, seed=0)
    noise = synth.make_noise(config, batch, subsample + 1, seed=2)
    inj = NoiseInjector(noise)
    orig = torch.randn_like
    torch.randn_like = inj
    try:
        res = dict(ddpm=ref_models.ddpm_sampler, ddim=ref_models.ddim_sampler)[kind](
            x.clone(), net, cond=cond, final_only=True, denoise=True, subsample_steps=subsample, clip_before=True,
            verbose=False, log=False)
    finally:
        torch.randn_like = orig
    extra = {}
    if measure_drift:
        # the same call in float64 (the oracle restatement, itself pinned to the reference at 2e-6 per forward: the reference hard-codes
        # float32 in its timestep embedding and cannot run in double): |fp32 - fp64| is the noise floor a tolerance on this fixture has to
        # stand on -- a deterministic sampler carries forward rounding through all of its steps
        from oracle import sampler_ref
        net64 = unet_ref.OracleScoreNet(config, synth.make_state_dict(config, seed=123), dtype=torch.float64)
        k = [0]

        def fn(i, like):
            k[0] += 1
            return noise[k[0] - 1].to(like.dtype)
        res64 = sampler_ref.sample(x.double().clone(), net64, cond=cond.double(), kind=kind, final_only=True, denoise=True,
                                   subsample_steps=subsample, clip_before=True, noise_fn=fn)
        extra["ref32_vs_ref64_max_abs"] = float((res.double() - res64.double()).abs().max())
        print(f"  reference fp32 vs fp64 evaluation: {extra['ref32_vs_ref64_max_abs']:.3e}")
    torch.save(dict(config_name=name, batch=batch, subsample=subsample, kind=kind, result=res.clone(), n_noise=inj.k, **extra),
               os.path.join(OUT, f"{name}_b{batch}_{kind}{subsample}.pt"))
    print(f"wrote {name}_b{batch}_{kind}{subsample}.pt  range [{res.min():.4f}, {res.max():.4f}]  draws {inj.k}")


def gen_autoregressive(name, batch, nfp, subsample):
    """The autoregressive block loop of NCSNRunner.video_gen (runners/ncsn_runner.py:1504-1569, future == 0, no data_init,
    init_prev_t <= 0) restated around the REAL reference `ddpm_sampler` + `UNetMore_DDPM` (the runner module itself does not
    import here: cv2 / imageio / torchvision are absent).  Per block: fresh init z (seed 50 + block), injected step noise
    (seed 60 + block); cond <- cat(cond[:, C*nf:], gen[:, C*max(0, nf - nc):]) (:1537-1539); result = cat(blocks)[:, :C*nfp] (:1569)."""
    import models as ref_models
    from math import ceil
    config = synth.make_config(name)
    net = build_ref_net(config)
    net.load_state_dict(synth.make_state_dict(config, seed=123), strict=False)
    C, nf, nc, S = config.data.channels, config.data.num_frames, config.data.num_frames_cond, config.data.image_size
    _, cond = synth.make_inputs(config, batch, seed=0)
    n_iter = ceil(nfp / nf)
    preds = []
    for i in range(n_iter):
        init = torch.randn(batch, C * nf, S, S, generator=torch.Generator().manual_seed(50 + i))
        inj = NoiseInjector(synth.make_noise(config, batch, subsample + 1, seed=60 + i))
        orig = torch.randn_like
        torch.randn_like = inj
        try:
            gen = ref_models.ddpm_sampler(init, net, cond=cond, cond_mask=None, n_steps_each=0, step_lr=0.0, verbose=False,
                                          final_only=True, denoise=True, subsample_steps=subsample, clip_before=True,
                                          t_min=-1, log=False, gamma=False)
        finally:
            torch.randn_like = orig
        gen = gen[-1].reshape(batch, C * nf, S, S)                                    # :1521-1522
        preds.append(gen)
        if i == n_iter - 1:
            continue
        cond = torch.cat([cond[:, C * nf:], gen[:, C * max(0, nf - nc):]], dim=1)     # :1537-1539
    pred = torch.cat(preds, dim=1)[:, :C * nfp]                                       # :1569
    # round 5: the fp32-vs-fp64 distance of the SAME chained call (oracle restatement in float64, same inits and noise): what the
    # 2e-4 gate of the two-block tests stands on
    from oracle import sampler_ref
    sd = synth.make_state_dict(config, seed=123)
    net64 = unet_ref.OracleScoreNet(config, sd, dtype=torch.float64)
    _, cond64 = synth.make_inputs(config, batch, seed=0)
    cond64 = cond64.double()
    preds64 = []
    for i in range(n_iter):
        init = torch.randn(batch, C * nf, S, S, generator=torch.Generator().manual_seed(50 + i)).double()
        nz = synth.make_noise(config, batch, subsample + 1, seed=60 + i)
        k = [0]

        def fn(j, like, nz=nz, k=k):
            k[0] += 1
            return nz[k[0] - 1].to(like.dtype)
        g64 = sampler_ref.sample(init, net64, cond=cond64, kind="ddpm", final_only=True, denoise=True, subsample_steps=subsample,
                                 clip_before=True, noise_fn=fn)[-1].reshape(batch, C * nf, S, S)
        preds64.append(g64)
        if i != n_iter - 1:
            cond64 = torch.cat([cond64[:, C * nf:], g64[:, C * max(0, nf - nc):]], dim=1)
    drift = float((pred.double() - torch.cat(preds64, dim=1)[:, :C * nfp]).abs().max())
    print(f"  reference fp32 vs fp64 evaluation of the {n_iter}-block chain: {drift:.3e}")
    torch.save(dict(config_name=name, batch=batch, nfp=nfp, subsample=subsample, pred=pred.clone(), ref32_vs_ref64_max_abs=drift),
               os.path.join(OUT, f"{name}_b{batch}_ar{nfp}.pt"))
    print(f"wrote {name}_b{batch}_ar{nfp}.pt  blocks {n_iter}  range [{pred.min():.4f}, {pred.max():.4f}]")


def gen_fir():
    sys.path.insert(0, REF)
    from models.better import up_or_down_sampling as uds
    from models.better.op.upfirdn2d import upfirdn2d_native
    g = torch.Generator().manual_seed(5)
    x = torch.randn(2, 3, 8, 8, generator=g)
    k = torch.tensor(uds._setup_kernel([1, 3, 3, 1]))
    out = dict(x=x,
               up=uds.upsample_2d(x, [1, 3, 3, 1], factor=2),
               down=uds.downsample_2d(x, [1, 3, 3, 1], factor=2),
               kernel=k,
               generic=upfirdn2d_native(x, k * 3.0, 3, 3, 2, 2, 2, 1, 2, 1),   # up 3, down 2, pad (2,1)
               generic_args=dict(up=3, down=2, pad0=2, pad1=1, gain=3.0))
    torch.save(out, os.path.join(OUT, "fir.pt"))
    print("wrote fir.pt")


def main():
    torch.manual_seed(0)
    torch.set_num_threads(os.cpu_count())
    os.makedirs(OUT, exist_ok=True)
    gen_fir()
    gen_model_case("tiny", 3, [("ddpm", 10, {}), ("ddim", 10, {}), ("ddpm", 10, dict(t_min=0.35))])
    gen_model_case("tiny_spade", 2, [("ddpm", 10, {})])
    gen_model_case("smmnist_big5", 2, [("ddpm", 100, {})])          # BASELINE config 1 (plumbing, CPU)
    gen_fpndm()
    for name, batch in (("kth64_big_ngf128", 2), ("bair_big_spade", 2), ("cityscapes_big", 1)):     # BASELINE configs 3-5
        gen_forward_only(name, batch)


class SiteInjector:
    """torch.randn_like replacement that serves two pre-drawn sequences by CALL SITE: draws made inside the network's forward
    (ncsnpp_more.py:766, the noise_in_cond branch) come from `cond_seq`, the samplers' own draws from `step_seq`."""

    def __init__(self, step_seq, cond_seq):
        self.step_seq, self.cond_seq, self.ks, self.kc = step_seq, cond_seq, 0, 0

    def __call__(self, like, *a, **kw):
        if sys._getframe(1).f_code.co_filename.endswith("ncsnpp_more.py"):
            z = self.cond_seq[self.kc]
            self.kc += 1
        else:
            z = self.step_seq[self.ks]
            self.ks += 1
        assert z.shape == like.shape, (z.shape, like.shape)
        return z.to(like)


def gen_f4():
    """SURVEY 8f rank 4 flags on the real reference (tiny nets): cond_emb (+ cond_mask), noise_in_cond (concat and SPADE), gamma
    (sampler + noise_in_cond, with the Gamma sampler replaced by a deterministic stand-in so the draws can be injected),
    output_all_frames (the reference's own failure)."""
    import models as ref_models
    import torch.distributions.gamma as tdg
    # ---- cond_emb
    name, B = "tiny_condemb", 3
    config = synth.make_config(name)
    net = build_ref_net(config)
    check_names(net, config)
    net.load_state_dict(synth.make_state_dict(config, seed=123), strict=False)
    x, cond = synth.make_inputs(config, B, seed=0)
    t = torch.tensor([(37 * (b + 1)) % 1000 for b in range(B)]).long()
    mask = torch.tensor([1, 0, 1], dtype=torch.int32)
    taps, hooks = {}, []
    for i, m in enumerate(net.unet.all_modules):
        hooks.append(m.register_forward_hook(lambda mod, inp, o, i=i: taps.__setitem__(i, probe(o))))
    with torch.no_grad():
        eps_mask = net(x, t, cond=cond, cond_mask=mask)
        taps_mask = dict(taps)
        eps_none = net(x, t, cond=cond)
    for h in hooks:
        h.remove()
    noise = synth.make_noise(config, B, 11, seed=2)
    inj = NoiseInjector(noise)
    orig = torch.randn_like
    torch.randn_like = inj
    try:
        res = ref_models.ddpm_sampler(x.clone(), net, cond=cond, final_only=True, denoise=True, subsample_steps=10, clip_before=True,
                                      verbose=False, log=False, cond_mask=mask)       # cond_mask lands in **kwargs and is dropped (:263)
    finally:
        torch.randn_like = orig
    torch.save(dict(config_name=name, batch=B, fwd_t=t, mask=mask, eps_mask=eps_mask.clone(), eps_none=eps_none.clone(),
                    fwd_taps=taps_mask, sampler=res.clone(), n_noise=inj.k), os.path.join(OUT, "tiny_condemb_b3.pt"))
    print(f"wrote tiny_condemb_b3.pt  |eps_mask - eps_none| {float((eps_mask - eps_none).abs().max()):.3f}")

    # ---- noise_in_cond (concat and SPADE)
    for name in ("tiny_noisecond", "tiny_spade_noisecond"):
        B = 2
        config = synth.make_config(name)
        net = build_ref_net(config)
        check_names(net, config)
        net.load_state_dict(synth.make_state_dict(config, seed=123), strict=False)
        x, cond = synth.make_inputs(config, B, seed=0)
        t = torch.tensor([700, 20]).long()
        g = torch.Generator().manual_seed(77)
        cond_seq = torch.randn(12, *cond.shape, generator=g)
        step_seq = synth.make_noise(config, B, 11, seed=2)
        inj = SiteInjector(step_seq, cond_seq)
        orig = torch.randn_like
        torch.randn_like = inj
        try:
            with torch.no_grad():
                eps = net(x, t, cond=cond)                               # consumes cond_seq[0]
            res = ref_models.ddpm_sampler(x.clone(), net, cond=cond, final_only=True, denoise=True, subsample_steps=10,
                                          clip_before=True, verbose=False, log=False)      # cond_seq[1..11], step_seq[0..8]
        finally:
            torch.randn_like = orig
        assert inj.kc == 12 and inj.ks == 9, (inj.kc, inj.ks)
        torch.save(dict(config_name=name, batch=B, fwd_t=t, fwd_eps=eps.clone(), cond_seq=cond_seq, sampler=res.clone()),
                   os.path.join(OUT, f"{name}_b2.pt"))
        print(f"wrote {name}_b2.pt  sampler range [{res.min():.3f}, {res.max():.3f}]")

    # ---- gamma (sampler draws + noise_in_cond draws), Gamma.sample replaced by a deterministic stand-in
    name, B = "tiny_gamma", 2
    config = synth.make_config(name)
    net = build_ref_net(config)
    net.load_state_dict(synth.make_state_dict(config, seed=123), strict=False)
    x, cond = synth.make_inputs(config, B, seed=0)
    normals = torch.randn(40, *x.shape, generator=torch.Generator().manual_seed(91))
    log_raw = []

    class FakeGamma:
        n = 0

        def __init__(self, concentration, rate):
            self.c, self.r = concentration, rate

        def sample(self, shape=()):
            c = self.c.expand(tuple(shape) + tuple(self.c.shape)) if len(shape) else self.c
            r = self.r.expand(tuple(shape) + tuple(self.r.shape)) if len(shape) else self.r
            g = c / r + c.sqrt() / r * normals[FakeGamma.n].reshape(c.shape)      # mean k theta, variance k theta^2
            site = "cond" if sys._getframe(1).f_code.co_filename.endswith("ncsnpp_more.py") else "step"
            log_raw.append((site, g.clone()))
            FakeGamma.n += 1
            return g
    o1, o2 = ref_models.Gamma, tdg.Gamma
    ref_models.Gamma = FakeGamma
    tdg.Gamma = FakeGamma
    torch.distributions.gamma.Gamma = FakeGamma
    try:
        res = ref_models.ddpm_sampler(x.clone(), net, cond=cond, final_only=True, denoise=True, subsample_steps=10, clip_before=True,
                                      verbose=False, log=False, gamma=True)
        res_tmin = ref_models.ddpm_sampler(x.clone(), net, cond=cond, final_only=True, denoise=True, subsample_steps=10,
                                           clip_before=True, verbose=False, log=False, gamma=True, t_min=0.35)
    finally:
        ref_models.Gamma, tdg.Gamma = o1, o2
        torch.distributions.gamma.Gamma = o2
    # first call: per step [cond, step] (no step draw after the last step), then the denoise forward's cond draw
    n1 = 11 + 9
    first, second = log_raw[:n1], log_raw[n1:]

    def split(log, labels):
        """raw step draws; conditioning draws standardised with the label's tables, as ncsnpp_more.py:761-765 does."""
        steps_raw = torch.stack([g for s_, g in log if s_ == "step"])
        conds = [g for s_, g in log if s_ == "cond"]
        assert len(conds) == len(labels)
        zc = []
        for g, lab in zip(conds, labels):
            k, th, a = net.k_cum[lab], net.theta_t[lab], net.alphas[lab]
            zc.append((g - k * th) / (1 - a).sqrt())
        return steps_raw, torch.stack(zc)
    steps10 = list(range(0, 1000, 100))
    s1, c1 = split(first, steps10 + [9])                                    # denoise label L - 1 = 9 (sic)
    kept = [t for t in steps10 if not (t < 0.35 * 10)]                       # step < t_min * len(alphas_subsampled): none skipped but 0..3?
    kept = [t for t in steps10 if not (t < 0.35 * len(steps10))]
    s2, c2 = split(second, kept + [9])
    torch.save(dict(config_name=name, batch=B, sampler=res.clone(), step_raw=s1, cond_z=c1, sampler_tmin=res_tmin.clone(),
                    step_raw_tmin=s2, cond_z_tmin=c2, k_cum=net.k_cum.clone(), theta_t=net.theta_t.clone()),
               os.path.join(OUT, "tiny_gamma_b2.pt"))
    print(f"wrote tiny_gamma_b2.pt  range [{res.min():.3f}, {res.max():.3f}]  draws {len(first)} + {len(second)}")

    # ---- output_all_frames: the reference itself cannot run it with cond (records the failure)
    config = synth.make_config("tiny_allframes")
    net = build_ref_net(config)
    net.load_state_dict(synth.make_state_dict(config, seed=123), strict=False)
    x, cond = synth.make_inputs(config, 2, seed=0)
    try:
        with torch.no_grad():
            net(x, torch.tensor([5, 6]), cond=cond)
        msg = None
    except RuntimeError as e:
        msg = str(e)
    assert msg is not None, "reference output_all_frames unexpectedly works"
    torch.save(dict(config_name="tiny_allframes", error=msg), os.path.join(OUT, "tiny_allframes_err.pt"))
    print("wrote tiny_allframes_err.pt:", msg[:100])


def gen_gamma_ddim():
    """Round 6: `ddim_sampler(..., gamma=True, t_min > 0)` on a model.gamma net -- the one place the DDIM sampler reads its `gamma` kwarg: the
    re-noise of the first executed step draws a standardised Gamma variate instead of a normal one (models/__init__.py:144-151), and the runner
    passes `gamma=config.model.gamma` to whichever sampler it bound (runners/ncsn_runner.py:1518).  Same deterministic stand-in for the Gamma
    sampler as gen_f4 (the raw draws are recorded by call site: the sampler's own draw, and the noise_in_cond draws inside the network)."""
    import models as ref_models
    import torch.distributions.gamma as tdg
    name, B = "tiny_gamma", 2
    config = synth.make_config(name)
    net = build_ref_net(config)
    net.load_state_dict(synth.make_state_dict(config, seed=123), strict=False)
    x, cond = synth.make_inputs(config, B, seed=0)
    normals = torch.randn(24, *x.shape, generator=torch.Generator().manual_seed(92))
    log_raw = []

    class FakeGamma:
        n = 0

        def __init__(self, concentration, rate):
            self.c, self.r = concentration, rate

        def sample(self, shape=()):
            c = self.c.expand(tuple(shape) + tuple(self.c.shape)) if len(shape) else self.c
            r = self.r.expand(tuple(shape) + tuple(self.r.shape)) if len(shape) else self.r
            g = c / r + c.sqrt() / r * normals[FakeGamma.n].reshape(c.shape)
            site = "cond" if sys._getframe(1).f_code.co_filename.endswith("ncsnpp_more.py") else "step"
            log_raw.append((site, g.clone()))
            FakeGamma.n += 1
            return g
    o1, o2 = ref_models.Gamma, tdg.Gamma
    ref_models.Gamma = FakeGamma
    tdg.Gamma = FakeGamma
    torch.distributions.gamma.Gamma = FakeGamma
    try:
        res = ref_models.ddim_sampler(x.clone(), net, cond=cond, final_only=True, denoise=True, subsample_steps=10, clip_before=True,
                                      verbose=False, log=False, gamma=True, t_min=0.35)
    finally:
        ref_models.Gamma, tdg.Gamma = o1, o2
        torch.distributions.gamma.Gamma = o2
    steps10 = list(range(0, 1000, 100))
    kept = [t for t in steps10 if not (t < 0.35 * len(steps10))]
    step_raw = torch.stack([g for s_, g in log_raw if s_ == "step"])
    conds = [g for s_, g in log_raw if s_ == "cond"]
    labels = kept + [9]
    assert len(step_raw) == 1 and len(conds) == len(labels), (len(step_raw), len(conds))
    zc = []
    for g, lab in zip(conds, labels):                                  # standardised as ncsnpp_more.py:761-765 does
        k, th, a = net.k_cum[lab], net.theta_t[lab], net.alphas[lab]
        zc.append((g - k * th) / (1 - a).sqrt())
    torch.save(dict(config_name=name, batch=B, sampler_tmin=res.clone(), step_raw_tmin=step_raw, cond_z_tmin=torch.stack(zc),
                    k_cum=net.k_cum.clone(), theta_t=net.theta_t.clone()), os.path.join(OUT, "tiny_gamma_ddim_b2.pt"))
    print(f"wrote tiny_gamma_ddim_b2.pt  range [{res.min():.3f}, {res.max():.3f}]  draws: {len(step_raw)} step + {len(conds)} cond")


def gen_init_moments():
    """Round 5: per-parameter moments of the REAL reference's construction-time initialisation
    (`UNetMore_DDPM(config)` untouched: models/better/layers.py:43-80 `default_init`, torch defaults elsewhere), for
    `HipScoreNet.reset_parameters` -- same distributions, not the same draws."""
    out = {}
    for name in ("tiny", "tiny_spade", "tiny_condemb", "smmnist_big5"):
        torch.manual_seed(7)
        net = build_ref_net(synth.make_config(name))
        out[name] = {k: dict(shape=list(v.shape), min=float(v.min()), max=float(v.max()), mean=float(v.double().mean()),
                             std=float(v.double().std()) if v.numel() > 1 else 0.0) for k, v in net.named_parameters()}
        nz = sum(1 for v in out[name].values() if max(abs(v["min"]), abs(v["max"])) < 1e-6)
        print(f"  {name}: {len(out[name])} parameter tensors, {nz} with |max| < 1e-6 (SURVEY 9.6-1)")
    torch.save(out, os.path.join(OUT, "init_moments.pt"))
    print("wrote init_moments.pt")


SURFACE_CASES = [
    # key, kind, kwargs of the reference call (final_only=True, denoise=True, clip_before=True unless stated), pre-drawn noise tensors
    ("ddpm_full1000", "ddpm", dict(subsample_steps=1000), 1000),          # `subsample_steps < len(alphas)` false (:229-230): tables as they are
    ("ddpm_fullNone", "ddpm", dict(subsample_steps=None), 1000),          # :228 `subsample_steps is not None` false
    ("ddim_full1000", "ddim", dict(subsample_steps=1000), 1),
    ("ddim_fullNone_tmin0.9", "ddim", dict(subsample_steps=None, t_min=0.9), 2),                       # DDIM, tables as they are, 100 steps + re-noise
    ("ddpm_full1000_tmin0.9_just_beta", "ddpm", dict(subsample_steps=1000, t_min=0.9, just_beta=True), 101),   # table betas + sqrt(beta) noise
    ("ddpm_fullNone_tmin0.95", "ddpm", dict(subsample_steps=None, t_min=0.95), 51),
    ("ddpm_10_just_beta", "ddpm", dict(subsample_steps=10, just_beta=True), 11),                       # :325-326
    ("ddpm_tail_just_beta", "ddpm", dict(subsample_steps=1000, frac_steps=0.05, just_beta=True), 60),  # frac_steps + just_beta (host loop)
    ("ddpm_10_same_noise", "ddpm", dict(subsample_steps=10, same_noise=True), 1),                      # :259-260, :316-317 (noise_val = x_mod copy)
    ("ddpm_10_same_noise_val", "ddpm", dict(subsample_steps=10, same_noise=True, noise_val="NOISE_VAL"), 1),
    ("ddpm_10_noise_val_ignored", "ddpm", dict(subsample_steps=10, same_noise=False, noise_val="NOISE_VAL"), 11),   # noise_val without same_noise: unused
    ("ddpm_10_same_noise_tmin", "ddpm", dict(subsample_steps=10, same_noise=True, t_min=0.35), 2),     # the re-noise still DRAWS (:278)
    ("ddpm_frac0.2", "ddpm", dict(subsample_steps=None, frac_steps=0.2), 200),                         # :250-254: the last 200 of 1000 steps
    ("ddpm_frac0.1_tmin0.5", "ddpm", dict(subsample_steps=1000, frac_steps=0.1, t_min=0.5), 101),      # t_min test against the CUT length + re-noise
    ("ddpm_10_nodenoise", "ddpm", dict(subsample_steps=10, denoise=False), 11),                        # :331
    ("ddim_10_nodenoise", "ddim", dict(subsample_steps=10, denoise=False), 1),
    ("ddpm_10_noclip", "ddpm", dict(subsample_steps=10, clip_before=False), 11),                       # :288
    ("ddim_10_noclip", "ddim", dict(subsample_steps=10, clip_before=False), 1),
    ("ddpm_10_images", "ddpm", dict(subsample_steps=10, final_only=False), 11),                        # :292-293, :334-335, :340 (CPU aliasing, see below)
    ("ddim_10_images", "ddim", dict(subsample_steps=10, final_only=False), 1),
    ("ddpm_10_images_nodenoise", "ddpm", dict(subsample_steps=10, final_only=False, denoise=False), 11),
    # the t_min test `step < t_min*len(alphas)` (:269) exactly ON a step: 0.1 * 100 == 10.0 -> step 10 runs (a C float 0.1f * 100 =
    # 10.0000001 would skip it); 0.3 * 100 = 30.000000000000004 in double but the subsampled schedule compares in float32 (0-dim int64
    # tensor vs Python scalar) -> step 30 runs; 0.998 * 1000 == 998.0 on the full schedule (numpy int64: double comparison) -> 998 runs
    ("ddpm_100_tmin0.1", "ddpm", dict(subsample_steps=100, t_min=0.1), 100),
    ("ddim_100_tmin0.3", "ddim", dict(subsample_steps=100, t_min=0.3), 2),
    ("ddpm_full1000_tmin0.998", "ddpm", dict(subsample_steps=1000, t_min=0.998), 3),
    # subsample_steps that does not divide the table length: skip = 1000 // 7 = 142 -> EIGHT steps (0, 142, ..., 994), labels into the full table
    ("ddpm_7", "ddpm", dict(subsample_steps=7), 9),
    ("ddim_7", "ddim", dict(subsample_steps=7), 1),
    ("ddpm_7_images_tmin", "ddpm", dict(subsample_steps=7, final_only=False, t_min=0.2), 9),     # skipped steps leave no image (:269-270 precede :292)
]


def surface_kwargs(kw, config, batch):
    """The fixture stores kwargs with a placeholder for the explicit same-noise tensor; this builds the call's kwargs."""
    out = dict(final_only=True, denoise=True, clip_before=True)
    out.update(kw)
    if out.get("noise_val", None) == "NOISE_VAL":
        out["noise_val"] = synth.make_noise(config, batch, 1, seed=7)[0]
    return out


def gen_sampler_surface(name="tiny", batch=2, only_missing=False):
    """Round 6: the parts of the kept `ddpm_sampler` / `ddim_sampler` surface that had no reference fixture.
    The REAL samplers (models/__init__.py:102-340) on the synthetic tiny net with the injected noise sequence:
      * the un-subsampled schedule (`subsample_steps` 1000 and None: schedule buffers used as they are, betas from the table and not
        1 - a / a_prev, :228-237) -- the branch BASELINE config 4 (`subsample=1000`) runs;
      * just_beta (:325-326), same_noise with and without an explicit noise_val (:259-260, :316-317), frac_steps (:250-254),
        denoise=False (:331), clip_before=False (:288), final_only=False (:292-293, :334-335, :340).
    `final_only=False` on a CPU run: `x_mod.to('cpu')` IS x_mod, so the later in-place `x_mod += c * noise` (:326-328) shows through in
    the list -- image i < L - 1 of a DDPM fixture is the POST-noise state.  On an accelerator the list holds pre-noise copies.  The
    fixture records what the CPU run returned; the tests add the step noise to the accelerator-semantics images before comparing."""
    import models as ref_models
    from oracle import sampler_ref
    config = synth.make_config(name)
    net = build_ref_net(config)
    sd = synth.make_state_dict(config, seed=123)
    net.load_state_dict(sd, strict=False)
    x, cond = synth.make_inputs(config, batch, seed=0)
    out = dict(config_name=name, batch=batch, cases={})
    path = os.path.join(OUT, f"{name}_b{batch}_surface.pt")
    if only_missing and os.path.exists(path):       # `surface_add`: keep the cases already in the file, run the new ones
        out = torch.load(path, weights_only=False)
    for key, kind, kw, n_noise in SURFACE_CASES:
        if key in out["cases"]:
            continue
        noise = synth.make_noise(config, batch, n_noise, seed=2)
        kwargs = surface_kwargs(kw, config, batch)
        inj = NoiseInjector(noise)
        orig = torch.randn_like
        torch.randn_like = inj
        try:
            res = dict(ddpm=ref_models.ddpm_sampler, ddim=ref_models.ddim_sampler)[kind](
                x.clone(), net, cond=cond, verbose=False, log=False, **kwargs)
        finally:
            torch.randn_like = orig
        # the same call in float64 on the restatement: the noise floor a tolerance on this case stands on
        k = [0]

        def fn(i, like, noise=noise, k=k):
            k[0] += 1
            return noise[k[0] - 1].to(like.dtype)
        kw64 = dict(kwargs)
        if kw64.get("noise_val", None) is not None:
            kw64["noise_val"] = kw64["noise_val"].double()
        net64 = unet_ref.OracleScoreNet(config, sd, dtype=torch.float64)
        res64 = sampler_ref.sample(x.double().clone(), net64, cond=cond.double(), kind=kind, noise_fn=fn, **kw64)
        assert k[0] == inj.k, (key, k[0], inj.k)
        cmp32 = res if kwargs["final_only"] else res[-1:]
        drift = float((cmp32.double() - (res64 if kwargs["final_only"] else res64[-1:])).abs().max())
        out["cases"][key] = dict(kind=kind, kwargs=dict(kw), n_noise=inj.k, n_predrawn=n_noise, result=res.clone(),
                                 ref32_vs_ref64_max_abs=drift)
        print(f"  {name}: {key}: draws {inj.k}, shape {tuple(res.shape)}, range [{res.min():.4f}, {res.max():.4f}], fp32 vs fp64 {drift:.3e}",
              flush=True)
    torch.save(out, path)
    print(f"wrote {name}_b{batch}_surface.pt")


def gen_full_schedule_wide(name="bair_big_spade", batch=1):
    """Round 6: BASELINE config 4 as the bench runs it -- the REAL `ddpm_sampler` over the FULL 1000-step schedule (`subsample=1000`,
    configs/bair_big_spade.yml; 1001 forwards of the SPADE net at full width), injected noise, final frames; plus the fp32-vs-fp64
    distance of the same call on the restatement."""
    import time
    import models as ref_models
    from oracle import sampler_ref
    config = synth.make_config(name)
    net = build_ref_net(config)
    sd = synth.make_state_dict(config, seed=123)
    net.load_state_dict(sd, strict=False)
    x, cond = synth.make_inputs(config, batch, seed=0)
    noise = synth.make_noise(config, batch, 1000, seed=2)
    inj = NoiseInjector(noise)
    orig = torch.randn_like
    torch.randn_like = inj
    t0 = time.time()
    try:
        res = ref_models.ddpm_sampler(x.clone(), net, cond=cond, final_only=True, denoise=True, subsample_steps=1000, clip_before=True,
                                      verbose=False, log=False)
    finally:
        torch.randn_like = orig
    print(f"  reference: {time.time() - t0:.0f} s, draws {inj.k}, range [{res.min():.4f}, {res.max():.4f}]", flush=True)
    k = [0]

    def fn(i, like):
        k[0] += 1
        return noise[k[0] - 1].to(like.dtype)
    t0 = time.time()
    net64 = unet_ref.OracleScoreNet(config, sd, dtype=torch.float64)
    res64 = sampler_ref.sample(x.double().clone(), net64, cond=cond.double(), kind="ddpm", final_only=True, denoise=True,
                               subsample_steps=1000, clip_before=True, noise_fn=fn)
    drift = float((res.double() - res64).abs().max())
    print(f"  fp64 restatement: {time.time() - t0:.0f} s; reference fp32 vs fp64 {drift:.3e}", flush=True)
    torch.save(dict(config_name=name, batch=batch, subsample=1000, kind="ddpm", result=res.clone(), n_noise=inj.k,
                    ref32_vs_ref64_max_abs=drift), os.path.join(OUT, f"{name}_b{batch}_ddpm1000.pt"))
    print(f"wrote {name}_b{batch}_ddpm1000.pt")


def main_round2():
    """Fixtures added in round 2 (VERDICT r01 'next round' item 1): the headline config end-to-end, configs 3 / 4 full samplers,
    the autoregressive driver at config 5 width, the BASELINE.json ch_mult variant, the cosine schedule."""
    torch.manual_seed(0)
    torch.set_num_threads(os.cpu_count())
    sys.path.insert(0, REF)
    which = sys.argv[2:] or ["cosine", "cfg2", "variant", "cfg3", "cfg4", "cfg5"]
    if "cosine" in which:
        gen_model_case("tiny_cosine", 2, [("ddpm", 10, {}), ("ddim", 10, {})])
    if "cfg2" in which:
        gen_model_case("smmnist_big5_ngf96", 2, [("ddpm", 100, {})])                  # BASELINE config 2 (the bench workload)
    if "variant" in which:
        gen_forward_only("cityscapes_big_variant", 1)
    if "cfg3" in which:
        gen_sampler_only("kth64_big_ngf128", 2, 100)
    if "cfg4" in which:
        gen_sampler_only("bair_big_spade", 2, 100)
    if "cfg5" in which:
        gen_autoregressive("cityscapes_big", 1, 8, 100)
    if "cfg2ddim" in which:        # round 4: DDIM (deterministic: no per-step noise to wash rounding out) over 100 steps at the headline width
        gen_sampler_only("smmnist_big5_ngf96", 2, 100, kind="ddim", measure_drift=True)
    if "cfg2fpndm" in which:       # round 4: the F-PNDM sampler at the headline width (25 sampler steps = 34 forwards)
        gen_fpndm_wide("smmnist_big5_ngf96", 2, 25)
    if "f4" in which:
        gen_f4()
    if "init" in which:
        gen_init_moments()
    if "tiny" in which:           # round 5: regenerate the tiny fixtures with the DDIM drift recorded inside them
        gen_model_case("tiny", 3, [("ddpm", 10, {}), ("ddim", 10, {}), ("ddpm", 10, dict(t_min=0.35))])
    if "surface" in which:        # round 6: the un-subsampled schedule and the untested kwargs of the kept sampler surface
        gen_sampler_surface("tiny", 2)
    if "gamma_ddim" in which:
        gen_gamma_ddim()
    if "surface_add" in which:
        gen_sampler_surface("tiny", 2, only_missing=True)
    if "cfg4full" in which:       # round 6: BASELINE config 4 over its full 1000-step schedule
        gen_full_schedule_wide("bair_big_spade", 1)
    if "cs_spade" in which:
        gen_forward_only("cityscapes_big_spade", 1)          # shipped config with 192-channel heads (VERDICT r01 missing item 5)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "round2":
        main_round2()
    else:
        main()

"""CPU: the oracle restatement vs fixtures produced by the real reference
(oracle/gen_golden.py).  Tolerances follow SURVEY 8c's measured noise floor."""
import os

import pytest
import torch

from oracle import sampler_ref, synth, unet_ref


def load(golden_dir, name):
    return torch.load(os.path.join(golden_dir, name), weights_only=False)


@pytest.mark.parametrize("fx", ["tiny_b3.pt", "tiny_spade_b2.pt", "smmnist_big5_b2.pt", "tiny_cosine_b2.pt"])
def test_schedule_bit_exact(golden_dir, fx):
    g = load(golden_dir, fx)
    c = unet_ref.hot_cfg(synth.make_config(g["config_name"]))
    betas, alphas, alphas_prev = unet_ref.make_schedule(c)
    assert torch.equal(betas, g["betas"])
    assert torch.equal(alphas, g["alphas"])
    assert torch.equal(alphas_prev, g["alphas_prev"])


@pytest.mark.parametrize("fx", ["tiny_b3.pt", "tiny_spade_b2.pt", "smmnist_big5_b2.pt", "tiny_cosine_b2.pt",
                                "smmnist_big5_ngf96_b2.pt"])
def test_forward_matches_reference(golden_dir, fx):
    g = load(golden_dir, fx)
    config = synth.make_config(g["config_name"])
    sd = synth.make_state_dict(config, seed=123)
    x, cond = synth.make_inputs(config, g["batch"], seed=0)
    taps = {}
    with torch.no_grad():
        eps = unet_ref.unet_forward(sd, config, x, g["fwd_t"], cond, taps=taps)
    ref = g["fwd_eps"]
    assert eps.shape == ref.shape
    assert (eps - ref).abs().max().item() <= 1e-4 * ref.abs().max().item()
    # every module output, via the strided probes
    assert sorted(taps.keys()) == sorted(g["fwd_taps"].keys())
    for i, p in g["fwd_taps"].items():
        mine = taps[i]
        assert list(mine.shape) == p["shape"], i
        got = mine.reshape(-1)[p["idx"]]
        torch.testing.assert_close(got, p["sample"], rtol=1e-4, atol=2e-5, msg=f"module {i}")
        assert abs(mine.double().mean().item() - p["mean"]) <= 1e-5 + 1e-4 * abs(p["absmean"])


def _injector(noise):
    k = [0]

    def fn(i, like):
        z = noise[k[0]]
        k[0] += 1
        return z
    return fn, k


@pytest.mark.parametrize("fx,key,kind,sub,extra", [
    ("tiny_b3.pt", "ddpm_10", "ddpm", 10, {}),
    ("tiny_b3.pt", "ddim_10", "ddim", 10, {}),
    ("tiny_b3.pt", "ddpm_10_t_min0.35", "ddpm", 10, dict(t_min=0.35)),
    ("tiny_spade_b2.pt", "ddpm_10", "ddpm", 10, {}),
    ("smmnist_big5_b2.pt", "ddpm_100", "ddpm", 100, {}),      # BASELINE config 1, full 101 forwards
    ("smmnist_big5_ngf96_b2.pt", "ddpm_100", "ddpm", 100, {}),   # BASELINE config 2 (the bench workload), full 101 forwards
    ("tiny_cosine_b2.pt", "ddpm_10", "ddpm", 10, {}),         # sigma_dist: cosine (models/__init__.py:28-32)
    ("tiny_cosine_b2.pt", "ddim_10", "ddim", 10, {}),
])
def test_sampler_matches_reference(golden_dir, fx, key, kind, sub, extra):
    g = load(golden_dir, fx)
    config = synth.make_config(g["config_name"])
    net = unet_ref.OracleScoreNet(config, synth.make_state_dict(config, seed=123))
    x, cond = synth.make_inputs(config, g["batch"], seed=0)
    noise = synth.make_noise(config, g["batch"], sub + 1, seed=2)
    fn, k = _injector(noise)
    out = sampler_ref.sample(x.clone(), net, cond=cond, kind=kind, final_only=True, denoise=True,
                             subsample_steps=sub, clip_before=True, noise_fn=fn, **extra)
    ref = g["sampler_" + key]["result"]
    assert k[0] == g["sampler_" + key]["n_noise"]
    assert out.shape == ref.shape
    assert (out - ref).abs().max().item() <= 1e-4


@pytest.mark.parametrize("fx", ["kth64_big_ngf128_b2_fwd.pt", "bair_big_spade_b2_fwd.pt", "cityscapes_big_b1_fwd.pt",
                                "cityscapes_big_variant_b1_fwd.pt", "cityscapes_big_spade_b1_fwd.pt"])
def test_forward_matches_reference_full_width(golden_dir, fx):
    """BASELINE configs 3-5 (ngf=128 / SPADE at full width / 128x128 five-level): the oracle vs the REAL reference's forward,
    module by module (strided probes) and on the final eps."""
    torch.set_num_threads(min(os.cpu_count() or 1, 16))
    g = load(golden_dir, fx)
    config = synth.make_config(g["config_name"])
    sd = synth.make_state_dict(config, seed=123)
    x, cond = synth.make_inputs(config, g["batch"], seed=0)
    taps = {}
    with torch.no_grad():
        eps = unet_ref.unet_forward(sd, config, x, g["fwd_t"], cond, taps=taps)
    for i, p in g["fwd_taps"].items():
        if i == 0 or i not in taps:
            continue
        got = taps[i].reshape(-1).double()[p["idx"]].float()
        torch.testing.assert_close(got, p["sample"], rtol=1e-4, atol=3e-5, msg=f"module {i}")
    p = g["fwd_eps_probe"]
    assert list(eps.shape) == p["shape"]
    torch.testing.assert_close(eps.reshape(-1).double()[p["idx"]].float(), p["sample"], rtol=1e-4, atol=3e-5)


@pytest.mark.parametrize("fx", ["kth64_big_ngf128_b2_ddpm100.pt", "bair_big_spade_b2_ddpm100.pt"])
def test_full_width_sampler_matches_reference(golden_dir, fx):
    """BASELINE configs 3 / 4 (ngf=128; SPADE) end to end: 100-step ddpm_sampler + denoise of the REAL reference with the injected
    noise sequence vs the oracle (final frames, 1e-4)."""
    torch.set_num_threads(min(os.cpu_count() or 1, 16))
    g = load(golden_dir, fx)
    config = synth.make_config(g["config_name"])
    net = unet_ref.OracleScoreNet(config, synth.make_state_dict(config, seed=123))
    x, cond = synth.make_inputs(config, g["batch"], seed=0)
    fn, k = _injector(synth.make_noise(config, g["batch"], g["subsample"] + 1, seed=2))
    out = sampler_ref.sample(x.clone(), net, cond=cond, kind=g["kind"], final_only=True, denoise=True,
                             subsample_steps=g["subsample"], clip_before=True, noise_fn=fn)
    assert k[0] == g["n_noise"] and out.shape == g["result"].shape
    assert (out - g["result"]).abs().max().item() <= 1e-4


def test_ddim_100_at_the_headline_width_matches_reference(golden_dir):
    """The oracle's ddim sampler against the REAL reference's 100-step DDIM at BASELINE config 2 (ngf 96): deterministic sampler, the
    tolerance is three times the reference's own fp32-vs-fp64 distance on this call (recorded in the fixture), at least 1e-4."""
    torch.set_num_threads(min(os.cpu_count() or 1, 16))
    g = load(golden_dir, "smmnist_big5_ngf96_b2_ddim100.pt")
    config = synth.make_config(g["config_name"])
    net = unet_ref.OracleScoreNet(config, synth.make_state_dict(config, seed=123))
    x, cond = synth.make_inputs(config, g["batch"], seed=0)
    out = sampler_ref.sample(x.clone(), net, cond=cond, kind="ddim", final_only=True, denoise=True, subsample_steps=100, clip_before=True)
    assert out.shape == g["result"].shape
    assert (out - g["result"]).abs().max().item() <= max(1e-4, 3.0 * g["ref32_vs_ref64_max_abs"])


def test_fpndm_25_at_the_headline_width_matches_reference(golden_dir):
    """The oracle's F-PNDM sampler against the REAL reference's FPNDM_sampler (models/__init__.py:38-99, models/pndm.py) at BASELINE
    config 2 (ngf 96), 25 sampler steps = 34 forwards: deterministic linear multistep -- the tolerance is three times the reference's
    fp32-vs-fp64 distance on this call (recorded in the fixture), at least 1e-4."""
    torch.set_num_threads(min(os.cpu_count() or 1, 16))
    g = load(golden_dir, "smmnist_big5_ngf96_b2_fpndm25.pt")
    config = synth.make_config(g["config_name"])
    net = unet_ref.OracleScoreNet(config, synth.make_state_dict(config, seed=123))
    x, cond = synth.make_inputs(config, g["batch"], seed=0)
    out = sampler_ref.fpndm_sample(x.clone(), net, cond=cond, final_only=True, subsample_steps=g["subsample"], clip_before=True)
    assert out.shape == g["result"].shape
    assert (out - g["result"]).abs().max().item() <= max(1e-4, 3.0 * g["ref32_vs_ref64_max_abs"])


def ar_oracle(config, sd, batch, nfp, subsample):
    """The autoregressive block loop (runners/ncsn_runner.py:1504-1569) over the oracle sampler, with the inputs of
    oracle/gen_golden.py:gen_autoregressive (init seed 50 + block, step noise seed 60 + block)."""
    from math import ceil
    net = unet_ref.OracleScoreNet(config, sd)
    C, nf, nc, S = config.data.channels, config.data.num_frames, config.data.num_frames_cond, config.data.image_size
    _, cond = synth.make_inputs(config, batch, seed=0)
    preds = []
    n_iter = ceil(nfp / nf)
    for i in range(n_iter):
        init = torch.randn(batch, C * nf, S, S, generator=torch.Generator().manual_seed(50 + i))
        fn, _ = _injector(synth.make_noise(config, batch, subsample + 1, seed=60 + i))
        gen = sampler_ref.sample(init, net, cond=cond, kind="ddpm", final_only=True, denoise=True, subsample_steps=subsample,
                                 clip_before=True, noise_fn=fn)[0]
        preds.append(gen)
        if i != n_iter - 1:
            cond = torch.cat([cond[:, C * nf:], gen[:, C * max(0, nf - nc):]], dim=1)
    return torch.cat(preds, dim=1)[:, :C * nfp]


def test_autoregressive_config5_matches_reference(golden_dir):
    """BASELINE config 5 (cityscapes, 128x128, nc=2 < nf=5): two autoregressive blocks cropped to 8 frames, reference vs oracle."""
    torch.set_num_threads(min(os.cpu_count() or 1, 16))
    g = load(golden_dir, "cityscapes_big_b1_ar8.pt")
    config = synth.make_config(g["config_name"])
    out = ar_oracle(config, synth.make_state_dict(config, seed=123), g["batch"], g["nfp"], g["subsample"])
    assert out.shape == g["pred"].shape
    assert (out - g["pred"]).abs().max().item() <= 2e-4      # two chained 100-step blocks


# ---------------------------------------------------------------- SURVEY 8f rank 4 flags
def _seq(fn_seq):
    k = [0]

    def fn(*_):
        k[0] += 1
        return fn_seq[k[0] - 1]
    return fn, k


def test_cond_emb_matches_reference(golden_dir):
    """model.cond_emb: Embedding(2, ngf/2)(cond_mask) concatenated to temb (ncsnpp_more.py:97-99, :282-286), with a mask, with the
    default all-ones mask, and through the sampler (which never forwards cond_mask, models/__init__.py:263)."""
    g = load(golden_dir, "tiny_condemb_b3.pt")
    config = synth.make_config(g["config_name"])
    sd = synth.make_state_dict(config, seed=123)
    x, cond = synth.make_inputs(config, g["batch"], seed=0)
    taps = {}
    with torch.no_grad():
        em = unet_ref.unet_forward(sd, config, x, g["fwd_t"], cond, taps=taps, cond_mask=g["mask"])
        en = unet_ref.unet_forward(sd, config, x, g["fwd_t"], cond)
    assert (em - g["eps_mask"]).abs().max().item() <= 1e-4 * g["eps_mask"].abs().max().item()
    assert (en - g["eps_none"]).abs().max().item() <= 1e-4 * g["eps_none"].abs().max().item()
    assert sorted(taps) == sorted(g["fwd_taps"])
    for i, p in g["fwd_taps"].items():
        torch.testing.assert_close(taps[i].reshape(-1)[p["idx"]], p["sample"], rtol=1e-4, atol=2e-5, msg=f"module {i}")
    fn, k = _injector(synth.make_noise(config, g["batch"], 11, seed=2))
    out = sampler_ref.sample(x.clone(), unet_ref.OracleScoreNet(config, sd), cond=cond, kind="ddpm", final_only=True, denoise=True,
                             subsample_steps=10, noise_fn=fn)
    assert k[0] == g["n_noise"] and (out - g["sampler"]).abs().max().item() <= 1e-4


@pytest.mark.parametrize("fx", ["tiny_noisecond_b2.pt", "tiny_spade_noisecond_b2.pt"])
def test_noise_in_cond_matches_reference(golden_dir, fx):
    """model.noise_in_cond (ncsnpp_more.py:755-768): every forward diffuses cond to its labels' level with a fresh draw."""
    g = load(golden_dir, fx)
    config = synth.make_config(g["config_name"])
    net = unet_ref.OracleScoreNet(config, synth.make_state_dict(config, seed=123))
    x, cond = synth.make_inputs(config, g["batch"], seed=0)
    net.cond_noise_fn, kc = _seq(g["cond_seq"])
    eps = net(x, g["fwd_t"], cond=cond)
    assert (eps - g["fwd_eps"]).abs().max().item() <= 1e-4 * g["fwd_eps"].abs().max().item()
    fn, k = _injector(synth.make_noise(config, g["batch"], 11, seed=2))
    out = sampler_ref.sample(x.clone(), net, cond=cond, kind="ddpm", final_only=True, denoise=True, subsample_steps=10, noise_fn=fn)
    assert kc[0] == 12 and k[0] == 9
    assert (out - g["sampler"]).abs().max().item() <= 1e-4


def test_gamma_sampler_matches_reference(golden_dir):
    """gamma=True (models/__init__.py:224-225, :273-276, :319-322) on a model.gamma + noise_in_cond net: the raw Gamma draws of the
    reference run are replayed; also with t_min > 0 (the re-noise draw)."""
    g = load(golden_dir, "tiny_gamma_b2.pt")
    config = synth.make_config(g["config_name"])
    net = unet_ref.OracleScoreNet(config, synth.make_state_dict(config, seed=123))
    torch.testing.assert_close(net.k_cum, g["k_cum"], rtol=1e-6, atol=0)
    torch.testing.assert_close(net.theta_t, g["theta_t"], rtol=1e-6, atol=0)
    x, cond = synth.make_inputs(config, g["batch"], seed=0)
    for key, extra in (("", {}), ("_tmin", dict(t_min=0.35))):
        net.cond_noise_fn, kc = _seq(g["cond_z" + key])
        fn, k = _seq(g["step_raw" + key])
        out = sampler_ref.sample(x.clone(), net, cond=cond, kind="ddpm", final_only=True, denoise=True, subsample_steps=10,
                                 noise_fn=fn, gamma=True, **extra)
        assert k[0] == len(g["step_raw" + key]) and kc[0] == len(g["cond_z" + key])
        assert (out - g["sampler" + key]).abs().max().item() <= 1e-4


def test_output_all_frames_fails_like_the_reference(golden_dir):
    g = load(golden_dir, "tiny_allframes_err.pt")
    config = synth.make_config("tiny_allframes")
    x, cond = synth.make_inputs(config, 2, seed=0)
    with pytest.raises(RuntimeError) as e:
        unet_ref.unet_forward(synth.make_state_dict(config, seed=123), config, x, torch.tensor([5, 6]), cond)
    assert "split_with_sizes" in str(e.value) and "split_with_sizes" in g["error"]


def test_fpndm_matches_reference(golden_dir):
    """FPNDM_sampler (models/__init__.py:38-99 + models/pndm.py): every step of the clipped run, and the un-clipped final state
    (which grows to |x| ~ 360 on random weights: relative tolerance)."""
    g = load(golden_dir, "tiny_b3_fpndm.pt")
    config = synth.make_config(g["config_name"])
    net = unet_ref.OracleScoreNet(config, synth.make_state_dict(config, seed=123))
    x, cond = synth.make_inputs(config, g["batch"], seed=0)
    out = sampler_ref.fpndm_sample(x.clone(), net, cond=cond, final_only=False, subsample_steps=g["subsample"], clip_before=True)
    assert out.shape == g["all_clip"].shape
    assert (out - g["all_clip"]).abs().max().item() <= 1e-4
    fin = sampler_ref.fpndm_sample(x.clone(), net, cond=cond, final_only=True, subsample_steps=g["subsample"], clip_before=False)
    torch.testing.assert_close(fin, g["final_noclip"], rtol=2e-3, atol=2e-3)


def test_fir_closed_forms(golden_dir):
    g = load(golden_dir, "fir.pt")
    torch.testing.assert_close(unet_ref.fir_up2(g["x"]), g["up"], rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(unet_ref.fir_down2(g["x"]), g["down"], rtol=1e-5, atol=1e-6)
    a = g["generic_args"]
    got = unet_ref.upfirdn2d_generic(g["x"], g["kernel"] * a["gain"], a["up"], a["down"], a["pad0"], a["pad1"])
    torch.testing.assert_close(got, g["generic"], rtol=1e-5, atol=1e-6)
    # the two hot-path uses expressed through the generic op (up_or_down_sampling.py:222-225, 255-258)
    k = g["kernel"]
    torch.testing.assert_close(unet_ref.upfirdn2d_generic(g["x"], k * 4, 2, 1, 2, 1), g["up"], rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(unet_ref.upfirdn2d_generic(g["x"], k, 1, 2, 1, 1), g["down"], rtol=1e-5, atol=1e-6)


def test_conv_unfold_restatement():
    g = torch.Generator().manual_seed(3)
    x = torch.randn(2, 5, 9, 9, generator=g)
    for ks in (1, 3):
        w = torch.randn(7, 5, ks, ks, generator=g)
        b = torch.randn(7, generator=g)
        torch.testing.assert_close(unet_ref.conv2d_unfold(x, w, b), unet_ref.conv2d(x, w, b), rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("name,n_mod,n_param", [
    ("smmnist_big5_ngf96", 43, 376), ("cityscapes_big", 50, None), ("bair_big_spade", 43, 716)])
def test_topology_counts(name, n_mod, n_param):
    """SURVEY 9.2 / 9.5 / 9.9 module and key counts (state_dict keys = params + 4 buffers)."""
    c = unet_ref.hot_cfg(synth.make_config(name))
    assert len(unet_ref.module_plan(c)) == n_mod
    if n_param is not None:
        assert len(unet_ref.param_shapes(c)) == n_param


# ------------------------------------------------------------------ round 6: the rest of the kept sampler surface (tiny_b2_surface.pt)
SURFACE_KEYS = [
    "ddpm_full1000", "ddpm_fullNone", "ddim_full1000", "ddim_fullNone_tmin0.9", "ddpm_full1000_tmin0.9_just_beta",
    "ddpm_fullNone_tmin0.95", "ddpm_10_just_beta", "ddpm_tail_just_beta", "ddpm_10_same_noise", "ddpm_10_same_noise_val",
    "ddpm_10_noise_val_ignored", "ddpm_10_same_noise_tmin", "ddpm_frac0.2", "ddpm_frac0.1_tmin0.5", "ddpm_10_nodenoise",
    "ddim_10_nodenoise", "ddpm_10_noclip", "ddim_10_noclip", "ddpm_10_images", "ddim_10_images", "ddpm_10_images_nodenoise",
    "ddpm_100_tmin0.1", "ddim_100_tmin0.3", "ddpm_full1000_tmin0.998", "ddpm_7", "ddim_7", "ddpm_7_images_tmin",
]


def surface_case(golden_dir, key):
    """(config, batch, x, cond, noise, case dict, call kwargs, tolerance) of one case of tiny_b2_surface.pt.  Tolerance: 1e-4 on frames
    in [-1, 1], 3e-4 for the deterministic DDIM (no step noise washes rounding out), relative 1e-5 where clip_before=False
    lets |x| grow to ~1e3; each stands at least 3x above the reference's own fp32-vs-fp64 distance recorded in the fixture."""
    g = load(golden_dir, "tiny_b2_surface.pt")
    c = g["cases"][key]
    config = synth.make_config(g["config_name"])
    B = g["batch"]
    x, cond = synth.make_inputs(config, B, seed=0)
    noise = synth.make_noise(config, B, c["n_predrawn"], seed=2)
    kw = dict(final_only=True, denoise=True, clip_before=True)
    kw.update(c["kwargs"])
    if kw.get("noise_val", None) == "NOISE_VAL":
        kw["noise_val"] = synth.make_noise(config, B, 1, seed=7)[0]
    ref = c["result"]
    tol = 3e-4 if c["kind"] == "ddim" else 1e-4
    if not kw["clip_before"]:
        tol = 1e-5 * ref.abs().max().item()
    assert 0.0 < c["ref32_vs_ref64_max_abs"] and 3.0 * c["ref32_vs_ref64_max_abs"] <= tol, (key, c["ref32_vs_ref64_max_abs"], tol)
    return config, B, x, cond, noise, c, kw, tol


def add_back_step_noise(images, kind, kw, noise, alphas_full, alphas_prev_full, betas_full):
    """`final_only=False` images with accelerator semantics (pre-noise copies) -> what a CPU run of the reference returns, where
    `x_mod.to('cpu')` aliases x_mod and the in-place `x_mod += c * noise` (models/__init__.py:326-328) shows through: image i of a DDPM
    run gains the step's noise term for every executed step but the last (DDIM adds no noise; the denoise image is a new tensor).  Steps the
    t_min test skips (:269-270) leave no image; the re-noise of the first executed step consumes one draw (:272-279)."""
    if kind != "ddpm":
        return images
    steps, al, alp, be = sampler_ref.subsampled_schedule(alphas_full, alphas_prev_full, betas_full, kw["subsample_steps"])
    t_min = kw.get("t_min", -1)
    out = images.clone()
    L = len(steps)
    img, draw, started = 0, 0, False
    for i, step in enumerate(steps):
        if step < t_min * len(al):
            continue
        if not started and t_min > 0:
            draw += 1
        started = True
        if i + 1 != L:
            cn = be[i].sqrt() if kw.get("just_beta", False) else ((1 - alp[i]) / (1 - al[i]) * be[i]).sqrt()
            out[img] = out[img] + cn * noise[draw]
            draw += 1
        img += 1
    return out


@pytest.mark.parametrize("key", SURFACE_KEYS)
def test_sampler_surface_matches_reference(golden_dir, key):
    """The un-subsampled schedule (subsample_steps 1000 / None: tables as they are, models/__init__.py:228-237 not taken) and the kwargs
    just_beta / same_noise / noise_val / frac_steps / denoise=False / clip_before=False / final_only=False of the REAL samplers."""
    config, B, x, cond, noise, c, kw, tol = surface_case(golden_dir, key)
    net = unet_ref.OracleScoreNet(config, synth.make_state_dict(config, seed=123))
    fn, k = _injector(noise)
    out = sampler_ref.sample(x.clone(), net, cond=cond, kind=c["kind"], noise_fn=fn, **kw)
    assert k[0] == c["n_noise"], (key, k[0], c["n_noise"])
    if not kw["final_only"]:
        out = add_back_step_noise(out, c["kind"], kw, noise, net.alphas, net.alphas_prev, net.betas)
    ref = c["result"]
    assert out.shape == ref.shape
    err = (out - ref).abs().max().item()
    assert err <= tol, f"{key}: {err:.3e} > {tol:.1e}"


def test_full_schedule_uses_the_table_betas(golden_dir):
    """What separates the un-subsampled branch from a subsampling with skip 1: betas are the TABLE's (linspace), not 1 - a / a_prev
    recomputed in fp32 (models/__init__.py:236 'for some reason we lose a bit of precision here') -- the two differ in the last bits,
    and the oracle must take the table."""
    config = synth.make_config("tiny")
    net = unet_ref.OracleScoreNet(config, synth.make_state_dict(config, seed=123))
    for sub in (1000, None, 2000):
        steps, al, alp, be = sampler_ref.subsampled_schedule(net.alphas, net.alphas_prev, net.betas, sub)
        assert len(steps) == 1000 and torch.equal(be, net.betas) and torch.equal(al, net.alphas) and torch.equal(alp, net.alphas_prev)
    recomputed = 1.0 - net.alphas / net.alphas_prev
    assert not torch.equal(recomputed, net.betas)
    steps, al, alp, be = sampler_ref.subsampled_schedule(net.alphas, net.alphas_prev, net.betas, 999)   # skip 1, but the branch IS taken
    assert len(steps) == 1000 and torch.equal(be, recomputed)


def test_ddim_gamma_renoise_matches_reference(golden_dir):
    """`ddim_sampler(..., gamma=True, t_min=0.35)` on a model.gamma + noise_in_cond net (models/__init__.py:118, :144-151): the one place the
    DDIM sampler reads `gamma` is the re-noise draw of the first executed step; the raw Gamma draws of the reference run are replayed."""
    g = load(golden_dir, "tiny_gamma_ddim_b2.pt")
    config = synth.make_config(g["config_name"])
    net = unet_ref.OracleScoreNet(config, synth.make_state_dict(config, seed=123))
    x, cond = synth.make_inputs(config, g["batch"], seed=0)
    net.cond_noise_fn, kc = _seq(g["cond_z_tmin"])
    fn, k = _seq(g["step_raw_tmin"])
    out = sampler_ref.sample(x.clone(), net, cond=cond, kind="ddim", final_only=True, denoise=True, subsample_steps=10, noise_fn=fn, gamma=True,
                             t_min=0.35)
    assert k[0] == 1 and kc[0] == len(g["cond_z_tmin"]) == 10
    assert (out - g["sampler_tmin"]).abs().max().item() <= 3e-4
    # without the kwarg the re-noise draw would be a normal one: the gamma draw is what the fixture holds
    fn2, _ = _seq(g["step_raw_tmin"])
    net.cond_noise_fn, _ = _seq(g["cond_z_tmin"])
    other = sampler_ref.sample(x.clone(), net, cond=cond, kind="ddim", final_only=True, denoise=True, subsample_steps=10, noise_fn=fn2, gamma=False,
                               t_min=0.35)
    assert (other - g["sampler_tmin"]).abs().max().item() > 1e-2

"""GPU parity of the REST of the kept sampler surface, through the C ABI, against fixtures the REAL reference samplers
produced (oracle/gen_golden.py: gen_sampler_surface, gen_full_schedule_wide):

  * the un-subsampled schedule -- `subsample_steps` 1000 / None, models/__init__.py:228-237 NOT taken: the schedule buffers are used as
    they are and the betas are the table's, not 1 - a / a_prev -- which is the branch BASELINE config 4 (`sampling.subsample: 1000`) runs
    in the bench: `mcvd_sampler_run`'s `else` leg (csrc/api.cpp) and `samplers._subsample`'s fall-through;
  * `just_beta` (:325-326; MCVD_FLAG_JUST_BETA on the device loop), `same_noise` with and without `noise_val` (:259-260, :316-317),
    `frac_steps` (:250-254), `denoise=False` (:331), `clip_before=False` (:288), `final_only=False` (:292-293, :334-335, :340);
  * BASELINE config 4 at full width over its full 1000-step schedule (B = 1 fixture; and row 0 of a B = 16 run, the bench's batch).

Every case runs the way the call selects (device loop where the library serves it) and, where that was the device loop, once more on
the host loop (`final_only=False`, last image)."""
import ctypes as C
import json
import os

import pytest
import torch

from oracle import synth
from tests.test_oracle_golden import SURFACE_KEYS, add_back_step_noise, surface_case

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _net(name):
    from mcvd_pytorch_amd.scorenet import HipScoreNet
    config = synth.make_config(name)
    config.device = "cuda:0"
    sd = synth.make_state_dict(config, seed=123)
    net = HipScoreNet(config)
    net.load_state_dict(sd, strict=True)
    return config, sd, net.eval()


class _RunSpy:
    """Counts the `mcvd_sampler_run` calls made through the ctypes handle and keeps their `flags` / `subsample_steps` arguments."""

    def __init__(self):
        from mcvd_pytorch_amd import _lib
        self.lib, self.real, self.calls = _lib.lib, _lib.lib.mcvd_sampler_run, []

    def __enter__(self):
        def spy(*a):
            self.calls.append(dict(subsample_steps=a[7], flags=a[8], t_min=a[9]))
            return self.real(*a)
        self.lib.mcvd_sampler_run = spy
        return self

    def __exit__(self, *exc):
        self.lib.mcvd_sampler_run = self.real


def _device_loop_serves(kw):
    """samplers._sample's `fast` test: the whole loop runs inside mcvd_sampler_run."""
    return kw["final_only"] and not kw.get("same_noise", False) and kw.get("noise_val", None) is None and kw.get("frac_steps", None) is None


@pytest.mark.parametrize("path", ["as_called", "host_loop"])
@pytest.mark.parametrize("key", SURFACE_KEYS)
def test_sampler_surface_vs_reference_golden(golden_dir, key, path):
    from mcvd_pytorch_amd import _lib
    from mcvd_pytorch_amd.samplers import ddim_sampler, ddpm_sampler
    config, B, x, cond, noise, c, kw, tol = surface_case(golden_dir, key)
    on_device = _device_loop_serves(kw)
    if path == "host_loop" and not on_device:
        pytest.skip("the call already runs on the host loop")
    _, _, net = _net("tiny")
    call = dict(kw)
    if call.get("noise_val", None) is not None:
        call["noise_val"] = call["noise_val"].cuda()
    if path == "host_loop":
        call["final_only"] = False
    sampler = ddpm_sampler if c["kind"] == "ddpm" else ddim_sampler
    with _RunSpy() as spy:
        out = sampler(x.cuda(), net, cond=cond.cuda(), verbose=False, log=False, noise=noise.cuda(), cond_mask=None, n_steps_each=0,
                      step_lr=0.0, config=config, **call)
    # the path that ran is the one this case is about
    if path == "as_called" and on_device:
        assert len(spy.calls) == 1, f"{key}: the device loop did not run"
        f = spy.calls[0]["flags"]
        assert bool(f & _lib.FLAG_JUST_BETA) == bool(kw.get("just_beta", False))
        assert bool(f & _lib.FLAG_DENOISE) == kw["denoise"] and bool(f & _lib.FLAG_CLIP_BEFORE) == kw["clip_before"]
        assert spy.calls[0]["subsample_steps"] == (kw["subsample_steps"] or 0)
        assert out.is_cuda and out.shape[0] == 1
    else:
        assert not spy.calls, f"{key} [{path}]: expected the host loop"
    ref = c["result"]
    if not call["final_only"]:
        assert out.device.type == "cpu"                       # the reference returns the stacked CPU images (:340)
        if kw["final_only"]:
            out = out[-1:]                                    # the fixture holds the final frames only
        else:
            from oracle import sampler_ref  # noqa: F401  (schedule helper lives beside the oracle sampler)
            out = add_back_step_noise(out, c["kind"], kw, noise, net.alphas.cpu(), net.alphas_prev.cpu(), net.betas.cpu())
    out = out.cpu()
    assert out.shape == ref.shape, (key, out.shape, ref.shape)
    err = (out - ref).abs().max().item()
    assert err <= tol, f"{key} [{path}]: {err:.3e} > {tol:.1e}"


def test_device_loop_full_schedule_takes_the_table_betas(golden_dir):
    """The un-subsampled leg of `mcvd_sampler_run` through the raw C ABI (no Python sampler around it): one DDPM step of the FULL schedule
    (t_min just under 1 leaves step 999 only... the last step adds no noise, so two steps: 998 and 999) must equal the same two steps
    computed from the TABLE betas -- and differ from the recomputed 1 - a / a_prev ones where those differ."""
    from mcvd_pytorch_amd import _lib
    config, sd, net = _net("tiny")
    B = 2
    x, cond = synth.make_inputs(config, B, seed=0)
    noise = synth.make_noise(config, B, 2, seed=2).cuda()
    net.sync_parameters(force=True)
    al, alp, be = net.alphas.cpu(), net.alphas_prev.cpu(), net.betas.cpu()
    for flags, jb in ((_lib.FLAG_CLIP_BEFORE, False), (_lib.FLAG_CLIP_BEFORE | _lib.FLAG_JUST_BETA, True)):
        xd = x.cuda().clone()
        with torch.cuda.device(net.device):
            net._bind_stream()
            _lib.check(_lib.lib.mcvd_sampler_run(net._model, _lib.SAMPLER_DDPM, C.c_void_p(xd.data_ptr()), C.c_void_p(cond.cuda().data_ptr()),
                                                 C.c_void_p(noise.data_ptr()), C.c_uint64(0), C.c_uint64(0), 1000, flags, float(0.998), B),
                       "sampler_run")
        torch.cuda.synchronize()
        # the same two steps with torch arithmetic around the HIP forward (models/__init__.py:272-328 on the table values)
        xx = x.cuda().clone()
        i0 = 998
        xx = al[i0].sqrt().item() * xx + (1 - al[i0]).sqrt().item() * noise[0]                           # t_min re-noise (:272-279)
        for k, i in enumerate((998, 999)):
            a, ap, b = al[i], alp[i], be[i]
            eps = net(xx, torch.full((B,), i, device="cuda", dtype=torch.long), cond=cond.cuda())
            x0 = ((1 / a.sqrt()) * (xx - (1 - a).sqrt() * eps)).clip_(-1, 1)
            xx = (ap.sqrt() * b / (1 - a)) * x0 + ((1 - b).sqrt() * (1 - ap) / (1 - a)) * xx
            if i != 999:
                xx = xx + (b.sqrt() if jb else ((1 - ap) / (1 - a) * b).sqrt()) * noise[1]
        err = (xd - xx).abs().max().item()
        assert err <= 1e-5, f"just_beta={jb}: {err:.3e}"


def test_frac_steps_on_a_subsampled_schedule_fails_like_the_reference():
    """models/__init__.py:251-254 index the (already subsampled, length-S) tables with the step VALUES: IndexError in the reference for any
    S < 1000; the same here, before any forward runs."""
    from mcvd_pytorch_amd.samplers import ddpm_sampler
    config, sd, net = _net("tiny")
    x, cond = synth.make_inputs(config, 2, seed=0)
    with pytest.raises(IndexError):
        ddpm_sampler(x.cuda(), net, cond=cond.cuda(), subsample_steps=10, frac_steps=0.5, final_only=True)


@pytest.mark.parametrize("path", ["device_loop", "host_loop"])
def test_config4_full_1000_step_schedule_vs_reference_golden(golden_dir, path):
    """BASELINE config 4 as the bench runs it: `bair_big_spade`, SPADE gamma/beta cached across 1001 forwards, `subsample=1000` -- the REAL
    reference's final frames (B = 1, injected noise)."""
    from mcvd_pytorch_amd.samplers import ddpm_sampler
    g = torch.load(os.path.join(golden_dir, "bair_big_spade_b1_ddpm1000.pt"), weights_only=False)
    assert g["subsample"] == 1000 and g["n_noise"] == 999
    config, sd, net = _net(g["config_name"])
    x, cond = synth.make_inputs(config, g["batch"], seed=0)
    noise = synth.make_noise(config, g["batch"], 1000, seed=2)
    tol = max(1e-4, 3.0 * g["ref32_vs_ref64_max_abs"])
    with _RunSpy() as spy:
        out = ddpm_sampler(x.cuda(), net, cond=cond.cuda(), denoise=True, subsample_steps=1000, clip_before=True, verbose=False, log=False,
                           noise=noise.cuda(), final_only=(path == "device_loop"))[-1:].cpu()
    assert len(spy.calls) == (1 if path == "device_loop" else 0)
    err = (out - g["result"]).abs().max().item()
    assert err <= tol, f"config 4, 1000 steps [{path}]: {err:.3e} > {tol:.2e} (reference fp32 vs fp64 {g['ref32_vs_ref64_max_abs']:.3e})"


def test_config4_full_schedule_at_the_benchmarked_batch(golden_dir):
    """The bench's config-4 call itself: B = 16 per GPU, 1000 steps + denoise on the device loop under the committed kernel table.  Row 0
    carries the fixture's sample (same x, cond and noise row) and must reproduce the REAL reference's frames; every row is finite and
    no two rows coincide."""
    from mcvd_pytorch_amd.samplers import ddpm_sampler
    g = torch.load(os.path.join(golden_dir, "bair_big_spade_b1_ddpm1000.pt"), weights_only=False)
    config, sd, net = _net(g["config_name"])
    B = 16
    path = os.path.join(ROOT, "profiles", f"tune_bair_big_spade_B{B}_bf16x3.json")
    if os.path.exists(path):
        net.set_tuning(B, json.load(open(path))[str(B)])
    x, cond = synth.make_inputs(config, B, seed=0)            # rows keyed by global index: row 0 is the fixture's row
    noise = torch.randn(1000, B, *x.shape[1:], generator=torch.Generator().manual_seed(99))
    noise[:, 0] = synth.make_noise(config, 1, 1000, seed=2)[:, 0]
    out = ddpm_sampler(x.cuda(), net, cond=cond.cuda(), denoise=True, subsample_steps=1000, clip_before=True, verbose=False, log=False,
                       noise=noise.cuda(), final_only=True)[-1].cpu()
    assert torch.isfinite(out).all()
    tol = max(1e-4, 3.0 * g["ref32_vs_ref64_max_abs"])
    err = (out[:1] - g["result"][0]).abs().max().item()
    assert err <= tol, f"row 0 of the B = {B} config-4 run vs the reference's frames: {err:.3e} > {tol:.2e}"
    flat = out.flatten(1)
    d = (flat[1:] - flat[:-1]).abs().max(dim=1).values
    assert (d > 1e-3).all()


@pytest.mark.parametrize("fx", ["tiny_runner_videogen.pt", "tiny_runner_videogen_prevt.pt", "tiny_runner_videogen_oneframe.pt",
                                "tiny_runner_videogen_plain.pt", "tiny_runner_videogen_ddim.pt", "tiny_runner_videogen_fpndm.pt"])
def test_three_edits_of_integration_md_against_the_real_runner(golden_dir, capsys, fx):
    """INTEGRATION.md section 2 end to end, against frames the REAL `NCSNRunner.video_gen` produced (oracle/gen_runner_golden.py drove the
    real `runners/ncsn_runner.py` -- get_model, get_sampler, the block loop :1476-1569 -- on the CPU; the module cannot travel to the GPU
    box, its output can): edit 1 `get_model` -> HipScoreNet, edit 2 `get_sampler` -> this package's, edit 3 the block loop -> `video_gen`,
    called with the kwargs the real runner passed (`verbose=True, log=True`: the logging host loop), on the real runner's clips, block
    inits and step noise.  Frames at 1e-4; the `verbose` lines (models/__init__.py:304-306) are the reference's, number for number."""
    import re
    from mcvd_pytorch_amd import runner as r
    from mcvd_pytorch_amd.samplers import ddim_sampler, ddpm_sampler, fpndm_sampler, get_sampler
    g = torch.load(os.path.join(golden_dir, fx), weights_only=False)
    config, sd, net = _net(g["config_name"])                                  # edit 1
    config.sampling.num_frames_pred, config.sampling.subsample = g["nfp"], g["subsample"]
    for k_, v_ in g.get("overrides", {}).items():                             # init_prev_t (blocks restart from the previous frames, re-noised) / one_frame_at_a_time
        setattr(config.sampling, k_, v_)
    version = g.get("version", "DDPM")                                        # model.version picks the sampler, as NCSNRunner.get_sampler does (:2702-2714)
    config.model.version = version
    bound = get_sampler(config)                                               # edit 2
    assert bound.func is {"DDPM": ddpm_sampler, "DDIM": ddim_sampler, "FPNDM": fpndm_sampler}[version] and bound.keywords == {"config": config}
    batch = g["clips"][g["order"]]
    real_t = r.data_transform(config, batch)
    real, cond, cond_mask = r.conditioning_fn(config, real_t, num_frames_pred=g["nfp"], prob_mask_cond=0.0, prob_mask_future=0.0)
    blk = [0]

    def sampler(x, scorenet, **kw):
        b = blk[0]
        blk[0] += 1
        extra = dict(noise=g["step_noise"][b].cuda()) if version == "DDPM" else {}          # (DDIM / F-PNDM draw nothing here)
        return bound(x, scorenet, n_steps_each=0, step_lr=0.0, **extra, **kw)
    kw = g["sampler_kwargs"][0]
    pred = r.video_gen(config, net, cond, num_frames_pred=g["nfp"], sampler=sampler, verbose=kw["verbose"], log=kw["log"],
                       init_noise_fn=lambda i, shp, dev: g["z_init"][i].to(dev))                    # edit 3
    assert blk[0] == 3 and pred.is_cuda
    err = (pred.cpu() - g["pred_raw"]).abs().max().item()
    # (the `plain` fixture: denoise = False, clip_before = False -- |frames| reach 1e3, the gate is 1e-5 of their range as for the surface cases)
    tol = 1e-4 if g.get("overrides", {}).get("clip_before", True) else 1e-5 * float(g["pred_raw"].abs().max())
    if version != "DDPM":      # deterministic samplers: no clip of the last iterate, no noise -- the three-block chain amplifies fp32 rounding; 3 x the reference's own drift
        tol = 3.0 * g["ref32_vs_ref64_max_abs"]
    assert 0.0 < g["ref32_vs_ref64_max_abs"] <= tol / 3 * (1 + 1e-6)
    assert err <= tol, f"three-edit integration vs the real runner's frames: {err:.3e} (gate {tol:.1e})"
    assert (r.inverse_data_transform(config, pred).cpu() - g["pred01"]).abs().max().item() <= tol
    # the verbose lines: same text, same step counters, the three norms to 1e-3 relative
    mine = [ln for ln in capsys.readouterr().out.splitlines() if ln.startswith(version + ": ")]
    assert len(mine) == len(g["log_lines"]) and len(mine) in (0, 27, 30)       # (t_min skips step 0 of every call: nine lines per block; F-PNDM prints none)
    num = re.compile(version + r": (\d+)/(\d+), grad_norm: ([-0-9.e+]+), image_norm: ([-0-9.e+]+), grad_mean_norm: ([-0-9.e+]+)$")
    for a, b in zip(mine, g["log_lines"]):
        ma, mb = num.match(a), num.match(b)
        assert ma and mb, (a, b)
        assert ma.group(1, 2) == mb.group(1, 2)
        for i in (3, 4, 5):
            va, vb = float(ma.group(i)), float(mb.group(i))
            assert abs(va - vb) <= 1e-3 * abs(vb), (a, b)


@pytest.mark.parametrize("cfg,B,kind,kw", [
    ("tiny", 3, "ddpm", dict(subsample_steps=10)),
    ("tiny", 2, "ddpm", dict(subsample_steps=10, t_min=0.35)),                 # skipped steps: the table holds the executed ones only
    ("tiny", 2, "ddim", dict(subsample_steps=10, denoise=False)),              # no denoise row
    ("tiny_spade", 2, "ddpm", dict(subsample_steps=10)),                       # SPADE: coef2 tables derived from the row
    ("smmnist_big5_ngf96", 2, "ddpm", dict(subsample_steps=20)),
])
@pytest.mark.parametrize("graph", [0, 1])
def test_time_embedding_table_of_a_sampler_call_is_bit_identical(cfg, B, kind, kw, graph):
    """Option temb_table (default on): a device-loop sampler call knows the labels of all its forwards before the first one
    (models/__init__.py:229-237, :283, :332) and the time MLP + Dense_0 projections depend on nothing else (ncsnpp_more.py:273-280,
    layerspp.py:521): they run once per call for all L (+ 1) labels and each forward copies its row.  Same kernels, independent rows: the
    frames must be BIT-IDENTICAL to the per-forward form (temb_table = 0), with and without hipGraph replay."""
    from mcvd_pytorch_amd.samplers import ddim_sampler, ddpm_sampler
    config, sd, net = _net(cfg)
    net.set_option("graph", graph)
    x, cond = synth.make_inputs(config, B, seed=0)
    noise = synth.make_noise(config, B, 21, seed=2).cuda()
    sampler = ddpm_sampler if kind == "ddpm" else ddim_sampler
    outs = []
    for tab in (1, 0, 1):
        net.set_option("temb_table", tab)
        outs.append(sampler(x.cuda(), net, cond=cond.cuda(), final_only=True, verbose=False, log=False, noise=noise, **kw).clone())
    assert torch.equal(outs[0], outs[1]), f"{float((outs[0] - outs[1]).abs().max()):.3e}"
    assert torch.equal(outs[0], outs[2])
    # and a plain forward after the call computes its own embedding again (the table is a property of the call)
    t = torch.tensor([700, 20, 333][:B]).cuda()
    a = net(x.cuda(), t, cond=cond.cuda()).clone()
    net.set_option("temb_table", 0)
    b = net(x.cuda(), t, cond=cond.cuda()).clone()
    assert torch.equal(a, b)


def test_ddim_gamma_renoise_vs_reference_golden(golden_dir, capsys):
    """`ddim_sampler(gamma=True, t_min > 0)` on a model.gamma net: the re-noise draw of the first executed step is a standardised Gamma variate
    (models/__init__.py:144-151; the runner passes `gamma=config.model.gamma` to whichever sampler it bound, ncsn_runner.py:1518).  The
    reference run's raw draws replayed (fixture from the REAL sampler); without t_min the kwarg changes nothing but the log prefix, and the
    whole loop stays on the device."""
    from mcvd_pytorch_amd.samplers import ddim_sampler
    g = torch.load(os.path.join(golden_dir, "tiny_gamma_ddim_b2.pt"), weights_only=False)
    config, sd, net = _net(g["config_name"])
    x, cond = synth.make_inputs(config, g["batch"], seed=0)
    for fo in (True, False):
        with _RunSpy() as spy:
            out = ddim_sampler(x.cuda(), net, cond=cond.cuda(), final_only=fo, subsample_steps=10, gamma=True, t_min=0.35, verbose=False, log=False,
                               noise=g["step_raw_tmin"].cuda(), cond_noise=g["cond_z_tmin"].cuda())
        assert not spy.calls                                   # a gamma re-noise draw: the host loop
        err = (out[-1:].cpu() - g["sampler_tmin"]).abs().max().item()
        assert err <= 3e-4, err                                # (the gate of the DDPM gamma fixture: test_gamma_sampler_vs_reference_golden has the reasons)
    # no t_min: no draw at all -- device loop, same frames as gamma=False, log prefix "DDIM gamma"
    cz = g["cond_z_tmin"]
    cz11 = torch.cat([cz, cz[:1]], dim=0).cuda()               # 11 forwards without t_min
    with _RunSpy() as spy:
        a = ddim_sampler(x.cuda(), net, cond=cond.cuda(), final_only=True, subsample_steps=10, gamma=True, verbose=False, log=False, cond_noise=cz11)
        b = ddim_sampler(x.cuda(), net, cond=cond.cuda(), final_only=True, subsample_steps=10, gamma=False, verbose=False, log=False, cond_noise=cz11)
    assert len(spy.calls) == 2 and torch.equal(a, b)
    capsys.readouterr()
    ddim_sampler(x.cuda(), net, cond=cond.cuda(), final_only=True, subsample_steps=10, gamma=True, verbose=True, log=False, cond_noise=cz11)
    lines = [ln for ln in capsys.readouterr().out.splitlines() if ln.startswith("DDIM")]
    assert len(lines) == 10 and all(ln.startswith("DDIM gamma: ") for ln in lines)

"""GPU parity AT THE BATCH THE BENCH RUNS (VERDICT r3 "What's weak"): every other GPU test uses B <= 6, while at the per-GPU batches of
BASELINE.json (64 / 32 / 16 / 8) the arena is several GB -- every `off * B` product and every `b * C * HW` index crosses 2^32 bytes, the
8x8 Winograd regions pair samples (2 * region), the 1x1 GEMM tiles span images and the persistent Winograd workgroups walk item ranges
that only exist at that size.  Rows are independent (GroupNorm and attention are per sample), so a large-batch forward can be held to
the small-batch evidence row by row:

  * rows 0..nb-1 of the big batch against the REAL reference's output (the committed fixture of the config, 1e-4 * max|eps|) and, where
    the CPU oracle is affordable, every module tap of those rows against the oracle (rtol 1e-4 + atol 2e-5 * scale);
  * rows {B/2 - 1, B - 2, B - 1} against the SAME rows recomputed at B = 3 under the SAME kernel table (the committed bench table where
    there is one, else the table the autotuner just produced at B): same kernels, same per-sample arithmetic -> held to 1e-6 * max|eps|
    (they are bit-equal unless a kernel's summation order depends on the batch);
  * a 5-step ddpm_sampler with an injected noise sequence whose first two rows reproduce a B = 2 run of the same rows.
"""
import json
import os

import pytest
import torch

from oracle import synth, unet_ref

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

# (config, per-GPU batch of BASELINE.json, fixture, rows the fixture holds, module taps vs the oracle?)
BIG = [
    ("smmnist_big5_ngf96", 64, "smmnist_big5_ngf96_b2.pt", 2, True),          # the headline workload (BASELINE config 2)
    ("smmnist_big5", 64, "smmnist_big5_b2.pt", 2, False),                     # config 1 at the bench's batch
    ("kth64_big_ngf128", 32, "kth64_big_ngf128_b2_fwd.pt", 2, True),          # config 3
    ("bair_big_spade", 16, "bair_big_spade_b2_fwd.pt", 2, False),             # config 4 (SPADE)
    ("cityscapes_big", 8, "cityscapes_big_b1_fwd.pt", 1, False),              # config 5 (128x128, five levels)
]


def _net(name):
    from mcvd_pytorch_amd.scorenet import HipScoreNet
    config = synth.make_config(name)
    config.device = "cuda:0"
    sd = synth.make_state_dict(config, seed=123)
    net = HipScoreNet(config)
    net.load_state_dict(sd, strict=True)
    return config, sd, net.eval()


def _table(net, name, B):
    """The committed bench table of (config, B) installed on `net` (and returned), else None: the autotuner then measures at B."""
    path = os.path.join(ROOT, "profiles", f"tune_{name}_B{B}_bf16x3.json")
    if not os.path.exists(path):
        return None
    table = json.load(open(path))[str(B)]
    net.set_tuning(B, table)
    return table


def _timesteps(g, nb, B):
    """The fixture's timesteps for its rows, then distinct ones for the rest (every row has its own time embedding)."""
    t = torch.tensor([(311 * (b + 1)) % 1000 for b in range(B)])
    t[:nb] = g["fwd_t"][:nb]
    return t


@pytest.mark.parametrize("name,B,fx,nb,taps_too", BIG, ids=[c[0] + "_B%d" % c[1] for c in BIG])
def test_forward_at_the_benchmarked_batch(golden_dir, name, B, fx, nb, taps_too):
    from tests.hiputil import module_output
    torch.set_num_threads(min(os.cpu_count() or 1, 16))
    g = torch.load(os.path.join(golden_dir, fx), weights_only=False)
    config, sd, net = _net(name)
    pinned = _table(net, name, B)
    x, cond = synth.make_inputs(config, B, seed=0)
    t = _timesteps(g, nb, B)
    eps = net(x.cuda(), t.cuda(), cond=cond.cuda())
    torch.cuda.synchronize()
    assert torch.isfinite(eps).all()
    table = pinned or net.get_tuning(B)
    if pinned:
        # what the bench runs is what is tested: under the committed table every conv op must have run the kernel family the table names
        # (mcvd_model_op_kernel reports what really executed; a hint that does not serve a launch falls back silently otherwise)
        import ctypes as C
        from mcvd_pytorch_amd import _lib
        n = _lib.lib.mcvd_model_profile_read(net._model, None, None, None, None, None, 0)
        assert n == len(table)
        info, fams = (C.c_int * 8)(), set()
        for i in range(n):
            _lib.check(_lib.lib.mcvd_model_op_info(net._model, i, info), "op_info")
            if info[0] != 3 or table[i][0] < 0:
                continue
            ran = _lib.lib.mcvd_model_op_kernel(net._model, i)
            if ran == -1:
                continue                 # cond-only (SPADE prep) convs run once per cond, not in this forward's record
            assert ran == table[i][0], f"{name} B={B} op {i} ({info[2]}x{info[2]} {info[4]}->{info[5]} @{info[3]}): ran kernel {ran}, the committed table names {table[i][0]}"
            fams.add(ran)
        assert fams & {10, 11, 16, 17, 18, 19, 20} and 15 in fams, f"{name}: kernel families that ran: {sorted(fams)}"
    eps_c = eps.cpu()
    # ---- rows 0..nb-1 against the real reference's output
    if "fwd_eps" in g:
        want = g["fwd_eps"][:nb]
        err = (eps_c[:nb] - want).abs().max().item()
        assert err <= 1e-4 * want.abs().max().item(), f"{name} B={B}: rows 0..{nb - 1} vs the reference fixture: {err:.3e}"
    else:                                        # strided probe of the flattened [nb, ...] output
        p = g["fwd_eps_probe"]
        got = eps_c[:nb].reshape(-1).double()[p["idx"]].float()
        err = (got - p["sample"]).abs().max().item()
        assert err <= 1e-4 * p["sample"].abs().max().item(), f"{name} B={B}: rows 0..{nb - 1} vs the reference probe: {err:.3e}"
    # ---- every module tap of those rows against the oracle
    if taps_too:
        taps = {}
        with torch.no_grad():
            ref = unet_ref.unet_forward(sd, config, x[:nb], t[:nb], cond[:nb], taps=taps)
        bad = []
        for i in sorted(taps):
            if i == 0 or i == len(taps) - 1:
                continue
            want = unet_ref.silu(taps[1]) if i == 1 else taps[i]
            try:
                got = module_output(net, i, B)[:nb]
            except RuntimeError:
                continue            # module without a workspace output
            sc = max(want.abs().max().item(), 1e-6)
            e = (got.cpu() - want).abs().max().item()
            if e > 2e-5 + 1e-4 * sc:
                bad.append((i, e, sc))
        assert not bad, f"{name} B={B}: modules off at rows 0..{nb - 1} (index, max-abs err, scale): {bad[:8]}"
        assert (eps_c[:nb] - ref).abs().max().item() <= 1e-4 * ref.abs().max().item()
    # ---- far rows against the same rows at B = 3 under the same table
    rows = sorted({B // 2 - 1, B - 2, B - 1})
    net.set_tuning(len(rows), table)
    eps3 = net(x[rows].cuda(), t[rows].cuda(), cond=cond[rows].cuda()).cpu()
    sc = eps3.abs().max().item()
    err = (eps_c[rows] - eps3).abs().max().item()
    assert err <= 1e-6 * sc, f"{name}: rows {rows} of the B={B} forward differ from the same rows at B={len(rows)} under the same table by {err:.3e} (scale {sc:.3e})"
    # and the other way round: every row of the big batch is finite and of the scale of the checked ones (a mis-indexed tile writes zeros / garbage)
    per_row = eps_c.flatten(1).abs().max(dim=1).values
    assert (per_row > 0.05 * sc).all() and (per_row < 20 * sc).all(), f"{name} B={B}: per-row max|eps| out of family: {per_row.tolist()}"


@pytest.mark.parametrize("name,B", [("smmnist_big5_ngf96", 64), ("kth64_big_ngf128", 32), ("cityscapes_big", 8)])
def test_sampler_at_the_benchmarked_batch(name, B):
    """5 DDPM steps + the denoise forward (models/__init__.py:282-333) at the bench's batch with an injected noise sequence: rows 0-1 must
    reproduce a B = 2 run of the same rows under the same kernel table, on the device loop."""
    from mcvd_pytorch_amd.samplers import ddpm_sampler
    config, sd, net = _net(name)
    pinned = _table(net, name, B)
    x, cond = synth.make_inputs(config, B, seed=0)
    noise = synth.make_noise(config, B, 6, seed=2)
    kw = dict(denoise=True, subsample_steps=5, clip_before=True, verbose=False, log=False, final_only=True)
    out = ddpm_sampler(x.cuda(), net, cond=cond.cuda(), noise=noise.cuda(), **kw)[-1].cpu()
    assert torch.isfinite(out).all() and out.shape == x.shape
    table = pinned or net.get_tuning(B)
    net.set_tuning(2, table)
    out2 = ddpm_sampler(x[:2].cuda(), net, cond=cond[:2].cuda(), noise=noise[:, :2].contiguous().cuda(), **kw)[-1].cpu()
    err = (out[:2] - out2).abs().max().item()
    assert err <= 1e-5, f"{name}: rows 0-1 of the B={B} sampler differ from the B=2 run by {err:.3e}"
    # rows are independent: no row may be a copy of another one (a wrong sample stride would do that)
    flat = out.flatten(1)
    d = (flat[1:] - flat[:-1]).abs().max(dim=1).values
    assert (d > 1e-3).all()

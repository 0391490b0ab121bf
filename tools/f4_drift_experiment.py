"""CPU experiment (zero GPU minutes): what would Winograd F(4x4,3x3) do to the parity contract?

The 3x3 convs of the 64x64 / 32x32 layers of the CPU oracle (oracle/unet_ref.py) are replaced by an EMULATION of a Winograd kernel:
weights transformed in fp64 and rounded to fp32 (a pack-time step), input and output transforms in fp32, the channel contraction in fp32
(torch CPU matmul; a kernel with exact piece products and an fp32 accumulator does no better).  The 100-step DDPM sampler of BASELINE
config 2 (the headline workload, B = 2, injected noise) is then run through it and compared with (i) the reference's own fixture
(tests/golden/smmnist_big5_ngf96_b2.pt, the 1e-4 gate of the parity tests) and (ii) the same oracle with plain F.conv2d.  F(2x2,3x3),
the form every production kernel uses, runs through the same emulation as the control.

    python tools/f4_drift_experiment.py [f2|f4|both] [steps]      -> profiles/r05_f4x4_drift_experiment.txt
"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402

from oracle import sampler_ref, synth, unet_ref  # noqa: E402

MATS = {
    2: dict(Bt=[[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]],
            G=[[1, 0, 0], [0.5, 0.5, 0.5], [0.5, -0.5, 0.5], [0, 0, 1]],
            At=[[1, 1, 1, 0], [0, 1, -1, -1]]),
    4: dict(Bt=[[4, 0, -5, 0, 1, 0], [0, -4, -4, 1, 1, 0], [0, 4, -4, -1, 1, 0], [0, -2, -1, 2, 1, 0], [0, 2, -1, -2, 1, 0],
                [0, 4, 0, -5, 0, 1]],
            G=[[1 / 4, 0, 0], [-1 / 6, -1 / 6, -1 / 6], [-1 / 6, 1 / 6, -1 / 6], [1 / 24, 1 / 12, 1 / 6], [1 / 24, -1 / 12, 1 / 6], [0, 0, 1]],
            At=[[1, 1, 1, 1, 1, 0], [0, 1, -1, 2, -2, 0], [0, 1, 1, 4, 4, 0], [0, 1, -1, 8, -8, 1]]),
}
_wcache = {}


def wino_conv(x, w, b, m):
    """3x3 conv, stride 1, zero padding 1, by Winograd F(m x m, 3x3) -- fp32 transforms around an fp32 contraction."""
    mats = MATS[m]
    t = m + 2
    Bt = torch.tensor(mats["Bt"], dtype=x.dtype)
    At = torch.tensor(mats["At"], dtype=x.dtype)
    key = (w.data_ptr(), m, x.dtype)
    if key not in _wcache:                       # U = G g G^T in fp64, rounded once (the weight pack)
        G = torch.tensor(mats["G"], dtype=torch.float64)
        _wcache[key] = torch.einsum("ij,ocjk,lk->iloc", G, w.double(), G).to(x.dtype).contiguous()        # [t, t, Cout, Cin]
    U = _wcache[key]
    B_, C, H, W = x.shape
    nty, ntx = H // m, W // m
    xp = F.pad(x, (1, 1, 1, 1))
    d = xp.unfold(2, t, m).unfold(3, t, m)                                     # [B, C, nty, ntx, t, t]
    V = torch.einsum("ij,bcyxjk,lk->ilcbyx", Bt, d, Bt)                         # B^T d B                       [t, t, C, B, nty, ntx]
    M = torch.matmul(U.reshape(t * t, U.shape[2], C), V.reshape(t * t, C, -1))  # per position: [Cout, Cin] x [Cin, tiles]
    M = M.reshape(t, t, U.shape[2], B_, nty, ntx)
    Y = torch.einsum("ij,jkobyx,lk->boyixl", At, M, At)                         # A^T M A                       [B, Cout, nty, m, ntx, m]
    return Y.reshape(B_, U.shape[2], H, W) + b.reshape(1, -1, 1, 1)


def run(mode, steps, config, sd, x, cond, noise):
    orig = unet_ref.conv2d
    stats = dict(n=0, worst=0.0)

    def patched(xx, w, b):
        if mode and w.shape[-1] == 3 and xx.shape[-1] in (32, 64) and xx.shape[-1] % 4 == 0:
            y = wino_conv(xx, w, b, mode)
            if stats["n"] < 40:                  # per-layer error against an fp64 convolution, first forward only
                ref = F.conv2d(xx.double(), w.double(), b.double(), padding=1)
                stats["worst"] = max(stats["worst"], float((y.double() - ref).abs().max() / ref.abs().max()))
                y32 = F.conv2d(xx, w, b, padding=1)
                stats["plain"] = max(stats.get("plain", 0.0), float((y32.double() - ref).abs().max() / ref.abs().max()))
            stats["n"] += 1
            return y
        return orig(xx, w, b)
    unet_ref.conv2d = patched
    try:
        net = unet_ref.OracleScoreNet(config, sd)
        k = [0]

        def fn(i, like):
            k[0] += 1
            return noise[k[0] - 1].to(like)
        t0 = time.time()
        out = sampler_ref.sample(x.clone(), net, cond=cond, kind="ddpm", final_only=True, denoise=True, subsample_steps=steps,
                                 clip_before=True, noise_fn=fn)
        return out, stats, time.time() - t0
    finally:
        unet_ref.conv2d = orig


def main():
    which = sys.argv[1] if len(sys.argv) > 1 else "both"
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 100
    torch.set_num_threads(os.cpu_count())
    g = torch.load(os.path.join(ROOT, "tests", "golden", "smmnist_big5_ngf96_b2.pt"), weights_only=False)
    config = synth.make_config(g["config_name"])
    sd = synth.make_state_dict(config, seed=123)
    x, cond = synth.make_inputs(config, g["batch"], seed=0)
    noise = synth.make_noise(config, g["batch"], steps + 1, seed=2)
    ref = g["sampler_ddpm_100"]["result"] if steps == 100 else None
    lines = [f"# tools/f4_drift_experiment.py: BASELINE config 2 ({g['config_name']}), B = {g['batch']}, DDPM {steps} steps + denoise, injected noise;",
             "# the 3x3 convs of the 64x64 and 32x32 layers (35 of 58 per forward, 78 % of the conv flops) through an fp32 Winograd EMULATION",
             "# (weights transformed in fp64 and rounded to fp32; fp32 input / output transforms; fp32 contraction), everything else as the oracle.",
             "# 'vs fixture' = max |final frames - the REAL reference's frames| (tests/golden, gate of the parity tests: 1e-4, data range [-1, 1])"]
    base, _, dt = run(0, steps, config, sd, x, cond, noise)
    lines.append(f"plain oracle (F.conv2d)        : vs fixture {float((base - ref).abs().max()) if ref is not None else float('nan'):.3e}   ({dt:.0f} s)")
    for m in ([2, 4] if which == "both" else [int(which[1])]):
        out, st, dt = run(m, steps, config, sd, x, cond, noise)
        lines.append(f"Winograd F({m}x{m},3x3) emulation : vs fixture {float((out - ref).abs().max()) if ref is not None else float('nan'):.3e}   "
                     f"vs plain oracle {float((out - base).abs().max()):.3e}   worst single layer vs an fp64 conv {st['worst']:.2e} of max|y| "
                     f"(F.conv2d fp32 on the same layers: {st.get('plain', 0):.2e})   [{st['n']} emulated convs, {dt:.0f} s]")
    txt = "\n".join(lines) + "\n"
    print(txt)
    if steps == 100 and which == "both":
        open(os.path.join(ROOT, "profiles", "r05_f4x4_drift_experiment.txt"), "w").write(txt)


if __name__ == "__main__":
    main()

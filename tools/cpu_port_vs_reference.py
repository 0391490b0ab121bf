"""Build-container tool (needs /root/reference; NOT run on the GPU box): time the REAL reference sampler
(`models.ddpm_sampler` + `models.better.ncsnpp_more.UNetMore_DDPM`, imported read-only) next to the CPU port that `bench.py`'s
`cpu_baseline` leg times on the GPU box (`oracle/`), on the same synthetic weights and inputs, same thread count, interleaved.
BASELINE.md section 3 asks for the reference itself as the CPU baseline; the Python reference cannot travel to the GPU box, so this
file pins the ratio  port / reference  where both exist.  Writes profiles/r05_cpu_port_vs_reference.{txt,json}.

    PYTHONDONTWRITEBYTECODE=1 python tools/cpu_port_vs_reference.py [config] [B] [forwards]
"""
import json
import os
import sys
import time

os.environ.setdefault("PYTHONDONTWRITEBYTECODE", "1")
sys.dont_write_bytecode = True
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

from oracle import sampler_ref, synth, unet_ref  # noqa: E402


class Stop(Exception):
    pass


def timed_sampler(run, net_call, n_fwd):
    """Run a sampler whose scorenet is `net_call` until n_fwd + 1 forwards happened; return seconds per forward (first one = warm-up)."""
    t = []

    def counting(x, y, cond=None):
        if len(t) > n_fwd:
            raise Stop()
        t0 = time.perf_counter()
        out = net_call(x, y, cond=cond)
        t.append(time.perf_counter() - t0)
        return out
    t_all = time.perf_counter()
    try:
        run(counting)
    except Stop:
        pass
    wall = time.perf_counter() - t_all
    return sum(t[1:]) / max(len(t) - 1, 1), t, wall


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "smmnist_big5_ngf96"
    B = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    n_fwd = int(sys.argv[3]) if len(sys.argv) > 3 else 6
    threads = os.cpu_count() or 1
    torch.set_num_threads(threads)
    sys.path.insert(0, "/root/reference")
    import models as ref_models
    from models.better.ncsnpp_more import UNetMore_DDPM
    config = synth.make_config(name)
    config.device = "cpu"
    sd = synth.make_state_dict(config, seed=123)
    ref_net = UNetMore_DDPM(config).eval()
    miss = ref_net.load_state_dict(sd, strict=False)
    assert not miss.unexpected_keys and all(k in ("betas", "alphas", "alphas_prev", "unet.sigmas") for k in miss.missing_keys), miss
    port_net = unet_ref.OracleScoreNet(config, sd)
    x, cond = synth.make_inputs(config, B, seed=0)
    sub = config.sampling.subsample
    lines = [f"# tools/cpu_port_vs_reference.py {name} B={B}: reference = /root/reference models.ddpm_sampler + UNetMore_DDPM (imported, unmodified),",
             f"# port = oracle/sampler_ref.sample + oracle/unet_ref.OracleScoreNet (what bench.py's cpu_baseline leg times on the GPU box);",
             f"# build container, {threads} threads of {os.cpu_count()} ({open('/proc/cpuinfo').read().split('model name')[1].split(':')[1].splitlines()[0].strip()}), torch {torch.__version__},",
             f"# same synthetic weights (seed 123) / inputs; DDPM subsample {sub}, the first {n_fwd + 1} forwards of the sampler call (first = warm-up), two rounds, interleaved"]
    res = {"reference": [], "port": []}
    with torch.no_grad():
        # one forward each on identical inputs: the port computes what the reference computes
        t = torch.full((B,), 500).long()
        e_ref, e_port = ref_net(x, t, cond=cond), port_net(x, t, cond=cond)
        diff = float((e_ref - e_port).abs().max())
        lines.append(f"one forward, same inputs: max |reference - port| = {diff:.3e} (max |eps| {float(e_ref.abs().max()):.3f})")
        for rnd in range(2):
            per, ts, wall = timed_sampler(lambda c: ref_models.ddpm_sampler(
                x.clone(), _Wrap(c, ref_net), cond=cond, final_only=True, denoise=True, subsample_steps=sub, clip_before=True,
                verbose=False, log=False), lambda xx, yy, cond=None: ref_net(xx, yy, cond=cond), n_fwd)
            res["reference"].append(per)
            lines.append(f"round {rnd}: reference  {per * 1e3:8.1f} ms / forward   ({', '.join(f'{v:.3f}' for v in ts)})")
            per, ts, wall = timed_sampler(lambda c: sampler_ref.sample(
                x.clone(), _Wrap(c, port_net), cond=cond, kind="ddpm", final_only=True, denoise=True, subsample_steps=sub),
                lambda xx, yy, cond=None: port_net(xx, yy, cond=cond), n_fwd)
            res["port"].append(per)
            lines.append(f"round {rnd}: port       {per * 1e3:8.1f} ms / forward   ({', '.join(f'{v:.3f}' for v in ts)})")
    r, p = min(res["reference"]), min(res["port"])
    nf = config.data.num_frames
    fps_r, fps_p = B * nf / (r * (sub + 1)), B * nf / (p * (sub + 1))
    lines.append(f"best of two rounds: reference {r * 1e3:.1f} ms / forward = {fps_r:.4f} frames/s; port {p * 1e3:.1f} ms / forward = {fps_p:.4f} frames/s")
    lines.append(f"port_vs_reference (frames/s of the port / frames/s of the reference) = {fps_p / fps_r:.4f}")
    lines.append("# reading: a cpu_baseline value measured with the port on the GPU box, divided by this ratio, estimates the reference's own rate there")
    txt = "\n".join(lines) + "\n"
    print(txt)
    tag = "" if name == "smmnist_big5_ngf96" else "_" + name
    open(os.path.join(ROOT, "profiles", f"r05_cpu_port_vs_reference{tag}.txt"), "w").write(txt)
    json.dump(dict(config=name, batch=B, threads=threads, forwards_timed=n_fwd, reference_ms_per_forward=round(r * 1e3, 2),
                   port_ms_per_forward=round(p * 1e3, 2), port_vs_reference=round(fps_p / fps_r, 4), forward_max_abs_diff=diff,
                   torch=torch.__version__, where="build container (the Python reference does not travel to the GPU box)"),
              open(os.path.join(ROOT, "profiles", f"r05_cpu_port_vs_reference{tag}.json"), "w"), indent=1)


class _Wrap:
    """scorenet protocol around a counting callable."""

    def __init__(self, fn, net):
        self.fn, self.alphas, self.alphas_prev, self.betas = fn, net.alphas, net.alphas_prev, net.betas
        self.type = "v1"

    def __call__(self, x, y, cond=None):
        return self.fn(x, y, cond=cond)


if __name__ == "__main__":
    main()

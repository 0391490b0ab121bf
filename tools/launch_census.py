"""Per-level launch census of one instrumented forward: how many launches a forward makes at each resolution, by
kind, how long they take (HIP events around every op, option "profile") and how many of them are shorter than 20 us -- the regime where a
launch is bound by its own ramp-up and tail rather than by its work.  Run on the GPU box:

    python tools/launch_census.py bair_big_spade 16 > gpurun_out/census.txt
"""
import ctypes as C
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mcvd_pytorch_amd import HipScoreNet, _lib, ddpm_sampler, synthetic  # noqa: E402
from bench import make_config  # noqa: E402

NAMES = {0: "temb_mlp", 1: "dense_all", 2: "gn_coef", 3: "conv", 4: "fir2", 5: "attention", 6: "nearest", 7: "coef2", 8: "spade_apply"}


def main():
    name, B = sys.argv[1], int(sys.argv[2])
    config = make_config(name)
    config.device = "cuda:0"
    net = HipScoreNet(config)
    net.load_state_dict(synthetic.random_state_dict(net, seed=123), strict=True)
    table = os.path.join(ROOT, "profiles", f"tune_{name}_B{B}_bf16x3.json")
    if os.path.exists(table):
        net.load_tuning(table)
    net.set_option("graph", 1)
    x, cond = synthetic.random_inputs(config, 0, B)
    x, cond = x.cuda(), cond.cuda()
    kw = dict(final_only=True, denoise=True, subsample_steps=20, clip_before=True, verbose=False, log=False, seed=1)
    ddpm_sampler(x, net, cond=cond, **kw)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    ddpm_sampler(x, net, cond=cond, **kw)
    e1.record()
    torch.cuda.synchronize()
    ms_fwd_graph = e0.elapsed_time(e1) / 21
    net.set_option("profile", 1)
    ddpm_sampler(x, net, cond=cond, **kw)
    torch.cuda.synchronize()
    n = _lib.lib.mcvd_model_profile_read(net._model, None, None, None, None, None, 0)
    kinds, kss = (C.c_int * n)(), (C.c_int * n)()
    ms, fl, by = (C.c_double * n)(), (C.c_double * n)(), (C.c_double * n)()
    _lib.lib.mcvd_model_profile_read(net._model, kinds, kss, ms, fl, by, n)
    info = (C.c_int * 8)()
    levels = {}
    for i in range(n):
        if ms[i] == 0.0:
            continue                      # cond-only prep ops run once per call, not per forward
        _lib.lib.mcvd_model_op_info(net._model, i, info)
        H = info[3]
        kind = NAMES.get(kinds[i], str(kinds[i])) + (f"{kss[i]}x{kss[i]}" if kinds[i] == 3 else "")
        d = levels.setdefault(H, {})
        e = d.setdefault(kind, [0, 0.0, 0])
        e[0] += 1
        e[1] += ms[i] * 1e3
        e[2] += ms[i] * 1e3 < 20.0
    if os.environ.get("PER_OP"):          # one line per conv: where inside a level the time goes
        tab = dict(enumerate(net.get_tuning(B)))
        print(f"# {name}, B = {B}: every conv of the instrumented forward (op, H, Cin -> Cout, ks, residual, kernel id / cout tile of the table, us, direct-form TFLOP/s)")
        for i in range(n):
            if kinds[i] != 3 or ms[i] == 0.0:
                continue
            _lib.lib.mcvd_model_op_info(net._model, i, info)
            print(f"  op {i:3d}  H {info[3]:3d}  {info[4]:4d} -> {info[5]:4d}  k{kss[i]}  res {info[6]}  pro {info[7]}  kernel {tab.get(i)}  {ms[i] * 1e3:8.1f} us  {fl[i] / ms[i] / 1e9:7.1f} TF/s")
    tot_n = sum(e[0] for d in levels.values() for e in d.values())
    tot_us = sum(e[1] for d in levels.values() for e in d.values())
    tot_short = sum(e[2] for d in levels.values() for e in d.values())
    short_us = 0.0
    print(f"# {name}, B = {B}: launch census of one forward (HIP events around every op; the graph-replayed forward takes {ms_fwd_graph * 1e3:.0f} us)")
    print(f"# level (H)  kind          launches   total us   avg us   launches < 20 us")
    for H in sorted(levels, reverse=True):
        for kind, (c, us, sh) in sorted(levels[H].items(), key=lambda kv: -kv[1][1]):
            print(f"  {H:4d}       {kind:13s} {c:6d} {us:10.1f} {us / c:8.1f} {sh:8d}")
    for i in range(n):
        if 0.0 < ms[i] * 1e3 < 20.0:
            short_us += ms[i] * 1e3
    print(f"# total: {tot_n} launches, {tot_us:.0f} us by events ({tot_us / ms_fwd_graph / 1e3:.3f} x the replayed forward); {tot_short} launches shorter than 20 us = "
          f"{short_us:.0f} us ({100 * short_us / tot_us:.1f} % of the event sum)")
    print(json.dumps(dict(config=name, B=B, launches=tot_n, event_sum_us=round(tot_us, 1), replayed_forward_us=round(ms_fwd_graph * 1e3, 1),
                          short_launches=tot_short, short_us=round(short_us, 1))))


if __name__ == "__main__":
    main()

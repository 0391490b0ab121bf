"""GPU diagnostics -- test infrastructure (uses the oracle as checker; writes gpurun_out/diag_*.txt): precision vs the
fp32/fp64 oracle, per-op timings, conv tile sweep.
    python tests/gpu_diag.py [precision] [ops] [sweep]
"""
import ctypes as C
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

torch.set_num_threads(min(os.cpu_count() or 1, 16))

from mcvd_pytorch_amd import HipScoreNet, _lib, ddim_sampler, ddpm_sampler, synthetic  # noqa: E402
from oracle import sampler_ref, synth, unet_ref  # noqa: E402

OUT = os.path.join(ROOT, "gpurun_out")
os.makedirs(OUT, exist_ok=True)


def mk(name):
    config = synth.make_config(name)
    config.device = "cuda:0"
    sd = synth.make_state_dict(config, seed=123)
    net = HipScoreNet(config)
    net.load_state_dict(sd, strict=True)
    return config, sd, net


def precision(f):
    from tests.hiputil import module_output
    for name, B, sub in (("tiny", 3, 10), ("smmnist_big5", 2, 100)):
        config, sd, net = mk(name)
        x, cond = synth.make_inputs(config, B, seed=0)
        t = torch.tensor([(37 * (b + 1)) % 1000 for b in range(B)]).long()
        taps32, taps64 = {}, {}
        with torch.no_grad():
            r32 = unet_ref.unet_forward(sd, config, x, t, cond, taps=taps32)
            sd64 = {k: v.double() for k, v in sd.items()}
            r64 = unet_ref.unet_forward(sd64, config, x.double(), t, cond.double(), taps=taps64)
        for naive in (0, 1):
            net.set_option("naive_conv", naive)
            net.set_option("naive_attn", naive)
            eps = net(x.cuda(), t.cuda(), cond=cond.cuda()).cpu()
            f.write(f"[{name} B={B} naive={naive}] forward: |eps|max {r64.abs().max():.3f}  hip-ref32 {(eps - r32).abs().max():.3e}  "
                    f"hip-ref64 {(eps.double() - r64).abs().max():.3e}  ref32-ref64 {(r32.double() - r64).abs().max():.3e}\n")
            rows = []
            for i in sorted(taps64):
                if i in (0,):
                    continue
                try:
                    got = eps if i == len(taps64) - 1 else module_output(net, i, B).cpu()
                except RuntimeError:
                    continue
                w64 = unet_ref.silu(taps64[1]) if i == 1 else taps64[i]
                w32 = unet_ref.silu(taps32[1]) if i == 1 else taps32[i]
                sc = w64.abs().max().item()
                rows.append((i, (got.double() - w64).abs().max().item() / sc, (w32.double() - w64).abs().max().item() / sc))
            f.write("   per-module relative max err (module: hip-vs-64 | ref32-vs-64): " +
                    "  ".join(f"{i}:{a:.1e}|{b:.1e}" for i, a, b in rows) + "\n")
        net.set_option("naive_conv", 0)
        net.set_option("naive_attn", 0)
        noise = synth.make_noise(config, B, sub + 1, seed=2)
        for kind, smp in (("ddpm", ddpm_sampler), ("ddim", ddim_sampler)):
            if name != "tiny" and kind == "ddim":
                continue
            outs = {}
            for dt in (torch.float32, torch.float64):
                k = [0]

                def fn(i, like):
                    k[0] += 1
                    return noise[k[0] - 1].to(like.dtype)
                o = sampler_ref.sample(x.to(dt), unet_ref.OracleScoreNet(config, sd, dtype=dt), cond=cond.to(dt), kind=kind,
                                       final_only=True, denoise=True, subsample_steps=sub, noise_fn=fn)
                outs[dt] = o
            got = smp(x.cuda(), net, cond=cond.cuda(), final_only=True, denoise=True, subsample_steps=sub, noise=noise.cuda(),
                      verbose=False, log=False).cpu()
            f.write(f"[{name} B={B}] {kind}-{sub}: hip-ref32 {(got - outs[torch.float32]).abs().max():.3e}  "
                    f"hip-ref64 {(got.double() - outs[torch.float64]).abs().max():.3e}  "
                    f"ref32-ref64 {(outs[torch.float32].double() - outs[torch.float64]).abs().max():.3e}\n")
        f.flush()


def ops(f, cfgname="smmnist_big5_ngf96", B=64):
    import bench
    config = bench.make_config(cfgname)
    config.device = "cuda:0"
    net = HipScoreNet(config)
    net.load_state_dict(synthetic.random_state_dict(net), strict=True)
    net.set_option("profile", 1)
    x, cond = synthetic.random_inputs(config, 0, B)
    x, cond = x.cuda(), cond.cuda()
    for _ in range(2):
        ddpm_sampler(x, net, cond=cond, final_only=True, subsample_steps=2, seed=1)
    n = _lib.lib.mcvd_model_profile_read(net._model, None, None, None, None, None, 0)
    kinds, kss = (C.c_int * n)(), (C.c_int * n)()
    ms, fl, by = (C.c_double * n)(), (C.c_double * n)(), (C.c_double * n)()
    assert _lib.lib.mcvd_model_profile_read(net._model, kinds, kss, ms, fl, by, n) == n
    names = {0: "temb", 1: "dense", 2: "gn", 3: "conv", 4: "fir", 5: "attn", 6: "near", 7: "coef2", 8: "apply"}
    info = (C.c_int * 8)()
    f.write(f"# {cfgname} B={B}: op, module, kind, ks, H, Cin, Cout, res, pro, us, TFLOP/s, GB/s(alg)\n")
    tot = 0.0
    for i in range(n):
        _lib.lib.mcvd_model_op_info(net._model, i, info)
        tot += ms[i]
        tune = f"s{(info[6] >> 4) & 15}c{(info[6] >> 8) & 15}" if info[6] >> 12 else "    "
        f.write(f"{i:4d} m{info[1]:3d} {names[info[0]]:5s} k{info[2]} H{info[3]:4d} ci{info[4]:5d} co{info[5]:5d} r{info[6] & 1} p{info[7]} {tune} "
                f"{ms[i] * 1e3:9.1f} us {fl[i] / max(ms[i], 1e-9) / 1e9:8.2f} TF {by[i] / max(ms[i], 1e-9) / 1e6:9.1f} GB/s\n")
    f.write(f"# total {tot:.3f} ms\n")
    f.flush()


def sweep(f):
    from tests.hiputil import Ctx
    ctx = Ctx()
    B = 64
    shapes = [(96, 96, 64, 3), (192, 96, 64, 3), (288, 96, 64, 3), (192, 192, 32, 3), (480, 192, 32, 3), (288, 288, 16, 3),
              (672, 288, 16, 3), (384, 384, 8, 3), (768, 384, 8, 3), (192, 576, 32, 1), (288, 864, 16, 1), (384, 1152, 8, 1),
              (96, 192, 32, 1), (768, 384, 8, 1)]
    f.write("# conv tile sweep at B=64: Cin Cout H ks res | shape0(256px) shape1(128px) shape2(64px split-K) [shape3 split-K+wdb, shape4 Winograd]: us, effective TFLOP/s (direct-conv flops) ; wdma=0 then wdma=1\n")
    shapes = [(c, o, h, k, r) for (c, o, h, k) in shapes for r in ((0, 1) if k == 3 and c == o else (0,))]
    for cin, cout, H, ks, use_res in shapes:
        x = torch.randn(B, cin, H, H, device="cuda")
        w = torch.randn(cout, cin, ks, ks, device="cuda") / (cin * ks * ks) ** 0.5
        b = torch.zeros(cout, device="cuda")
        coef = torch.ones(B, cin, 2, device="cuda")
        flops = 2.0 * B * H * H * cout * cin * ks * ks
        res = torch.randn(B, cout, H, H, device="cuda") if use_res else None
        line = f"{cin:4d} {cout:4d} {H:3d} k{ks} r{use_res} |"
        for wdma in (0, 1):
            ctx.opt("conv_wdma", wdma)
            for shape in ((0, 1, 2, 3, 4) if ks == 3 else (0, 1, 2)):
                if shape >= 3 and wdma == 0:
                    continue
                ctx.opt("conv_shape", shape)
                try:
                    for _ in range(2):
                        ctx.conv2d(x, w, b, coef=coef, act=1, res=res, scale=0.7)
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    for _ in range(5):
                        ctx.conv2d(x, w, b, coef=coef, act=1, res=res, scale=0.7)
                    e1.record()
                    torch.cuda.synchronize()
                    us = e0.elapsed_time(e1) * 1e3 / 5
                    line += f" {us:7.1f} us {flops / us / 1e6:6.1f} TF |"
                except RuntimeError as e:
                    line += f"  n/a ({str(e)[-40:]}) |"
            line += "|"
        f.write(line + "\n")
        f.flush()
    ctx.opt("conv_shape", -1)
    ctx.opt("conv_wdma", 1)




def phases(f):
    """Where a conv block's time goes: per-phase shader-cycle counters of wave 0 (mcvd_ctx_set_debug_buffer)."""
    from tests.hiputil import Ctx, P
    ctx = Ctx()
    B = 64
    names = ["prologue", "mfma", "barrier1(+vmcnt)", "stage-write", "barrier2", "splitK-reduce", "epilogue", "total"]
    cases = [(96, 96, 64, 3, 0, 0), (96, 96, 64, 3, 1, 0), (192, 192, 32, 3, 0, 0), (480, 192, 32, 3, 0, 0), (288, 288, 16, 3, 0, 1),
             (288, 288, 16, 3, 0, 2), (384, 384, 8, 3, 0, 2), (384, 384, 8, 3, 0, 3), (768, 384, 8, 3, 0, 2), (768, 384, 8, 3, 0, 3), (192, 576, 32, 1, 0, 1), (192, 192, 32, 1, 1, 1)]
    f.write("# conv phase breakdown, B=64 (cycles of wave 0, mean over blocks; % of the block's total)\n")
    for cin, cout, H, ks, use_res, shape in cases:
        x = torch.randn(B, cin, H, H, device="cuda")
        w = torch.randn(cout, cin, ks, ks, device="cuda") / (cin * ks * ks) ** 0.5
        b = torch.zeros(cout, device="cuda")
        coef = torch.ones(B, cin, 2, device="cuda")
        res = torch.randn(B, cout, H, H, device="cuda") if use_res else None
        dbg = torch.zeros(65536 * 8, dtype=torch.int64, device="cuda")
        ctx.opt("conv_shape", shape)
        for _ in range(2):
            ctx.conv2d(x, w, b, coef=coef, act=1, res=res, scale=0.7)
        _lib.check(_lib.lib.mcvd_ctx_set_debug_buffer(ctx.h, P(dbg)))
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        ctx.conv2d(x, w, b, coef=coef, act=1, res=res, scale=0.7)
        e1.record()
        torch.cuda.synchronize()
        _lib.check(_lib.lib.mcvd_ctx_set_debug_buffer(ctx.h, None))
        d = dbg.view(-1, 8).cpu().double()
        d = d[d[:, 7] > 0]
        m = d.mean(0)
        f.write(f"cin{cin} cout{cout} H{H} k{ks} res{use_res} shape{shape}: {d.shape[0]} blocks, kernel {e0.elapsed_time(e1) * 1e3:.0f} us, "
                f"block total {m[7]:.0f} cyc ({m[7] / 2400:.1f} us @2.4GHz) | " +
                " ".join(f"{n} {100 * m[i] / m[7]:.1f}%" for i, n in enumerate(names[:7])) + "\n")
        f.flush()
    ctx.opt("conv_shape", -1)


def wphases(f):
    """Winograd (conv_wino.cpp) coarse phase breakdown of one wave (env MCVD_DBG_WAVE), shader-clock units."""
    from tests.hiputil import Ctx, P
    ctx = Ctx()
    B = 64
    names = ["prologue", "K-loop", "-", "-", "-", "epilogue"]
    cases = [(96, 96, 64, 0), (192, 192, 32, 0)]
    f.write(f"# winograd phase breakdown, B=64, wave {os.environ.get('MCVD_DBG_WAVE', '0')} (mean over blocks)\n")
    ctx.opt("conv_shape", 4)
    for cin, cout, H, use_res in cases:
        x = torch.randn(B, cin, H, H, device="cuda")
        w = torch.randn(cout, cin, 3, 3, device="cuda") / (cin * 9) ** 0.5
        b = torch.zeros(cout, device="cuda")
        coef = torch.ones(B, cin, 2, device="cuda")
        res = torch.randn(B, cout, H, H, device="cuda") if use_res else None
        dbg = torch.zeros(65536 * 8, dtype=torch.int64, device="cuda")
        for _ in range(2):
            ctx.conv2d(x, w, b, coef=coef, act=1, res=res, scale=0.7)
        _lib.check(_lib.lib.mcvd_ctx_set_debug_buffer(ctx.h, P(dbg)))
        ctx.conv2d(x, w, b, coef=coef, act=1, res=res, scale=0.7)
        torch.cuda.synchronize()
        _lib.check(_lib.lib.mcvd_ctx_set_debug_buffer(ctx.h, None))
        d = dbg.view(-1, 8).cpu().double()
        d = d[d[:, 7] > 0]
        m = d.mean(0)
        nch = m[6]
        f.write(f"cin{cin} cout{cout} H{H} res{use_res}: {d.shape[0]} blocks, {nch:.0f} chunks, block total {m[7]:.0f} | " +
                " ".join(f"{n} {m[i]:.0f} ({100 * m[i] / m[7]:.1f}%)" for i, n in enumerate(names)) +
                f" | K loop per chunk: {m[1] / nch:.0f}\n")
        f.flush()
    ctx.opt("conv_shape", -1)


def sweep1(f):
    """Every 1x1 layer shape of config 2 at B=64 against every kernel candidate (direct tiles 0-2, all-DMA 16/32-channel chunks x cout tiles)."""
    from tests.hiputil import Ctx
    ctx = Ctx()
    B = 64
    shapes = [(192, 192, 64, 0, 0), (288, 96, 64, 0, 0), (192, 96, 64, 0, 0), (96, 96, 32, 0, 0), (96, 192, 32, 0, 0), (480, 192, 32, 0, 0),
              (384, 192, 32, 0, 0), (288, 192, 32, 0, 0), (288, 288, 32, 0, 0), (192, 576, 32, 1, 0), (192, 192, 32, 0, 1),
              (288, 864, 16, 1, 0), (288, 288, 16, 0, 1), (672, 288, 16, 0, 0), (384, 384, 16, 0, 0), (384, 1152, 8, 1, 0),
              (384, 384, 8, 0, 1), (768, 384, 8, 0, 0)]
    f.write("# 1x1 conv candidates at B=64: Cin Cout H coef res | candidate: us (TF)\n")
    for cin, cout, H, use_coef, use_res in shapes:
        x = torch.randn(B, cin, H, H, device="cuda")
        w = torch.randn(cout, cin, 1, 1, device="cuda") / cin ** 0.5
        b = torch.zeros(cout, device="cuda")
        coef = torch.ones(B, cin, 2, device="cuda") if use_coef else None
        res = torch.randn(B, cout, H, H, device="cuda") if use_res else None
        flops = 2.0 * B * H * H * cout * cin
        byts = 4.0 * B * H * H * (cin + cout * (2 if use_res else 1))
        line = f"{cin:4d} {cout:4d} {H:3d} c{use_coef} r{use_res} |"
        best = (1e9, "")
        cands = [(1, 0), (2, 0)] + [(5, c) for c in (3, 2, 1)] + [(6, c) for c in (2, 1)] + [(9, 1), (9, 2)]
        for shape, cot in cands:
            n32 = -(-cout // 32)
            if shape >= 5 and n32 % cot != 0:
                continue
            if shape == 9 and cot == 2 and n32 % 2 != 0:
                continue
            ctx.opt("conv_shape", shape)
            ctx.opt("conv_cot", cot)
            try:
                for _ in range(2):
                    ctx.conv2d(x, w, b, coef=coef, act=0, res=res, scale=0.7)
                if shape >= 5 and _lib.lib.mcvd_last_conv_kernel() != shape:
                    continue
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(5):
                    ctx.conv2d(x, w, b, coef=coef, act=0, res=res, scale=0.7)
                e1.record()
                torch.cuda.synchronize()
                us = e0.elapsed_time(e1) * 1e3 / 5
                line += f" s{shape}c{cot}:{us:6.1f}({flops / us / 1e6:5.1f})"
                if us < best[0]:
                    best = (us, f"s{shape}c{cot}")
            except RuntimeError as e:
                line += f" s{shape}c{cot}:n/a"
        f.write(line + f" || best {best[1]} {best[0]:.1f} us = {flops / best[0] / 1e6:.1f} TF, {byts / best[0] / 1e6:.2f} TB/s alg\n")
        f.flush()
    ctx.opt("conv_shape", -1)
    ctx.opt("conv_cot", 0)


def wexp(f):
    """Timing-only ablations of the Winograd K loop (conv_wino.cpp EXP builds, env MCVD_WINO_EXP): kernel time and the K-loop
    cycles per 16-channel chunk of one wave, per ablation."""
    from tests.hiputil import Ctx, P
    ctx = Ctx()
    B = 64
    labels = {0: "baseline", 1: "no transform (WRITE_V)", 2: "no patch activation/park (WRITE_P)", 3: "no WRITE_V, no WRITE_P",
              4: "no VMEM in the loop", 7: "MFMA + B reads + barrier only", 15: "MFMA + barrier only", 47: "MFMA only, no barrier",
              16: "everything but the MFMAs", 17: "no MFMA, no transform", 18: "no MFMA, no WRITE_P", 19: "no MFMA, no V, no P (VMEM only)",
              20: "no MFMA, no VMEM", 21: "WRITE_P only", 22: "transform only", 23: "barrier + loop overhead only", 32: "no chunk barrier (racy)", 256: "s_setprio 3-grp", 512: "s_setprio grp",
              768: "s_setprio 1 everywhere"}
    cases = [(96, 96, 64), (192, 192, 32), (480, 192, 32)]
    if os.environ.get("MCVD_WEXP_CASES", "all") != "all":
        cases = [cases[int(v)] for v in os.environ["MCVD_WEXP_CASES"].split(",")]
    if os.environ.get("MCVD_WEXP_ONLY"):
        keep = [int(v) for v in os.environ["MCVD_WEXP_ONLY"].split(",")]
        labels = {k: v for k, v in labels.items() if k in keep}
    f.write("# winograd K-loop ablations, B=64, conv_wino_kernel<3,2,false>; ideal MFMA time per chunk per SIMD = 4 waves x 24 x 64 = 6144 cycles\n")
    ctx.opt("conv_shape", 4)
    for cin, cout, H in cases:
        x = torch.randn(B, cin, H, H, device="cuda")
        w = torch.randn(cout, cin, 3, 3, device="cuda") / (cin * 9) ** 0.5
        b = torch.zeros(cout, device="cuda")
        coef = torch.ones(B, cin, 2, device="cuda")
        for e, lab in labels.items():
            os.environ["MCVD_WINO_EXP"] = str(e)
            for _ in range(2):
                ctx.conv2d(x, w, b, coef=coef, act=1, scale=0.7)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(4):
                ctx.conv2d(x, w, b, coef=coef, act=1, scale=0.7)
            e1.record()
            torch.cuda.synchronize()
            us = e0.elapsed_time(e1) * 1e3 / 4
            per = []
            for wv in (0, 15):
                os.environ["MCVD_DBG_WAVE"] = str(wv)
                dbg = torch.zeros(65536 * 8, dtype=torch.int64, device="cuda")
                _lib.check(_lib.lib.mcvd_ctx_set_debug_buffer(ctx.h, P(dbg)))
                ctx.conv2d(x, w, b, coef=coef, act=1, scale=0.7)
                torch.cuda.synchronize()
                _lib.check(_lib.lib.mcvd_ctx_set_debug_buffer(ctx.h, None))
                d = dbg.view(-1, 8).cpu().double()
                d = d[d[:, 7] > 0]
                m = d.mean(0)
                per.append((m[0].item(), m[1].item() / max(m[6].item(), 1), m[5].item(), m[7].item(), m[2].item(), m[3].item(), m[4].item()))
            f.write(f"cin{cin} cout{cout} H{H} exp{e:4d} {lab:40s}: kernel {us:7.1f} us | wave0 pro {per[0][0]:6.0f} loop/chunk {per[0][1]:6.0f} epi {per[0][2]:6.0f} total {per[0][3]:7.0f} [pro: issue {per[0][4]:5.0f} +mem {per[0][5]:5.0f} +P {per[0][6]:5.0f}]"
                    f" | wave15 pro {per[1][0]:6.0f} loop/chunk {per[1][1]:6.0f} epi {per[1][2]:6.0f} total {per[1][3]:7.0f}\n")
            f.flush()
    os.environ["MCVD_WINO_EXP"] = "0"
    ctx.opt("conv_shape", -1)


def w3exp(f):
    """conv_wino3.cpp (split-operand bf16 Winograd): kernel time and K-loop cycles per 16-channel chunk per ablation (env
    MCVD_WINO3_EXP), next to the fp32-MFMA Winograd kernel on the same layers."""
    from tests.hiputil import Ctx, P
    ctx = Ctx()
    B = 64
    kshape = int(os.environ.get("MCVD_WEXP_SHAPE", "10"))         # 10: conv_wino3.cpp (bf16x3), 12: conv_wino2h.cpp (f16x2)
    envname = "MCVD_WINO3_EXP" if kshape == 10 else "MCVD_WINO2H_EXP"
    # (the ablations exist in a -DMCVD_DIAG build only: python mcvd_pytorch_amd/csrc/build.py --diag; the production library ignores the env)
    labels = {0: "baseline", 1: "no transform (WRITE_V)", 2: "no patch activation/park (WRITE_P)", 3: "neither", 4: "no VMEM in the loop",
              16: "everything but the MFMAs", 15: "MFMA only", 27: "VMEM only", 11: "VMEM + MFMA only",
              128: "two-phase loop (first form)", 132: "two-phase loop, no VMEM"}
    cases = [(96, 96, 64), (192, 192, 32), (480, 192, 32), (576, 288, 16)]
    if os.environ.get("MCVD_WEXP_CASES", "all") != "all":
        cases = [cases[int(v)] for v in os.environ["MCVD_WEXP_CASES"].split(",")]
    if os.environ.get("MCVD_WEXP_ONLY"):
        keep = [int(v) for v in os.environ["MCVD_WEXP_ONLY"].split(",")]
        labels = {k: v for k, v in labels.items() if k in keep}
    if kshape == 10:
        f.write("# bf16x3 winograd K-loop, B=64, conv_wino3_kernel<3,2>; ideal MFMA time per chunk per SIMD = 2 waves x 36 x 32 = 2304 cycles\n")
    else:
        f.write("# f16x2 winograd K-loop, B=64, conv_wino2h_kernel<3,2>; ideal MFMA time per chunk per SIMD = 2 waves x 18 x 32 = 1152 cycles\n")

    def run(shape, e):
        ctx.opt("conv_shape", shape)
        os.environ[envname] = str(e)
        for _ in range(2):
            ctx.conv2d(x, w, b, coef=coef, act=1, scale=0.7)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(4):
            ctx.conv2d(x, w, b, coef=coef, act=1, scale=0.7)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / 4
        per = []
        for wv in (0, 7):
            os.environ["MCVD_DBG_WAVE"] = str(wv)
            dbg = torch.zeros(65536 * 8, dtype=torch.int64, device="cuda")
            _lib.check(_lib.lib.mcvd_ctx_set_debug_buffer(ctx.h, P(dbg)))
            ctx.conv2d(x, w, b, coef=coef, act=1, scale=0.7)
            torch.cuda.synchronize()
            _lib.check(_lib.lib.mcvd_ctx_set_debug_buffer(ctx.h, None))
            d = dbg.view(-1, 8).cpu().double()
            d = d[d[:, 7] > 0]
            m = d.mean(0)
            per.append((m[0].item(), m[1].item() / max(m[6].item(), 1), m[5].item(), m[7].item()))
        return us, per

    for cin, cout, H in cases:
        x = torch.randn(B, cin, H, H, device="cuda")
        w = torch.randn(cout, cin, 3, 3, device="cuda") / (cin * 9) ** 0.5
        b = torch.zeros(cout, device="cuda")
        coef = torch.ones(B, cin, 2, device="cuda")
        os.environ["MCVD_DBG_WAVE"] = "0"
        us4, _ = run(4, 0)
        f.write(f"cin{cin} cout{cout} H{H} fp32-MFMA winograd (shape 4): kernel {us4:7.1f} us\n")
        for e, lab in labels.items():
            us, per = run(kshape, e)
            f.write(f"cin{cin} cout{cout} H{H} exp{e:4d} {lab:36s}: kernel {us:7.1f} us | wave0 pro {per[0][0]:6.0f} loop/chunk {per[0][1]:6.0f} epi {per[0][2]:6.0f} total {per[0][3]:7.0f}"
                    f" | wave7 pro {per[1][0]:6.0f} loop/chunk {per[1][1]:6.0f} epi {per[1][2]:6.0f} total {per[1][3]:7.0f}\n")
            f.flush()
    os.environ[envname] = "0"
    ctx.opt("conv_shape", -1)


def w3sub(f):
    """conv_wino3.cpp (diagnostics library): cycles per chunk the recording wave spends in each sub-phase of the K loop
    (MCVD_DBG_WAVE = 128 + wave)."""
    from tests.hiputil import Ctx, P
    ctx = Ctx()
    B = 64
    names = ["C/OFF reads + patch wait", "activation + park", "patch-load issue", "transform + split + stores", "pos 0: B reads + weight wait + MFMAs + reload",
             "pos 1: same", "barrier"]
    for cin, cout, H in [(96, 96, 64), (480, 192, 32)]:
        x = torch.randn(B, cin, H, H, device="cuda")
        w = torch.randn(cout, cin, 3, 3, device="cuda") / (cin * 9) ** 0.5
        b = torch.zeros(cout, device="cuda")
        coef = torch.ones(B, cin, 2, device="cuda")
        ctx.opt("conv_shape", 10)
        for wv in (0, 3, 4, 7):
            os.environ["MCVD_DBG_WAVE"] = str(128 + wv)
            dbg = torch.zeros(65536 * 8, dtype=torch.int64, device="cuda")
            ctx.conv2d(x, w, b, coef=coef, act=1, scale=0.7)
            _lib.check(_lib.lib.mcvd_ctx_set_debug_buffer(ctx.h, P(dbg)))
            ctx.conv2d(x, w, b, coef=coef, act=1, scale=0.7)
            torch.cuda.synchronize()
            _lib.check(_lib.lib.mcvd_ctx_set_debug_buffer(ctx.h, None))
            d = dbg.view(-1, 8).cpu().double()
            d = d[d[:, 6] > 0]
            n = (d[:, 6] - 1).clamp(min=1)                # chunks the loop ran
            vals = [(d[:, i] / n).mean().item() for i in (0, 1, 2, 3, 4, 5, 7)]
            f.write(f"cin{cin} cout{cout} H{H} wave {wv}: " + " | ".join(f"{nm} {v:6.0f}" for nm, v in zip(names, vals)) + f" | sum {sum(vals):6.0f}\n")
    os.environ["MCVD_DBG_WAVE"] = "0"
    ctx.opt("conv_shape", -1)


def w3pro(f):
    """conv_wino3.cpp: where the prologue of a workgroup goes (MCVD_DBG_WAVE = 64 + wave: cycles from the start of the kernel at which
    the index arithmetic is done, the loads are issued, coefficients + first patches have landed, the coefficient table is visible,
    the first two patches are parked; then the whole prologue)."""
    from tests.hiputil import Ctx, P
    ctx = Ctx()
    B = 64
    for cin, cout, H in [(96, 96, 64), (192, 192, 32), (480, 192, 32), (384, 384, 8)]:
        x = torch.randn(B, cin, H, H, device="cuda")
        w = torch.randn(cout, cin, 3, 3, device="cuda") / (cin * 9) ** 0.5
        b = torch.zeros(cout, device="cuda")
        coef = torch.ones(B, cin, 2, device="cuda")
        ctx.opt("conv_shape", 10)
        for wv in (0, 7):
            os.environ["MCVD_DBG_WAVE"] = str(64 + wv)
            dbg = torch.zeros(65536 * 8, dtype=torch.int64, device="cuda")
            ctx.conv2d(x, w, b, coef=coef, act=1, scale=0.7)
            _lib.check(_lib.lib.mcvd_ctx_set_debug_buffer(ctx.h, P(dbg)))
            ctx.conv2d(x, w, b, coef=coef, act=1, scale=0.7)
            torch.cuda.synchronize()
            _lib.check(_lib.lib.mcvd_ctx_set_debug_buffer(ctx.h, None))
            d = dbg.view(-1, 8).cpu().double()
            d = d[d[:, 6] > 0]
            m = d.mean(0)
            f.write(f"cin{cin} cout{cout} H{H} wave {wv}: index math done {m[1]:6.0f} | loads issued {m[2]:6.0f} | landed {m[3]:6.0f} | coefficient table visible {m[4]:6.0f}"
                    f" | patches parked {m[5]:6.0f} | prologue {m[0]:6.0f} | total {m[7]:7.0f} ({int(m[6])} chunks)\n")
    os.environ["MCVD_DBG_WAVE"] = "0"
    ctx.opt("conv_shape", -1)


def hostloop(f):
    """What `verbose=True` / `log=True` cost: those kwargs (the reference's video_gen passes them, runners/ncsn_runner.py:1516-1519) take
    the host loop of samplers.py (one forward + one fused update per step driven from Python, the ten log lines computed with torch)
    instead of the device loop mcvd_sampler_run.  Headline workload (config 2, B = 64, 100 steps + denoise), 2 calls each after a warm-up."""
    import io
    import logging
    from contextlib import redirect_stdout
    config, sd, net = mk("smmnist_big5_ngf96")
    B = 64
    x, cond = synthetic.random_inputs(config, 0, B)
    x, cond = x.cuda(), cond.cuda()
    net.set_option("graph", 1)
    logging.getLogger().setLevel(logging.ERROR)

    def run(verbose, log, final_only=True):
        with redirect_stdout(io.StringIO()):
            return ddpm_sampler(x, net, cond=cond, final_only=final_only, denoise=True, subsample_steps=100, clip_before=True,
                                verbose=verbose, log=log, seed=7)
    res = {}
    for name, kw in (("device loop (verbose=False, log=False)", dict(verbose=False, log=False)),
                     ("host loop (verbose=True, log=True)", dict(verbose=True, log=True))):
        run(**kw)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(2):
            run(**kw)
        torch.cuda.synchronize()
        res[name] = (time.perf_counter() - t0) / 2
        f.write(f"{name}: {res[name] * 1e3:.1f} ms per sampler call, {B * 5 / res[name]:.1f} frames/s\n")
    a, b = list(res.values())
    f.write(f"host-loop overhead: {(b - a) * 1e3:.1f} ms per call = {(b - a) / 101 * 1e3:.3f} ms per step ({100 * (b - a) / a:.2f} %)\n")


def w2htl(f):
    """Per-CU timeline of conv_wino2h_kernel launches: every workgroup records its start / end on the 100 MHz wall clock, its
    shader-cycle count and the CU it ran on.  Answers: what is the shader clock under this kernel, and how long does a CU sit
    between two workgroups (dispatch gap)."""
    from tests.hiputil import Ctx, P
    ctx = Ctx()
    B = 64
    cases = [(96, 96, 64, 3, 12, 0), (480, 192, 32, 3, 12, 0), (192, 576, 32, 1, 14, 2), (192, 576, 32, 1, 14, 3), (288, 96, 64, 1, 14, 3), (384, 1152, 8, 1, 14, 3)]
    if os.environ.get("MCVD_TL_B3", "1") != "0":          # the three-piece bf16 kernels (default) instead of the two-piece fp16 ones
        cases = [(a, b_, c, d, {12: 10, 14: 15}[e], g) for a, b_, c, d, e, g in cases]
    if os.environ.get("MCVD_TL_CASES"):
        cases = [cases[int(v)] for v in os.environ["MCVD_TL_CASES"].split(",")]
    act_tl = int(os.environ.get("MCVD_TL_ACT", "1"))        # 0: affine prologue only (what the q|k|v projections run)
    for cin, cout, H, ks, shp, cot in cases:
        x = torch.randn(B, cin, H, H, device="cuda")
        w = torch.randn(cout, cin, ks, ks, device="cuda") / (cin * ks * ks) ** 0.5
        b = torch.zeros(cout, device="cuda")
        coef = torch.ones(B, cin, 2, device="cuda")
        ctx.opt("conv_shape", shp)
        ctx.opt("conv_cot", cot)
        f.write(f"--- {ks}x{ks} shape {shp} cot {cot} exp {os.environ.get('MCVD_Q1_EXP', '0')}: ")
        os.environ["MCVD_DBG_WAVE"] = "0"
        for _ in range(3):
            ctx.conv2d(x, w, b, coef=coef, act=act_tl, scale=0.7)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(4):
            ctx.conv2d(x, w, b, coef=coef, act=act_tl, scale=0.7)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / 4
        dbg = torch.zeros(65536 * 8, dtype=torch.int64, device="cuda")
        _lib.check(_lib.lib.mcvd_ctx_set_debug_buffer(ctx.h, P(dbg)))
        ctx.conv2d(x, w, b, coef=coef, act=act_tl, scale=0.7)
        torch.cuda.synchronize()
        _lib.check(_lib.lib.mcvd_ctx_set_debug_buffer(ctx.h, None))
        d = dbg.view(-1, 8).cpu()
        d = d[d[:, 7] > 0]
        rt0, rt1, cyc, who = d[:, 2].double(), d[:, 3].double(), d[:, 7].double(), d[:, 4]
        span = (rt1.max() - rt0.min()).item() * 10e-3          # us (100 MHz ticks)
        clk = (cyc / ((rt1 - rt0) * 10e-9)).median().item() / 1e9
        f.write(f"cin{cin} cout{cout} H{H}: kernel {us:.1f} us (events), first start -> last end {span:.1f} us, {len(d)} workgroups, "
                f"shader clock (cycles / wall time inside a workgroup, median) {clk:.3f} GHz, workgroup {cyc.mean().item():.0f} cycles = {((rt1 - rt0).mean().item() * 10e-3):.2f} us"
                f" (prologue {d[:, 0].double().mean().item():.0f}, loop {d[:, 1].double().mean().item():.0f} = {(d[:, 1].double() / d[:, 6].double().clamp(min=1)).mean().item():.0f} per chunk, epilogue {d[:, 5].double().mean().item():.0f})\n")
        # per-CU timelines
        cus = {}
        for i in range(len(d)):
            cus.setdefault(((int(who[i].item()) >> 32) & 0xf, (int(who[i].item()) >> 8) & 0xff), []).append((rt0[i].item(), rt1[i].item()))   # key: (XCC, SE | SH | CU) of HW_ID
        gaps, busy, nper = [], [], []
        for k, v in cus.items():
            v.sort()
            nper.append(len(v))
            busy.append(sum(b - a for a, b in v))
            gaps += [v[j + 1][0] - v[j][1] for j in range(len(v) - 1)]
        g = torch.tensor(gaps) * 10e-3 if gaps else torch.zeros(1)
        f.write(f"   {len(cus)} CUs, workgroups per CU {min(nper)}..{max(nper)}, busy per CU {min(busy) * 10e-3:.1f}..{max(busy) * 10e-3:.1f} us, "
                f"gap between consecutive workgroups on a CU: median {g.median().item():.2f} us, mean {g.mean().item():.2f} us, max {g.max().item():.2f} us; "
                f"first workgroup starts {((torch.tensor([v[0][0] for v in cus.values()]) - rt0.min().item()) * 10e-3).max().item():.2f} us after the earliest\n")
    ctx.opt("conv_shape", -1)
    ctx.opt("conv_cot", 0)


def w2hsub(f):
    """Prologue sub-phases of conv_wino2h_kernel (cycles from the workgroup's first stamp, averaged over the workgroups): loads
    issued, loads landed, first two patches activated + parked (barrier passed), whole prologue (first V tile visible)."""
    from tests.hiputil import Ctx, P
    ctx = Ctx()
    B = 64
    for cin, cout, H in [(96, 96, 64), (192, 192, 32), (480, 192, 32)]:
        x = torch.randn(B, cin, H, H, device="cuda")
        w = torch.randn(cout, cin, 3, 3, device="cuda") / (cin * 9) ** 0.5
        b = torch.zeros(cout, device="cuda")
        coef = torch.ones(B, cin, 2, device="cuda")
        ctx.opt("conv_shape", 12)
        for _ in range(3):
            ctx.conv2d(x, w, b, coef=coef, act=1, scale=0.7)
        for wv in (0, 5):
            os.environ["MCVD_DBG_WAVE"] = str(64 + wv)
            dbg = torch.zeros(65536 * 8, dtype=torch.int64, device="cuda")
            _lib.check(_lib.lib.mcvd_ctx_set_debug_buffer(ctx.h, P(dbg)))
            ctx.conv2d(x, w, b, coef=coef, act=1, scale=0.7)
            torch.cuda.synchronize()
            _lib.check(_lib.lib.mcvd_ctx_set_debug_buffer(ctx.h, None))
            d = dbg.view(-1, 8).cpu().double()
            d = d[d[:, 7] > 0].mean(0)
            f.write(f"cin{cin} cout{cout} H{H} wave {wv}: loads issued {d[2].item():6.0f}  landed {d[3].item():6.0f}  patches parked {d[4].item():6.0f}  prologue {d[0].item():6.0f}"
                    f" | loop {d[1].item():7.0f} ({d[1].item() / max(d[6].item() - 1, 1):5.0f} per chunk)  epilogue {d[5].item():6.0f}  total {d[7].item():7.0f}\n")
    os.environ["MCVD_DBG_WAVE"] = "0"
    ctx.opt("conv_shape", -1)


def w3ptl(f):
    """conv_wino3p_kernel (persistent workgroups, shape 16 / 17) next to conv_wino3_kernel (10 / 11) on the bench's layer shapes at B = 64
    (env B): event time of both, and for the persistent kernel the cycles a workgroup (wave MCVD_DBG_WAVE of it; product library: wave 0)
    spends in prologues / K loops / epilogues, items per workgroup, shader clock."""
    from tests.hiputil import Ctx, P
    ctx = Ctx()
    B = int(os.environ.get("B", 64))
    cases = [(96, 96, 64), (192, 192, 64), (288, 96, 64), (192, 96, 64), (96, 192, 32), (192, 192, 32), (480, 192, 32), (288, 288, 32),
             (192, 288, 16), (288, 288, 16), (672, 288, 16), (384, 384, 16), (192, 192, 16)]
    if os.environ.get("MCVD_TL_CASES"):
        cases = [cases[int(v)] for v in os.environ["MCVD_TL_CASES"].split(",")]
    for cin, cout, H in cases:
        x = torch.randn(B, cin, H, H, device="cuda")
        w = torch.randn(cout, cin, 3, 3, device="cuda") / (cin * 9) ** 0.5
        b = torch.zeros(cout, device="cuda")
        coef = torch.ones(B, cin, 2, device="cuda")
        res = torch.randn(B, cout, H, H, device="cuda")
        line = f"cin{cin:4d} cout{cout:4d} H{H:3d}:"
        ref = None
        for shp in (10, 16, 11, 17):
            tag = shp
            ctx.opt("conv_shape", shp)
            for _ in range(3):
                y = ctx.conv2d(x, w, b, coef=coef, act=1, res=res, scale=0.7)
            ran = _lib.lib.mcvd_last_conv_kernel()
            if ran != shp:
                line += f"  [{tag}: n/a]"
                continue
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(6):
                ctx.conv2d(x, w, b, coef=coef, act=1, res=res, scale=0.7)
            e1.record()
            torch.cuda.synchronize()
            line += f"  [{tag}: {e0.elapsed_time(e1) * 1e3 / 6:7.1f} us]"
            if shp in (10, 11):
                ref = y.clone()
            elif ref is not None:
                line += " bit-equal" if torch.equal(y, ref) else " DIFFERENT"
            if shp in (16, 17):
                dbg = torch.zeros(4096 * 8, dtype=torch.int64, device="cuda")
                _lib.check(_lib.lib.mcvd_ctx_set_debug_buffer(ctx.h, P(dbg)))
                ctx.conv2d(x, w, b, coef=coef, act=1, res=res, scale=0.7)
                torch.cuda.synchronize()
                _lib.check(_lib.lib.mcvd_ctx_set_debug_buffer(ctx.h, None))
                d = dbg.view(-1, 8).cpu()
                d = d[d[:, 7] > 0]
                rt0, rt1, cyc = d[:, 2].double(), d[:, 3].double(), d[:, 7].double()
                items = (d[:, 6] >> 32).double()
                chunks = (d[:, 6] & 0xffffffff).double()
                clk = (cyc / ((rt1 - rt0) * 10e-9)).median().item() / 1e9
                span = (rt1.max() - rt0.min()).item() * 10e-3
                line += (f" {{{len(d)} wg, items {int(items.min())}..{int(items.max())}, {clk:.2f} GHz, span {span:.1f} us, per wg: total {cyc.mean():.0f} cyc = prologue {d[:, 0].double().mean():.0f}"
                         f" + loops {d[:, 1].double().mean():.0f} ({(d[:, 1].double() / chunks).mean():.0f}/chunk) + epilogues {d[:, 5].double().mean():.0f} ({(d[:, 5].double() / items).mean():.0f}/item)}}")
        f.write(line + "\n")
        f.flush()
    ctx.opt("conv_shape", -1)


def convops(f):
    """Per-op times of the 3x3 convs of one instrumented forward (BASELINE config 2, B = 64): kernel the autotuner chose and ms,
    with the split-operand bf16 Winograd kernel offered (MCVD_BF16X3 unset) -- run again with MCVD_BF16X3=0 for the fp32 table."""
    import ctypes as C
    from oracle import synth
    from mcvd_pytorch_amd.scorenet import HipScoreNet
    B = int(os.environ.get("B", 64))
    config = synth.make_config(os.environ.get("CFG", "smmnist_big5_ngf96")); config.device = "cuda:0"
    sd = synth.make_state_dict(config, seed=123)
    net = HipScoreNet(config); net.load_state_dict({"module." + k: v for k, v in sd.items()}, strict=False); net.eval()
    for kv in os.environ.get("OPTS", "").split(","):                 # e.g. OPTS=f16x2=1 or OPTS=conv_shape=10,conv_shape1=15
        if kv:
            net.set_option(kv.split("=")[0], int(kv.split("=")[1]))
    x, cond = synth.make_inputs(config, B, seed=0)
    x, cond = x.cuda(), cond.cuda()
    t = torch.full((B,), 500, dtype=torch.long, device="cuda")
    for _ in range(3):
        net(x, t, cond=cond)
    net.set_option("profile", 1)
    from mcvd_pytorch_amd.samplers import ddpm_sampler
    ddpm_sampler(x, net, cond=cond, denoise=True, subsample_steps=2, clip_before=True, verbose=False, log=False, config=config, final_only=True)
    torch.cuda.synchronize()
    n = _lib.lib.mcvd_model_profile_read(net._model, None, None, None, None, None, 0)
    kinds, kss = (C.c_int * n)(), (C.c_int * n)()
    ms, fl, by = (C.c_double * n)(), (C.c_double * n)(), (C.c_double * n)()
    _lib.lib.mcvd_model_profile_read(net._model, kinds, kss, ms, fl, by, n)
    info = (C.c_int * 8)()
    tot = {}
    for i in range(n):
        if kinds[i] != 3 or ms[i] == 0.0:
            continue
        _lib.lib.mcvd_model_op_info(net._model, i, info)
        shape = _lib.lib.mcvd_model_op_kernel(net._model, i)          # the kernel family that really ran
        key = (kss[i], shape)
        tot[key] = tot.get(key, 0.0) + ms[i]
        f.write(f"op {i:3d} {kss[i]}x{kss[i]} H{info[3]:3d} cin{info[4]:4d} cout{info[5]:4d} shape {shape:2d} cot {(info[6] >> 8) & 15}: {ms[i] * 1e3:7.1f} us  {fl[i] / ms[i] / 1e9:6.1f} TF/s  {by[i] / ms[i] / 1e6:7.1f} GB/s\n")
    f.write("totals (ks, shape) -> ms: " + ", ".join(f"{k}: {v:.3f}" for k, v in sorted(tot.items())) + "\n")


if __name__ == "__main__":
    what = sys.argv[1:] or ["precision", "ops", "sweep"]
    for w in what:
        with open(os.path.join(OUT, f"diag_{w}.txt"), "w") as f:
            t0 = time.time()
            {"precision": precision, "ops": ops, "sweep": sweep, "phases": phases, "wphases": wphases, "wexp": wexp, "w3exp": w3exp, "w3sub": w3sub, "w3pro": w3pro, "hostloop": hostloop, "convops": convops, "w3ptl": w3ptl, "sweep1": sweep1, "w2htl": w2htl, "w2hsub": w2hsub}[w](f)
            f.write(f"# done in {time.time() - t0:.1f}s\n")


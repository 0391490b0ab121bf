#!/usr/bin/env python
"""Headline benchmark: sampled frames/s (whole job) of the DDPM sampling hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W [--config NAME]

A "step" is ONE full sampling job over one batch per GPU.  Default workload = BASELINE.json configs[1]: `ddpm_sampler` with
subsample=100 (100 UNet forwards + fused updates + the denoise forward = 101 forwards) on smmnist_DDPM_big5 with ngf=96, 64x64,
5 cond + 5 predicted frames, batch 64 per GPU, synthetic random weights / inputs, on-device Philox noise.  Frames = B * 5 per
step per GPU.  For the autoregressive config (`--config cityscapes_big`: num_frames_pred=28 > num_frames=5) a step is the whole
`video_gen` block loop (6 sampler calls, cond shift on the device) and frames = B * 28 KEPT frames.
`value` = frames of all ranks / max-over-ranks wall time (weak scaling: per-GPU batch fixed).  Inputs are resident in HBM before
the timed region.  One JSON line on stdout (rank 0).

`--gpus N` with N > 1 and no launcher (WORLD_SIZE unset) re-executes itself under `python -m torch.distributed.run` with one
rank per GPU (RCCL); under the driver's own torchrun launch it just reads RANK / LOCAL_RANK / WORLD_SIZE.
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

FP32_MFMA_PEAK_TFLOPS = 157.3      # /opt/skills/guides/MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 dense peak
BF16_MFMA_PEAK_TFLOPS = 2500.0     # same guide: v_mfma_f32_32x32x16_bf16 / _f16 dense peak (no sparsity)
HBM_PEAK_GBS = 8000.0

# per-config defaults: (per-GPU batch, subsample, metric label) -- BASELINE.json configs[1..4] at their per-GPU batch
DEFAULTS = {
    "smmnist_big5_ngf96": (64, 100, "SMMNIST 64x64 DDPM 100-step"),
    "smmnist_big5": (64, 100, "SMMNIST 64x64 (ngf=64) DDPM 100-step"),
    "kth64_big_ngf128": (32, 100, "KTH 64x64 (ngf=128) DDPM 100-step"),
    "bair_big_spade": (16, 1000, "BAIR 64x64 SPADE DDPM 1000-step"),
    "cityscapes_big": (8, 100, "Cityscapes 128x128 autoregressive 28 frames, DDPM 100-step per block"),
    "cityscapes_big_variant": (8, 100, "Cityscapes 128x128 ch_mult=[1,2,3,4,4] attn@16, autoregressive 28 frames"),
}


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def plan_launch(gpus, env, argv=None):
    """None: run in this process (N == 1, or already one rank of a launcher).  Otherwise the command that starts `gpus` ranks of
    this script on this node, one per GPU, over RCCL.  A launcher whose WORLD_SIZE disagrees with --gpus is refused (a run that
    silently benchmarks another N reads as 'no scaling')."""
    world = env.get("WORLD_SIZE")
    if world is not None:
        if int(world) != gpus:
            raise SystemExit(f"bench.py: --gpus {gpus} but the launcher set WORLD_SIZE={world}; refusing to run an ambiguous job")
        return None
    if gpus <= 1:
        return None
    port = env.get("MASTER_PORT", str(29500 + os.getpid() % 2000))
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(gpus), "--master-addr", "127.0.0.1",
            "--master-port", port, os.path.join(ROOT, "bench.py")] + list(argv if argv is not None else sys.argv[1:])


def make_config(name):
    """BASELINE.json configs as reference-schema namespaces (hot-path keys; SURVEY section 0 override table)."""
    from mcvd_pytorch_amd import dict2namespace
    data = dict(image_size=64, channels=1, num_frames=5, num_frames_cond=5, num_frames_future=0)
    model = dict(version="DDPM", arch="unetmore", type="v1", time_conditional=True, dropout=0.1, sigma_dist="linear",
                 sigma_begin=0.02, sigma_end=0.0001, num_classes=1000, ngf=96, ch_mult=[1, 2, 3, 4], num_res_blocks=2,
                 attn_resolutions=[8, 16, 32], n_head_channels=96, spade=False, spade_dim=128)
    sampling = dict(subsample=100, denoise=True, clip_before=True, num_frames_pred=5, init_prev_t=-1.0)
    if name == "smmnist_big5_ngf96":
        pass
    elif name == "smmnist_big5":
        model.update(ngf=64, n_head_channels=64)
    elif name == "kth64_big_ngf128":
        data.update(num_frames_cond=10)
        model.update(ngf=128, n_head_channels=128)
    elif name == "bair_big_spade":
        data.update(channels=3, num_frames_cond=2)
        model.update(spade=True, spade_dim=128)
    elif name == "cityscapes_big":
        data.update(image_size=128, channels=3, num_frames_cond=2)
        model.update(ngf=128, n_head_channels=128, ch_mult=[1, 1, 2, 3, 4])
        sampling.update(num_frames_pred=28)
    elif name == "cityscapes_big_variant":
        data.update(image_size=128, channels=3, num_frames_cond=2)
        model.update(ngf=128, n_head_channels=128, ch_mult=[1, 2, 3, 4, 4], attn_resolutions=[16])
        sampling.update(num_frames_pred=28)
    else:
        raise KeyError(name)
    return dict2namespace(dict(data=data, model=model, sampling=sampling))


def cpu_baseline(config, state_dict, subsample, budget_s=24.0, kept_fraction=1.0):
    """The CPU port of the reference sampler (oracle/, torch CPU fp32) timed on the GPU box's host cores on a bounded sample of
    the same workload: B=8 rows of the same synthetic inputs.  First a thread-count sweep (1 warm-up + 2 timed forwards at 16 /
    32 / 64 / 128 threads, capped at the box's count), then the sampler itself at the best count for the rest of the budget (each
    step = one UNet forward + update), extrapolated linearly to the subsample + 1 forwards of a full call.
    kind="port": the Python reference cannot travel to the GPU box."""
    import torch
    from oracle import sampler_ref, unet_ref
    from mcvd_pytorch_amd import synthetic
    ncpu = os.cpu_count() or 1
    B = 8
    net = unet_ref.OracleScoreNet(config, {k: v.float().cpu() for k, v in state_dict.items()})
    x, cond = synthetic.random_inputs(config, 0, B)
    t = torch.full((B,), 500).long()
    sweep = {}
    t_begin = time.perf_counter()
    for th in [c for c in (16, 32, 64, 128) if c <= ncpu] or [ncpu]:
        torch.set_num_threads(th)
        with torch.no_grad():
            net(x, t, cond=cond)                                    # warm-up at this thread count
            t0 = time.perf_counter()
            for _ in range(2):
                net(x, t, cond=cond)
        sweep[th] = (time.perf_counter() - t0) / 2
        if time.perf_counter() - t_begin > 0.6 * budget_s:
            break
    cores = min(sweep, key=sweep.get)
    torch.set_num_threads(cores)
    n_steps, t0 = 0, time.perf_counter()
    remaining = max(budget_s - (t0 - t_begin), 3.0)

    class Stop(Exception):
        pass

    def counting(xx, yy, cond=None):
        nonlocal n_steps
        if time.perf_counter() - t0 > remaining and n_steps >= 3:
            raise Stop()
        n_steps += 1
        return net(xx, yy, cond=cond)
    counting.alphas, counting.alphas_prev, counting.betas = net.alphas, net.alphas_prev, net.betas
    try:
        sampler_ref.sample(x, counting, cond=cond, kind="ddpm", final_only=True, denoise=True, subsample_steps=subsample)
    except Stop:
        pass
    dt = time.perf_counter() - t0
    per_fwd = dt / max(n_steps, 1)
    fps = kept_fraction * B * config.data.num_frames / (per_fwd * (subsample + 1))      # autoregressive: kept / generated frames
    pvr = {}
    try:      # the ratio port / REAL reference, measured where both exist (the build container; tools/cpu_port_vs_reference.py)
        pj = json.load(open(os.path.join(ROOT, "profiles", "r05_cpu_port_vs_reference.json")))
        pvr = dict(port_vs_reference=pj["port_vs_reference"],
                   port_vs_reference_source="profiles/r05_cpu_port_vs_reference.json (build container, %d threads, %s B=%d: the Python reference "
                                            "does not travel to the GPU box)" % (pj["threads"], pj["config"], pj["batch"]),
                   reference_estimate=round(fps / pj["port_vs_reference"], 4),
                   reference_estimate_note="value / port_vs_reference: the reference's own frames/s on this box's cores, estimated")
    except Exception:
        pass
    return dict(value=round(fps, 4), unit="frames/s", cores=cores, kind="port", host_threads_available=ncpu, **pvr,
                thread_sweep_s_per_forward={str(k): round(v, 4) for k, v in sweep.items()},
                sample=f"oracle ddpm_sampler, B={B}, {n_steps} of {subsample + 1} forwards timed ({dt:.1f}s) at the best of the "
                       f"swept thread counts ({cores} of {ncpu} hardware threads), extrapolated linearly; torch {torch.__version__} CPU")


def roofline_of_leg(net, args, B, arith_name):
    """Per-op HIP-event timings of ONE forward inside the leg's timed region (the first forward of its last sampler call runs op by op
    with events around each launch, so the sum is an UPPER bound of a back-to-back forward: event gaps) -> the roofline block of the
    leg's dominant kernel family + the arithmetic (`dtype`) string of what really ran (mcvd_model_op_kernel)."""
    import ctypes as C
    from mcvd_pytorch_amd import _lib
    n = _lib.lib.mcvd_model_profile_read(net._model, None, None, None, None, None, 0)
    kinds, kss = (C.c_int * n)(), (C.c_int * n)()
    ms, fl, by = (C.c_double * n)(), (C.c_double * n)(), (C.c_double * n)()
    _lib.check(0 if _lib.lib.mcvd_model_profile_read(net._model, kinds, kss, ms, fl, by, n) == n else -1, "profile_read")
    names = {0: "temb_mlp", 1: "dense_all", 2: "gn_coef", 3: "conv", 4: "fir2", 5: "attention", 6: "nearest", 7: "coef2", 8: "spade_apply"}
    agg = {}
    for i in range(n):
        if ms[i] == 0.0:          # cond-only (SPADE prep) ops are not part of the per-step forward
            continue
        key = names.get(kinds[i], f"op{kinds[i]}") + (f"{kss[i]}x{kss[i]}" if kinds[i] == 3 else "")
        a = agg.setdefault(key, dict(launches=0, ms=0.0, flops=0.0, bytes=0.0))
        a["launches"] += 1; a["ms"] += ms[i]; a["flops"] += fl[i]; a["bytes"] += by[i]
    fwd_ms = sum(a["ms"] for a in agg.values())
    breakdown = {k: dict(launches=a["launches"], ms=round(a["ms"], 3), share=round(a["ms"] / fwd_ms, 4),
                         tflops=round(a["flops"] / a["ms"] / 1e9, 2), gbs=round(a["bytes"] / a["ms"] / 1e6, 1))
                 for k, a in sorted(agg.items(), key=lambda kv: -kv[1]["ms"])}
    # ---- dominant kernel: the 3x3 convs, by the kernel family each layer REALLY ran
    FAM3 = {4: "wino_f32", 8: "wino_f32", 10: "wino_bf16x3", 11: "wino_bf16x3", 12: "wino_f16x2", 13: "wino_f16x2",
            16: "wino_bf16x3", 17: "wino_bf16x3", 18: "wino_bf16x3", 19: "wino_bf16x3", 20: "wino_bf16x3",          # 16 / 17: the same kernel as persistent workgroups (conv_wino3p.cpp)
            22: "gemm_bf16x3", 23: "gemm_bf16x3"}                  # the stem / last conv as a 1x1 GEMM on the three-piece kernel (conv_gemm_forms.cpp)
    fam = {k: dict(launches=0, ms=0.0, flops=0.0, bytes=0.0) for k in ("wino_f32", "wino_bf16x3", "wino_f16x2", "gemm_bf16x3", "direct")}
    k1 = {}
    for i in range(n):
        if kinds[i] != 3 or ms[i] == 0.0:
            continue
        ran = _lib.lib.mcvd_model_op_kernel(net._model, i)
        if kss[i] == 1:
            k1[ran] = k1.get(ran, 0) + 1
            continue
        f = fam[FAM3.get(ran, "direct")]                      # 8 / 11 / 13: + the K-split reduce pass
        f["launches"] += 1; f["ms"] += ms[i]; f["flops"] += fl[i]; f["bytes"] += by[i]
    c3 = agg["conv3x3"]
    dom_key = max(fam, key=lambda k: fam[k]["ms"])
    dom = fam[dom_key]
    # (kernel name, matrix-pipe flops executed per direct-form flop, peak of the pipe it runs on)
    #   Winograd F(2x2,3x3) executes 16 of the direct form's 36 multiplies per output tile; the three-piece bf16 kernel issues SIX
    #   piece products per fp32 product, the two-piece fp16 kernel THREE
    FAMILY = {
        "wino_f32": ("conv_wino_kernel (3x3 conv, Winograd F(2x2,3x3) on v_mfma_f32_32x32x2_f32)", 16.0 / 36.0, FP32_MFMA_PEAK_TFLOPS),
        "wino_bf16x3": ("conv_wino3_kernel / conv_wino3p_kernel (persistent workgroups) (3x3 conv, Winograd F(2x2,3x3), operands split exactly into 3 bf16 pieces, weights pre-split at "
                        "pack time, 6 piece products on v_mfma_f32_32x32x16_bf16, fp32 accumulate: fp32-equivalent)", 6.0 * 16.0 / 36.0,
                        BF16_MFMA_PEAK_TFLOPS),
        "wino_f16x2": ("conv_wino2h_kernel (3x3 conv, Winograd F(2x2,3x3), operands split into 2 fp16 pieces = 22 significant bits, weights "
                       "pre-split at pack time, 3 piece products on v_mfma_f32_32x32x16_f16, fp32 accumulate)", 3.0 * 16.0 / 36.0, BF16_MFMA_PEAK_TFLOPS),
        "gemm_bf16x3": ("conv1x1_h2_kernel behind an im2col / in front of a shift-and-add pass (3x3 conv with few channels on one side as a 1x1 "
                        "GEMM, three bf16 pieces per operand, 6 piece products)", 6.0, BF16_MFMA_PEAK_TFLOPS),
        "direct": ("conv_mfma_kernel<3x3> (direct implicit GEMM, v_mfma_f32_32x32x2_f32)", 1.0, FP32_MFMA_PEAK_TFLOPS),
    }
    dom_name, mult_ratio, pipe_peak = FAMILY[dom_key]
    algorithmic = dom["flops"] / dom["ms"] / 1e9
    executed = algorithmic * mult_ratio
    # HBM-side bytes per launch: NOT measured in this run -- read from the committed PMC passes (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE,
    # tools/gpu_check.sh prof) over the same launch population of the same kernel family, when such a file exists
    traffic, traffic_src, traffic_sha = None, None, None
    try:
        if args.config == "smmnist_big5_ngf96" and B == 64:
            for fn in sorted(f for f in os.listdir(os.path.join(ROOT, "profiles")) if f.endswith("_traffic.json")):
                tj = json.load(open(os.path.join(ROOT, "profiles", fn)))
                if tj.get("family", "wino_f32") == dom_key and "traffic_bytes_per_launch" in tj:
                    import hashlib
                    traffic, traffic_src = round(tj["traffic_bytes_per_launch"]), "profiles/" + fn      # the latest file of the family wins
                    traffic_sha = hashlib.sha256(open(os.path.join(ROOT, "profiles", fn), "rb").read()).hexdigest()[:16]
    except Exception:
        traffic = None
    roofline = dict(bound="mfma", kernel=dom_name,
                    achieved=round(executed, 2), peak=pipe_peak, unit="TFLOP/s",
                    frac=round(executed / pipe_peak, 4), traffic=traffic, traffic_from_file=bool(traffic), traffic_source=traffic_src,
                    traffic_file_sha16=traffic_sha,
                    note=("achieved / frac = flops the kernel EXECUTES on its matrix pipe (Winograd F(2x2,3x3): 16/36 of the "
                          "direct-form multiplies; x6 piece products for the three-piece bf16 kernel, x3 for the two-piece fp16 kernel) / "
                          "HIP-event time of its launches in one forward of the timed region / that pipe's dense peak, i.e. the matrix-pipe "
                          "utilisation (agrees with PMC SQ_VALU_MFMA_BUSY_CYCLES, profiles/); algorithmic_* applies the contract's "
                          "direct-form count 2*B*HW*Cout*Cin*9 against the FP32 matrix peak (the precision the path delivers) and may exceed 1; "
                          + TRAFFIC_NOTE_FILE),
                    algorithmic_achieved=round(algorithmic, 2), algorithmic_frac=round(algorithmic / FP32_MFMA_PEAK_TFLOPS, 4),
                    algorithmic_bytes_per_launch=round(dom["bytes"] / dom["launches"]),
                    launches=dom["launches"], avg_launch_us=round(1e3 * dom["ms"] / dom["launches"], 1),
                    hbm_gbs=(round(traffic / (1e3 * dom["ms"] / dom["launches"]) / 1e3, 1) if traffic else None), hbm_peak_gbs=HBM_PEAK_GBS,
                    flops_per_launch_avg=dom["flops"] / dom["launches"],
                    conv3x3_families={k: dict(launches=v["launches"], ms=round(v["ms"], 3),
                                              algorithmic_tflops=round(v["flops"] / v["ms"] / 1e9, 2) if v["ms"] else None)
                                      for k, v in fam.items()},
                    conv1x1_kernels={str(k): v for k, v in sorted(k1.items())},
                    all_conv3x3=dict(launches=c3["launches"], ms=round(c3["ms"], 3), tflops=round(c3["flops"] / c3["ms"] / 1e9, 2)),
                    forward_ms_events=round(fwd_ms, 3),
                    forward_ms_events_note="sum of per-op event intervals of ONE instrumented forward, which runs op by op on the stream, OUTSIDE the "
                                           "hipGraph every other forward of the call replays: an upper bound of a replayed forward (event gaps, no "
                                           "overlap of neighbouring launches' ramp-up and tail); the line's forward_events_vs_step has the measured ratio",
                    breakdown=breakdown)
    n1 = sum(k1.values())
    if fam["wino_f16x2"]["launches"] or k1.get(14):
        arith = ("f16x2: f32 storage and accumulation; %d of %d 3x3 convs, %d of %d 1x1 convs (those with a GroupNorm-ed input) and the attention "
                 "products (head dims 32..128) multiply operands rounded to two fp16 pieces (22 significant bits, fp16 exponent range, 3 piece "
                 "products on the fp16 matrix pipe); the other convs on three exact bf16 pieces; everything else f32.  NARROWER than the "
                 "reference's fp32: opt-in (context option f16x2 = 1), never the headline value"
                 % (fam["wino_f16x2"]["launches"], c3["launches"], k1.get(14, 0), n1))
    elif fam["wino_bf16x3"]["launches"] or k1.get(15):
        arith = ("f32-equivalent: f32 storage and accumulation; %d of %d 3x3 convs, %d of %d 1x1 convs and the attention products (head dims "
                 "32..128) as SIX bf16 piece products of operands split EXACTLY into three bf16 pieces (all 24 bits, fp32 exponent range, less "
                 "than one fp32 rounding per product; tests/test_gpu_parity.py::test_conv_bf16x3_is_fp32_accurate, "
                 "test_default_kernels_have_the_fp32_range); the rest on the fp32 MFMA / fp32 VALU"
                 % (fam["wino_bf16x3"]["launches"] + fam["gemm_bf16x3"]["launches"], c3["launches"], k1.get(15, 0), n1))
    else:
        arith = "f32 (fp32 MFMA / fp32 VALU everywhere)"
    return roofline, arith


TRAFFIC_NOTE_FILE = "traffic is read from a committed PMC file (traffic_from_file), not measured in this run"
TRAFFIC_NOTE_RUN = "traffic was measured by this run's own rocprofv3 PMC passes (traffic_measured_in_run)"


def pmc_traffic(args, roofline):
    """`--pmc-traffic`: HBM-side bytes per launch of the dominant kernel family measured NOW, by two rocprofv3 PMC passes of this very
    script on this GPU (one counter per pass, kernel-trace only -- MI355X_MICROARCH.md's recipe; the outer process is idle meanwhile),
    under the same committed kernel table, 6 forwards each, no graph (one dispatch record per kernel).  Returns the roofline fields."""
    import shutil
    import tempfile
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import summarize_prof
    if shutil.which("rocprofv3") is None:
        return dict(traffic_measured_in_run=False, traffic_note="--pmc-traffic: rocprofv3 not on PATH")
    base = tempfile.mkdtemp(prefix="mcvd_pmc_", dir=os.environ.get("TMPDIR", "/tmp"))
    env = dict(os.environ, TMPDIR="/tmp", MCVD_BENCH_INNER="1", MCVD_ALLOW_SHARED_DEVICE="1")      # this (idle) process holds the device lock
    inner = [sys.executable, os.path.join(ROOT, "bench.py"), "--config", args.config, "--steps", "1", "--warmup", "0", "--subsample", "5",
             "--no-cpu-baseline", "--no-f16x2-leg", "--no-selfcheck", "--graph", "0", "--f16x2", str(args.f16x2)]
    if args.batch:
        inner += ["--batch", str(args.batch)]
    dirs = {}
    for counter in ("FETCH_SIZE", "WRITE_SIZE"):
        dirs[counter] = os.path.join(base, counter.lower())
        cmd = ["rocprofv3", "--kernel-trace", "--pmc", counter, "--output-format", "csv", "-d", dirs[counter], "-o", "bench", "--"] + inner
        rc = subprocess.call(cmd, env=env, cwd="/tmp", stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=600)
        if rc != 0:
            shutil.rmtree(base, ignore_errors=True)
            return dict(traffic_measured_in_run=False, traffic_note=f"--pmc-traffic: the {counter} pass returned {rc}")
    t = summarize_prof.family_traffic(dirs["FETCH_SIZE"], dirs["WRITE_SIZE"], roofline["kernel"])
    shutil.rmtree(base, ignore_errors=True)
    if not t:
        return dict(traffic_measured_in_run=False, traffic_note="--pmc-traffic: no dispatch of the dominant family in the PMC passes")
    tb = t["traffic_bytes_per_launch"]
    return dict(traffic=round(tb), traffic_from_file=False, traffic_measured_in_run=True,
                note=roofline["note"].replace(TRAFFIC_NOTE_FILE, TRAFFIC_NOTE_RUN), traffic_source="rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE "
                "passes spawned by this run (6 forwards each, same kernel table, no graph)", traffic_file_sha16=None,
                traffic_fetch_bytes_per_launch_corrected=round(t["fetch_bytes_per_launch_corrected"]),
                traffic_write_bytes_per_launch=round(t["write_bytes_per_launch"]),
                traffic_launches=[t["launches_fetch_pass"], t["launches_write_pass"]],
                traffic_vs_algorithmic=round(tb / roofline["algorithmic_bytes_per_launch"], 3),
                hbm_gbs=round(tb / (roofline["avg_launch_us"] * 1e-6) / 1e9, 1))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--config", default="smmnist_big5_ngf96", choices=sorted(DEFAULTS))
    ap.add_argument("--batch", type=int, default=None, help="samples per GPU (default: the config's per-GPU batch)")
    ap.add_argument("--global-batch", type=int, default=None, help="total samples of the job, sharded over the ranks by shard_rows (uneven "
                    "shards when it is not a multiple of --gpus); default: --batch x --gpus")
    ap.add_argument("--rank-spread-tol", type=float, default=1.10, help="N > 1 with even shards: the line is marked valid: false when the "
                    "slowest rank's own sampling time exceeds the fastest rank's by more than this factor (a straggler GPU: the whole-node "
                    "number then measures that GPU, not the design)")
    ap.add_argument("--subsample", type=int, default=None)
    ap.add_argument("--frames-pred", type=int, default=None, help="autoregressive configs: frames to keep (default 28)")
    ap.add_argument("--graph", type=int, default=int(os.environ.get("MCVD_GRAPH", "1")), help="hipGraph replay of the forwards")
    ap.add_argument("--tune-cache", default=None, help="JSON file: kernel-selection table of the headline leg -- loaded if it exists, else "
                    "written after the warm-up.  Default: profiles/tune_<config>_B<batch>_<arithmetic>.json when that file is committed "
                    "(pinned kernels: the run executes the table the parity tests check), else the autotuner measures")
    ap.add_argument("--save-tuning", default=None, help="directory: write the table of every leg there (tune_<config>_B<batch>_<arithmetic>.json)")
    ap.add_argument("--no-tune-file", action="store_true", help="ignore committed tables: let the autotuner measure")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-selfcheck", action="store_true", help="skip the B = 1 recomputation of row 0 after the timed region (profiling runs: its "
                    "101 single-sample forwards would enter the per-kernel averages of rocprofv3 --stats)")
    ap.add_argument("--dump-frames", default=None, help="file: rank 0 saves the gathered frames of the headline leg's last timed step (torch.save) "
                    "-- tests compare an N-rank job with the N = 1 job of the same global batch bit for bit")
    ap.add_argument("--no-f16x2-leg", "--no-fp32-leg", dest="no_second_leg", action="store_true",
                    help="skip the second, reported-only timing with the two-piece fp16 kernels offered")
    ap.add_argument("--pmc-traffic", action="store_true", help="N = 1: measure roofline.traffic IN THIS RUN -- after the timed region, two short "
                    "rocprofv3 passes of this script (--pmc FETCH_SIZE, then --pmc WRITE_SIZE; kernel-trace only, one counter per pass, 6 forwards "
                    "under the same kernel table) and (2 x FETCH + WRITE) / launches of the dominant family (MI355X_MICROARCH.md, HBM section). "
                    "Adds about a minute; without it the line quotes the committed PMC file (traffic_from_file)")
    ap.add_argument("--selfcheck-tol", type=float, default=1e-6, help="the run FAILS (valid: false, exit code 3) when row 0 of the last timed "
                    "call differs from the same row sampled alone by more than this (expected: exactly 0.0 under one kernel table)")
    ap.add_argument("--f16x2", type=int, default=0, help="1: the headline leg itself offers the two-piece fp16 kernels (narrower arithmetic "
                    "than the reference's: NOT the contract number; the default headline is the fp32-equivalent three-piece bf16 path)")
    args = ap.parse_args()

    cmd = plan_launch(args.gpus, os.environ)
    if cmd is not None:                                       # self-launch: N ranks, one per GPU; rank 0 prints the line
        env = dict(os.environ)
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        raise SystemExit(subprocess.call(cmd, env=env))

    import torch
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    import torch.distributed as dist
    backend = os.environ.get("MCVD_DIST_BACKEND", "nccl")     # "gloo": N>1 plumbing check on a box with fewer GPUs than ranks
    local = local % max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local))   # RCCL over xGMI
        else:
            dist.init_process_group(backend=backend)

    from mcvd_pytorch_amd import HipScoreNet, ddpm_sampler, synthetic, video_gen
    from mcvd_pytorch_amd.dist import broadcast_weights, gather_rows, shard_rows

    dB, dS, label = DEFAULTS[args.config]
    B = args.batch or dB
    subsample = args.subsample or dS
    config = make_config(args.config)
    config.sampling.subsample = subsample
    if args.frames_pred:
        config.sampling.num_frames_pred = args.frames_pred
    nfr, nfp = config.data.num_frames, config.sampling.num_frames_pred
    autoreg = nfp > nfr
    n_blocks = -(-nfp // nfr)
    config.device = f"cuda:{local}"
    if world > 1 and os.environ.get("MCVD_BENCH_SERIALIZE", "0") == "1":
        # several ranks on ONE GPU taking turns (tests): the library refuses a second process on a device (one process per GPU, api.cpp)
        # unless told that the sharing is deliberate
        os.environ.setdefault("MCVD_ALLOW_SHARED_DEVICE", "1")
    net = HipScoreNet(config)
    sd = None
    if rank == 0:
        sd = synthetic.random_state_dict(net, seed=123)
        net.load_state_dict(sd, strict=True)
    broadcast_weights(net, src=0)                    # ONE RCCL broadcast of the packed blob (no-op at N=1)
    net.set_option("profile", 1)
    net.set_option("graph", args.graph)
    if world > 1 and os.environ.get("MCVD_BENCH_SERIALIZE", "0") == "1":
        net.set_option("naive_attn", 4)      # every rank the same attention kernel, whatever a "share_fence" in MCVD_BENCH_OPTS would choose for a shared device
    for kv in os.environ.get("MCVD_BENCH_OPTS", "").split(","):      # diagnostics: context options, e.g. MCVD_BENCH_OPTS=conv_shape=10,naive_attn=2
        if kv:
            net.set_option(kv.split("=")[0], int(kv.split("=")[1]))

    total = args.global_batch or B * world
    b0, b1 = shard_rows(total, rank, world)
    assert b1 > b0, f"rank {rank} of {world} has no rows of a global batch of {total}"
    B = b1 - b0 if args.global_batch else B          # what this rank runs (the roofline figures of rank 0's line are per its own batch)
    x, cond = synthetic.random_inputs(config, b0, b1 - b0)
    x, cond = x.cuda(), cond.cuda()

    # MCVD_BENCH_SERIALIZE=1 (tests that put several ranks on ONE GPU): the ranks take turns on the device instead of overlapping.  Two
    # processes whose kernels share the CUs of one MI355X are not a supported configuration (one process per GPU is the design) -- and
    # not a safe one: co-resident kernels of the two processes corrupted each other's results (profiles/r04_two_process_corruption.txt).
    serialize = world > 1 and os.environ.get("MCVD_BENCH_SERIALIZE", "0") == "1"

    def local_step(i):
        if autoreg:            # the whole autoregressive job: n_blocks sampler calls, cond shifted on the device, crop to nfp frames
            g = torch.Generator(device="cuda").manual_seed(1000 + i)
            return video_gen(config, net, cond, num_frames_pred=nfp, seed=1000 + i, sample_offset=b0,
                             init_noise_fn=lambda k, shp, dev: torch.randn(shp, device=dev, generator=g))
        return ddpm_sampler(x, net, cond=cond, final_only=True, denoise=True, subsample_steps=subsample,
                            clip_before=True, verbose=False, log=False, seed=1000 + i, sample_offset=b0)[0]

    busy = [0.0]                                     # this rank's own sampling time (launch to device idle), without barrier / gather waits

    def timed_local_step(i):
        t = time.perf_counter()
        out = local_step(i)
        torch.cuda.synchronize()                     # (the gather needs the frames anyway)
        busy[0] += time.perf_counter() - t
        return out

    def one_step(i):
        if serialize:
            out = None
            for r in range(world):
                if r == rank:
                    out = timed_local_step(i)
                dist.barrier()
        else:
            out = timed_local_step(i)
        return gather_rows(out, total)               # final gather of the generated frames

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def table_path(arith_name):
        return os.path.join(ROOT, "profiles", f"tune_{args.config}_B{b1 - b0}_{arith_name}.json")

    def run_leg(arith_name, f16x2, warmup, steps, seed0, tune_file):
        """One timed leg under one arithmetic: W warm-up calls, then exactly K calls between fences (max over ranks)."""
        net.set_option("f16x2", f16x2)                  # (a change of the option drops the kernel table of the other arithmetic)
        pinned = None
        if tune_file and os.path.exists(tune_file) and not args.no_tune_file:
            net.load_tuning(tune_file)
            pinned = os.path.relpath(tune_file, ROOT)
        for i in range(warmup):
            one_step(seed0 - 1 - i)
        if rank == 0:
            if tune_file and pinned is None and tune_file == args.tune_cache:
                net.save_tuning(tune_file, [b1 - b0])
            if args.save_tuning:
                os.makedirs(args.save_tuning, exist_ok=True)
                net.save_tuning(os.path.join(args.save_tuning, os.path.basename(table_path(arith_name))), [b1 - b0])
        fence()
        busy[0] = 0.0
        t0 = time.perf_counter()
        for i in range(steps):
            frames = one_step(seed0 + i)
        fence()
        dt_local = time.perf_counter() - t0
        dt, per_rank, per_rank_busy, per_rank_rows = dt_local, [round(dt_local, 4)], [round(busy[0], 4)], [b1 - b0]
        if world > 1:
            tt = torch.tensor([dt_local, busy[0], float(b1 - b0)], device="cuda" if backend == "nccl" else "cpu", dtype=torch.float64)
            allt = [torch.zeros_like(tt) for _ in range(world)]
            dist.all_gather(allt, tt)
            per_rank = [round(v[0].item(), 4) for v in allt]
            per_rank_busy = [round(v[1].item(), 4) for v in allt]
            per_rank_rows = [int(v[2].item()) for v in allt]
            dt = max(per_rank)
        assert torch.isfinite(frames).all() and frames.shape[0] == total and frames.shape[1] == config.data.channels * nfp
        if args.dump_frames and rank == 0 and arith_name == main_arith:
            torch.save(frames.cpu(), args.dump_frames)
        roofline, arith = roofline_of_leg(net, args, B, arith_name)
        # ---- self-check, outside the timed region: row 0 of this rank's shard recomputed ALONE (B = 1) under the same kernel table with the
        # same Philox key (seed, global row) must reproduce the row the full batch produced -- a mis-indexed tile at the benchmarked batch
        # (arena offsets beyond 2^32 bytes, paired 8x8 regions, tiles that span images) cannot hide behind `isfinite`
        last_seed = seed0 + steps - 1
        kw = dict(final_only=True, denoise=True, subsample_steps=subsample, clip_before=True, verbose=False, log=False,
                  seed=1000 + last_seed, sample_offset=b0)
        selfcheck = None
        if not args.no_selfcheck:
            if autoreg:    # (video_gen draws its block inits from a torch generator whose stream depends on the batch: check one block's sampler call)
                full = ddpm_sampler(x, net, cond=cond, **kw)[0]
            else:
                full = frames[b0:b1]
            net.set_tuning(1, net.get_tuning(b1 - b0))
            for r in range(world if serialize else 1):
                if not serialize or r == rank:
                    one = ddpm_sampler(x[:1], net, cond=cond[:1], **kw)[0]
                    torch.cuda.synchronize()
                if serialize:
                    dist.barrier()
            selfcheck = float((full[:1] - one).abs().max().item())
        return dict(value=steps * total * nfp / dt, ms_per_step=1e3 * dt / steps, per_rank=per_rank, per_rank_busy=per_rank_busy,
                    per_rank_rows=per_rank_rows, roofline=roofline, dtype=arith,
                    kernel_table=pinned or "autotuned in this run (HIP-event timing per distinct layer shape)", selfcheck_max_abs=selfcheck)

    main_arith = "f16x2" if args.f16x2 else "bf16x3"
    main_leg = run_leg(main_arith, args.f16x2, args.warmup, args.steps, 0, args.tune_cache or table_path(main_arith))
    # ---- reported only (N = 1): the same timed region with the two-piece fp16 kernels offered as well (narrower arithmetic: an option)
    second = None
    if world == 1 and not args.no_second_leg and not args.f16x2:
        second = run_leg("f16x2", 1, 1, args.steps, 500, table_path("f16x2"))
        net.set_option("f16x2", 0)

    # which ranks / devices really took part (the driver's 8-GPU run is the first time RCCL ranks > 1 exist: make the line say what ran)
    ranks_seen = dict(world_size=world, backend=backend if world > 1 else None, devices=[local])
    if world > 1:
        mine = torch.tensor([rank, local], device="cuda" if backend == "nccl" else "cpu", dtype=torch.int64)
        allr = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(allr, mine)
        ranks_seen = dict(world_size=dist.get_world_size(), backend=backend, ranks=[int(v[0]) for v in allr], devices=[int(v[1]) for v in allr])
    if rank == 0:
        import ctypes as C
        from mcvd_pytorch_amd import _lib
        cap, rep = C.c_int64(), C.c_int64()
        _lib.lib.mcvd_model_graph_stats(net._model, C.byref(cap), C.byref(rep))
        fwd_per_step = (subsample + 1) * (n_blocks if autoreg else 1)
        res = dict(metric=f"sampled frames/sec (whole node), {label}", value=round(main_leg["value"], 3),
                   unit="frames/s", n_gpus=world, steps=args.steps, warmup=args.warmup,
                   ms_per_step=round(main_leg["ms_per_step"], 2), higher_is_better=True, scaling="weak", vs_baseline=None,
                   dtype=main_leg["dtype"], data="synthetic",
                   config=dict(workload=f"{args.config}: " + (f"video_gen, {n_blocks} autoregressive blocks of " if autoreg else "")
                               + f"ddpm_sampler subsample={subsample} (+1 denoise forward), "
                               f"{config.data.image_size}x{config.data.image_size}, {config.data.num_frames_cond} cond + {nfr} pred frames"
                               + (f" per block, {nfp} kept frames" if autoreg else "")
                               + f", batch {B}/GPU, random-init weights, Philox noise",
                               global_batch=total, frames_per_step=total * nfp, forwards_per_step=fwd_per_step,
                               parallelism=f"sample-sharded x{world} (1 weight broadcast + 1 final all_gather)",
                               kernel_table=main_leg["kernel_table"],
                               hip_graph=dict(enabled=bool(args.graph), captures=cap.value, replays=rep.value)),
                   per_rank_s=main_leg["per_rank"], per_rank_busy_s=main_leg["per_rank_busy"], per_rank_rows=main_leg["per_rank_rows"],
                   per_rank_spread=round(max(main_leg["per_rank_busy"]) / max(min(main_leg["per_rank_busy"]), 1e-9), 4),
                   per_rank_note="per_rank_s: each rank's wall time between the two fences (barrier-bounded: equal by construction); "
                                 "per_rank_busy_s: its own sampler calls, launch to device idle, without barrier / gather waits; "
                                 "per_rank_spread = max / min of the busy times",
                   roofline=main_leg["roofline"],
                   selfcheck_max_abs=main_leg["selfcheck_max_abs"],
                   selfcheck_note="max |row 0 of the last timed call - the same row sampled alone (B = 1, same kernel table, same Philox key)| "
                                  "over the final frames (data range [-1, 1]); computed after the timed region",
                   profile=1, profile_note="context option profile = 1: the first forward of each leg's LAST timed call runs op by op with HIP "
                                           "events around every launch (the roofline breakdown); it is inside the timed region (< 0.1 % of it)",
                   rccl_ranks_seen=ranks_seen)
        if second:
            res["f16x2_leg"] = dict(value=round(second["value"], 3), unit="frames/s", ms_per_step=round(second["ms_per_step"], 2),
                                    dtype=second["dtype"], kernel_table=second["kernel_table"], roofline=second["roofline"],
                                    selfcheck_max_abs=second["selfcheck_max_abs"],
                                    note="same workload, same K steps, context option f16x2 = 1 (1 warm-up call: re-tune or table load): "
                                         "reported only, narrower arithmetic than the reference's")
        # the instrumented forward against the replayed ones: (its event sum x forwards per call) / measured time per call
        res["roofline"]["forward_events_vs_step"] = round(main_leg["roofline"]["forward_ms_events"] * fwd_per_step / main_leg["ms_per_step"], 4)
        if args.pmc_traffic and world == 1:
            res["roofline"].update(pmc_traffic(args, main_leg["roofline"]))
        # ---- validity: a throughput line whose frames are wrong is not a measurement
        checks = [main_leg["selfcheck_max_abs"]] + ([second["selfcheck_max_abs"]] if second else [])
        bad = [c for c in checks if c is not None and not (c <= args.selfcheck_tol)]
        res["valid"] = not bad
        if bad:
            res["invalid_reason"] = (f"selfcheck_max_abs {bad[0]:.3e} > {args.selfcheck_tol:.1e}: row 0 of the benchmarked batch differs from the "
                                     "same row sampled alone under the same kernel table (expected 0.0)")
        even = len(set(main_leg["per_rank_rows"])) == 1
        if world > 1 and even and not serialize and res["per_rank_spread"] > args.rank_spread_tol:
            res["valid"] = False
            res["invalid_reason"] = (res.get("invalid_reason", "") + f" per_rank_spread {res['per_rank_spread']} > {args.rank_spread_tol}: ranks with equal "
                                     f"shards took {main_leg['per_rank_busy']} s for their own sampler calls -- a straggler GPU").strip()
        if world == 1 and not args.no_cpu_baseline:
            try:
                res["cpu_baseline"] = cpu_baseline(config, sd, subsample, kept_fraction=nfp / (n_blocks * nfr) if autoreg else 1.0)
                res["cpu_baseline"]["speedup"] = round(main_leg["value"] / res["cpu_baseline"]["value"], 1)
                res["cpu_baseline"]["speedup_note"] = "headline (fp32-equivalent) value / cpu_baseline value"
            except Exception as e:      # the baseline is reporting only; never lose the GPU line
                res["cpu_baseline"] = dict(value=None, unit="frames/s", cores=os.cpu_count(), kind="port", sample=f"failed: {e}")
        print(json.dumps(res), flush=True)
        if not res["valid"]:
            print("bench.py: INVALID RUN -- " + res["invalid_reason"], file=sys.stderr, flush=True)
            if bad:                          # wrong frames: a failed run (exit code 3).  A straggler GPU marks the line valid: false but the
                if world > 1:                # measurement itself is sound -- the line is printed and the process ends normally
                    dist.destroy_process_group()
                raise SystemExit(3)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

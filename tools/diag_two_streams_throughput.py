"""Does one GPU sample faster when a small per-GPU batch is split over TWO streams of the process?

Round 6 removed the cause of the co-residency corruption from the library's kernels (profiles/r06_coresident_cause.txt), so two contexts of one
process may overlap on the CUs.  The small-batch configs (4: B = 16, 5: B = 8) spend 13-16 % of a forward in launches shorter than 20 us and run
their 8 x 8 / 16 x 16 layers on half of the CUs (profiles/r06_launch_census_cfg4.txt): a second stream could fill those holes.  This script measures
it: the same sampler call (device loop, hipGraph replay) on ONE context at batch B against TWO contexts -- own stream, own thread, own graph --
at batch B / 2 each, and reports how far the frames are from the one-context rows (0 under one kernel table).

    python tools/diag_two_streams_throughput.py [config ...]        env STEPS=2 SUBSAMPLE=100
"""
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402


def main():
    import bench
    from mcvd_pytorch_amd import HipScoreNet, ddpm_sampler, synthetic
    steps = int(os.environ.get("STEPS", "2"))
    for name in (sys.argv[1:] or ["bair_big_spade", "cityscapes_big", "smmnist_big5_ngf96"]):
        B, dS, _ = bench.DEFAULTS[name]
        subsample = int(os.environ.get("SUBSAMPLE", "100"))
        config = bench.make_config(name)
        config.sampling.subsample = subsample
        config.device = "cuda:0"
        nfr = config.data.num_frames

        def make(stream, batch):
            with torch.cuda.stream(stream):
                net = HipScoreNet(config)
                net.load_state_dict(synthetic.random_state_dict(net, seed=123), strict=True)
                net.set_option("graph", 1)
                net.set_option("naive_attn", 4)          # the three-piece bf16 kernel whatever the device fence would choose
                tab = os.path.join(ROOT, "profiles", f"tune_{name}_B{batch}_bf16x3.json")
                if os.path.exists(tab):
                    net.load_tuning(tab)
                stream.synchronize()
            return net

        def call(net, stream, x, cond, b0, seed):
            with torch.cuda.stream(stream):
                out = ddpm_sampler(x, net, cond=cond, final_only=True, denoise=True, subsample_steps=subsample, clip_before=True, verbose=False, log=False,
                                   seed=seed, sample_offset=b0)[0]
                stream.synchronize()
            return out

        x, cond = synthetic.random_inputs(config, 0, B)
        x, cond = x.cuda(), cond.cuda()
        torch.cuda.synchronize()
        s0 = torch.cuda.Stream()
        one = make(s0, B)
        call(one, s0, x, cond, 0, 1)                       # warm-up: autotune (no table for this batch) + graph capture
        call(one, s0, x, cond, 0, 1)
        t0 = time.perf_counter()
        for i in range(steps):
            full = call(one, s0, x, cond, 0, 100 + i)
        t_one = (time.perf_counter() - t0) / steps
        del one
        res = {}
        for parts in (2, 3):
            if B % parts:
                continue
            h = B // parts
            streams = [torch.cuda.Stream() for _ in range(parts)]
            nets = [make(s, h) for s in streams]
            outs = [None] * parts

            def work(k, seed):
                outs[k] = call(nets[k], streams[k], x[k * h:(k + 1) * h], cond[k * h:(k + 1) * h], k * h, seed)

            def step(seed):
                th = [threading.Thread(target=work, args=(k, seed)) for k in range(parts)]
                for t in th:
                    t.start()
                for t in th:
                    t.join()
            step(1)
            step(1)
            t0 = time.perf_counter()
            for i in range(steps):
                step(100 + i)
            res[parts] = (time.perf_counter() - t0) / steps
            same = float((torch.cat(outs) - full).abs().max())      # (0 under one kernel table; the autotuner may pick other kernels at B / 2)
            res[parts] = (res[parts], same)
            del nets
        line = f"{name}: B = {B}, {subsample} steps: one context {B * nfr / t_one:8.2f} frames/s ({1e3 * t_one:.0f} ms per call)"
        for parts, (t, same) in res.items():
            line += f"; {parts} contexts x B = {B // parts} on {parts} streams {B * nfr / t:8.2f} frames/s ({1e3 * t:.0f} ms, x{t_one / t:.3f}, max |frames - one context's| {same:.1e})"
        print(line, flush=True)


if __name__ == "__main__":
    main()

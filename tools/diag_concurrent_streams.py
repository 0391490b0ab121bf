"""Two STREAMS of ONE process on one MI355X: does a kernel that corrupts its neighbour across PROCESSES
(attn_h2_kernel<3,3>, profiles/r04_two_process_corruption.txt) also do so across streams?  SURVEY 8b allows one context per stream from
one thread each, so this is a supported configuration and must be clean -- or fenced.
(Round 6: it IS clean since the cause was removed from the library's kernels -- profiles/r06_coresident_cause.txt; this script forces the
aggressor kernel past the fence, and tests/test_gpu_parity.py::test_former_victims_are_clean_beside_the_bf16_attention_kernel runs it.)

One process, two threads, each with its own HIP stream and its own mcvd context.  The VICTIM thread loops one op and compares every
result bit for bit with its own reference (taken while the other thread is idle); the AGGRESSOR thread keeps launching its op without
ever synchronising more than every 64 launches, so the two streams' kernels overlap on the CUs.  Roles are then swapped.

    python tools/diag_concurrent_streams.py          env SECS=6 (per phase)
"""
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402


def main():
    from tests.hiputil import Ctx
    secs = float(os.environ.get("SECS", "6"))
    g = torch.Generator().manual_seed(9)
    dev = torch.device("cuda:0")
    xf = torch.randn(3, 192, 32, 32, generator=g).to(dev)
    coeff = torch.stack([1 + 0.3 * torch.randn(3, 192, generator=g), 0.3 * torch.randn(3, 192, generator=g)], dim=-1).to(dev)
    qkv = torch.randn(3, 3 * 2 * 96, 1024, generator=g).to(dev)
    x3 = torch.randn(3, 96, 64, 64, generator=g).to(dev)
    w3 = (torch.randn(96, 96, 3, 3, generator=g) / 29).to(dev)
    b3 = torch.zeros(96).to(dev)
    coef3 = torch.ones(3, 96, 2).to(dev)
    streams = [torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)]
    ctxs = []
    for s in streams:
        with torch.cuda.stream(s):
            ctxs.append(Ctx())
    torch.cuda.synchronize()

    def op(kind, c):
        if kind == "fir":                       # the x2 FIR upsampler as the library launches it today (LDS strip form)
            c.opt("fir_form", 0)
            return c.fir2(xf, 1, coef=coeff, act=1)
        if kind == "fir1":                      # fir_up2_kernel, the register form: the victim of the round-4 / round-5 reports
            c.opt("fir_form", 1)
            return c.fir2(xf, 1, coef=coeff, act=1)
        if kind == "attn96":                    # attn_h2_kernel<3,3>: the aggressor of the two-process report
            c.opt("naive_attn", 4)
            return c.attention(qkv, 2)
        if kind == "attn96f32":
            c.opt("naive_attn", 2)
            return c.attention(qkv, 2)
        if kind == "c3":                        # direct conv: the second victim of the report
            c.opt("conv_shape", 0)
            return c.conv2d(x3, w3, b3, coef=coef3, act=1)
        raise KeyError(kind)

    def phase(victim, aggressor):
        stop = threading.Event()
        n_aggr = [0]

        def aggr():
            with torch.cuda.stream(streams[1]):
                while not stop.is_set():
                    for _ in range(64):
                        op(aggressor, ctxs[1])
                    n_aggr[0] += 64
                    streams[1].synchronize()
        with torch.cuda.stream(streams[0]):
            ref = op(victim, ctxs[0]).clone()
            streams[0].synchronize()
            assert torch.equal(op(victim, ctxs[0]), ref), "victim is not deterministic alone"
        th = None
        if aggressor:
            th = threading.Thread(target=aggr)
            th.start()
            time.sleep(0.2)
        bad, n, shown = 0, 0, 0
        t0 = time.time()
        with torch.cuda.stream(streams[0]):
            while time.time() - t0 < secs:
                o = op(victim, ctxs[0])
                n += 1
                if not torch.equal(o, ref):
                    bad += 1
                    if shown < 2:
                        shown += 1
                        idx = (o != ref).nonzero()
                        print(f"   diff: {idx.shape[0]} of {o.numel()} elements, max {float((o - ref).abs().max()):.3e}, first {idx[0].tolist()} "
                              f"last {idx[-1].tolist()}", flush=True)
        stop.set()
        if th:
            th.join()
        torch.cuda.synchronize()
        print(f"victim {victim:10s} beside {str(aggressor):10s} on another stream of the same process: {bad} of {n} launches differ "
              f"({n_aggr[0]} aggressor launches meanwhile)", flush=True)
        return bad

    total = 0
    plan = (("fir", None), ("fir", "attn96"), ("fir1", None), ("fir1", "attn96"), ("c3", None), ("c3", "attn96"), ("attn96", "fir"), ("fir", "attn96f32"), ("c3", "attn96f32"),
            ("attn96", "attn96"))
    for victim, aggressor in plan:
        total += phase(victim, aggressor)
    print("TOTAL corrupted launches across streams of one process:", total, flush=True)
    return total


if __name__ == "__main__":
    main()

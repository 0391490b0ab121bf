"""Bit-determinism of single ops while ANOTHER process keeps the same GPU busy (two processes on one device: their workgroups share CUs,
which a single-stream process never sees).  usage: diag_concurrent_ops.py [worker TAG]  -- without arguments starts two workers."""
import os, sys, subprocess, time
os.environ.setdefault("MCVD_ALLOW_SHARED_DEVICE", "1")     # this tool puts two processes on one device ON PURPOSE (api.cpp: device lock)
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

def worker(tag):
    import torch
    from tests.hiputil import Ctx
    ctx = Ctx()
    g = torch.Generator().manual_seed(9)
    cases = []
    for (B, C, heads, H) in ((3, 192, 2, 32), (3, 288, 3, 16), (3, 384, 4, 8)):
        qkv = torch.randn(B, 3 * C, H * H, generator=g); qkv[:, :C] *= 1.5
        cases.append(("attn", (B, C, heads, H), qkv.cuda()))
    x = torch.randn(3, 192, 32, 32, generator=g).cuda(); w = (torch.randn(576, 192, 1, 1, generator=g) / 14).cuda(); b = torch.zeros(576).cuda()
    x3 = torch.randn(3, 96, 64, 64, generator=g).cuda(); w3 = (torch.randn(96, 96, 3, 3, generator=g) / 29).cuda(); b3 = torch.zeros(96).cuda()
    coef3 = torch.ones(3, 96, 2).cuda()
    qkvD = {D: torch.randn(3, 3 * 2 * D, 1024, generator=g).cuda() for D in (32, 64, 96, 128)}
    xf = torch.randn(3, 192, 32, 32, generator=g).cuda(); coeff = torch.stack([1 + 0.3 * torch.randn(3, 192, generator=g), 0.3 * torch.randn(3, 192, generator=g)], dim=-1).cuda()
    def run(kind, mode):
        outs = []
        if kind.startswith("attnD"):            # one (sample, head) shape with head dim D = kind[5:], 32 x 32 pixels
            D = int(kind[5:])
            ctx.opt("naive_attn", mode)
            outs.append(ctx.attention(qkvD[D], 2))
            ctx.opt("naive_attn", 0)
        elif kind == "attn":
            ctx.opt("naive_attn", mode)
            for _, (B, C, heads, H), qkv in cases:
                outs.append(ctx.attention(qkv, heads))
            ctx.opt("naive_attn", 0)
        elif kind == "fir":
            outs.append(ctx.fir2(xf, mode, coef=coeff, act=1))
        elif kind == "c1":
            ctx.opt("conv_shape", mode); ctx.opt("conv_cot", 3)
            outs.append(ctx.conv2d(x, w, b))
            ctx.opt("conv_shape", -1); ctx.opt("conv_cot", 0)
        else:
            ctx.opt("conv_shape", mode)
            outs.append(ctx.conv2d(x3, w3, b3, coef=coef3, act=1))
            ctx.opt("conv_shape", -1)
        return outs
    res = {}
    plan = (("attn", 4), ("attn", 2), ("attn", 3), ("c1", 15), ("c1", 5), ("c3", 10), ("c3", 16), ("c3", 4))
    if os.environ.get("PLAN"):          # e.g. PLAN="attn:4,c1:15;c1:15,attn:4": worker i runs the i-th list (so that DIFFERENT kernels overlap)
        mine = os.environ["PLAN"].split(";")[int(tag[1])].split(",")
        plan = tuple((m.split(":")[0], int(m.split(":")[1])) for m in mine)      # tokens kind:mode, e.g. attn:4, c1:15, c3:10
    for phase, (kind, mode) in enumerate(plan):
        ref = [o.clone() for o in run(kind, mode)]
        for _ in range(4):                      # the reference itself: the majority of a few runs
            again = [o.clone() for o in run(kind, mode)]
            if all(torch.equal(a, r) for a, r in zip(again, ref)):
                break
            ref = again
        bad = 0
        t0 = time.time()
        n = 0
        shown = 0
        while time.time() - t0 < float(os.environ.get("SECS", "4")):
            outs = run(kind, mode)
            n += 1
            for o, r in zip(outs, ref):
                if not torch.equal(o, r):
                    bad += 1
                    if shown < 3:
                        shown += 1
                        idx = (o != r).nonzero()
                        dmax = (o - r).abs().max().item()
                        print(tag, phase, kind, mode, "diff elements", idx.shape[0], "of", o.numel(), "max", f"{dmax:.3e}", "first", idx[0].tolist(), "last", idx[-1].tolist(),
                              "distinct (b, c):", len({(int(a), int(b)) for a, b in idx[:, :2].tolist()}), flush=True)
        torch.cuda.synchronize()
        res[(f'{phase}:{kind}', mode)] = (bad, n)
    print(tag, {f"{k}{m}": v for (k, m), v in res.items()}, flush=True)

if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "worker":
        worker(sys.argv[2])
    else:
        n = int(os.environ.get("NPROC", "2"))
        ps = [subprocess.Popen([sys.executable, os.path.abspath(__file__), "worker", f"p{i}of{n}"]) for i in range(n)]
        for p in ps:
            p.wait()

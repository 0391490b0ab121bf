"""Whole-net bit-determinism while another process runs the same net on the same GPU: on a mismatch, the first module whose output
differs from the reference run.  usage: diag_concurrent_net.py [worker TAG];  env OPTS=conv_shape=10,... NPROC=2 SECS=8"""
import os, sys, subprocess, time
os.environ.setdefault("MCVD_ALLOW_SHARED_DEVICE", "1")     # this tool puts two processes on one device ON PURPOSE (api.cpp: device lock)
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

def worker(tag):
    import torch, ctypes as C
    from oracle import synth
    from mcvd_pytorch_amd import _lib
    from mcvd_pytorch_amd.scorenet import HipScoreNet
    from tests.hiputil import module_output
    config = synth.make_config("smmnist_big5_ngf96"); config.device = "cuda:0"
    net = HipScoreNet(config); net.load_state_dict(synth.make_state_dict(config, seed=123), strict=True); net.eval()
    for kv in os.environ.get("OPTS", "").split(","):
        if kv:
            net.set_option(kv.split("=")[0], int(kv.split("=")[1]))
    B = 3
    x, cond = synth.make_inputs(config, B, seed=0)
    x, cond = x.cuda(), cond.cuda()
    t = torch.tensor([37, 74, 111]).cuda()
    nmod = 60
    def taps():
        out = {}
        for i in range(1, nmod):
            try:
                out[i] = module_output(net, i, B).clone()
            except RuntimeError:
                pass
        return out
    for _ in range(3):
        ref = net(x, t, cond=cond).clone()
    ref_taps = taps()
    again = net(x, t, cond=cond)
    assert torch.equal(again, ref), "reference itself unstable"
    bad = n = shown = 0
    t0 = time.time()
    while time.time() - t0 < float(os.environ.get("SECS", "8")):
        e = net(x, t, cond=cond)
        n += 1
        if not torch.equal(e, ref):
            bad += 1
            if shown < 4:
                shown += 1
                cur = taps()
                for i in sorted(cur):
                    if i in ref_taps and not torch.equal(cur[i], ref_taps[i]):
                        d = cur[i] != ref_taps[i]
                        idx = d.nonzero()
                        print(tag, "first differing module", i, "shape", tuple(cur[i].shape), "diff elements", int(d.sum()), "max", f"{(cur[i] - ref_taps[i]).abs().max().item():.3e}",
                              "first", idx[0].tolist(), "last", idx[-1].tolist(), flush=True)
                        break
                else:
                    print(tag, "eps differs but no module tap does", flush=True)
    print(tag, "forwards", n, "mismatching", bad, flush=True)

if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "worker":
        worker(sys.argv[2])
    else:
        n = int(os.environ.get("NPROC", "2"))
        ps = [subprocess.Popen([sys.executable, os.path.abspath(__file__), "worker", f"p{i}of{n}"]) for i in range(n)]
        for p in ps:
            p.wait()

"""Row-independence probe: eps of row 0 at B = 1 / 3 / 6 under one kernel table (forced families or the table autotuned at B = 6)."""
import sys, os, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from oracle import synth
from mcvd_pytorch_amd.scorenet import HipScoreNet
from mcvd_pytorch_amd.samplers import ddpm_sampler

def net_for(name):
    config = synth.make_config(name); config.device = "cuda:0"
    net = HipScoreNet(config); net.load_state_dict(synth.make_state_dict(config, seed=123), strict=True)
    return config, net.eval()

config, net = net_for("smmnist_big5_ngf96")
x, cond = synth.make_inputs(config, 6, seed=0)
t = torch.tensor([37, 74, 111, 148, 185, 222])
def fwd(B, rows=None):
    r = list(range(B)) if rows is None else rows
    return net(x[r].cuda(), t[r].cuda(), cond=cond[r].cuda()).cpu()
for mode in ("auto6", 10, 11, 16, 17):
    if mode == "auto6":
        net.set_option("conv_shape", -1)
        e6 = fwd(6)
        table = net.get_tuning(6)
        import collections
        print("table@6", collections.Counter(s for s, c in table))
        for b in (1, 2, 3, 4, 5):
            net.set_tuning(b, table)
    else:
        net.set_option("conv_shape", mode); net.set_option("conv_shape1", 15)
        e6 = fwd(6)
    for B in (1, 2, 3, 4, 5):
        e = fwd(B)
        d = (e - e6[:B]).abs().flatten(1).max(dim=1).values
        print(mode, "B", B, "vs B=6 per-row max diff", [f"{v:.2e}" for v in d.tolist()])
# the sampler, device loop, seeds as bench.py
net.set_option("conv_shape", -1); net.set_option("conv_shape1", -1)
kw = dict(final_only=True, denoise=True, subsample_steps=5, clip_before=True, verbose=False, log=False, seed=1001, sample_offset=0)
o6 = ddpm_sampler(x.cuda(), net, cond=cond.cuda(), **kw)[0].cpu()
for B in (1, 3):
    o = ddpm_sampler(x[:B].cuda(), net, cond=cond[:B].cuda(), **kw)[0].cpu()
    print("sampler B", B, "vs B=6:", [f"{v:.2e}" for v in (o - o6[:B]).abs().flatten(1).max(dim=1).values.tolist()])

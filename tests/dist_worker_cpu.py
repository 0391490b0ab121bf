"""One rank of the CPU stand-in for bench.py's multi-rank job (tests/test_dist_cpu.py starts N of these with the exact launcher command
bench.plan_launch() builds for `--gpus N`, under MCVD_DIST_BACKEND=gloo): reads RANK / WORLD_SIZE / MASTER_* from the environment as
bench.py does, receives the weights through mcvd_pytorch_amd.dist.broadcast_weights (ONE broadcast), samples its contiguous row shard
with the CPU oracle as the compute stand-in (test infrastructure), gathers with gather_rows (ONE all_gather) and lets rank 0 write the
gathered frames + the ranks that took part.  usage: dist_worker_cpu.py TOTAL_ROWS OUT_DIR"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import torch
import torch.distributed as dist

from mcvd_pytorch_amd import dist as mdist
from oracle import synth, unet_ref
from tests.test_dist_cpu import FakeNet, _sampler


def main():
    total, out_dir = int(sys.argv[1]), sys.argv[2]
    world, rank = int(os.environ["WORLD_SIZE"]), int(os.environ["RANK"])
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    dist.init_process_group(backend=os.environ.get("MCVD_DIST_BACKEND", "gloo"))
    assert dist.get_world_size() == world and dist.get_rank() == rank
    torch.set_num_threads(1)
    config = synth.make_config("tiny")
    sd0 = synth.make_state_dict(config, seed=123)
    names = list(sd0.keys())
    flat0 = torch.cat([sd0[k].reshape(-1) for k in names])
    fake = FakeNet(flat0 if rank == 0 else torch.zeros_like(flat0))       # weights live on rank 0 only
    mdist.broadcast_weights(fake, src=0)
    sd, off = {}, 0
    for k in names:
        n = sd0[k].numel()
        sd[k] = fake.blob[off:off + n].view_as(sd0[k])
        off += n
    net = unet_ref.OracleScoreNet(config, sd)
    b0, b1 = mdist.shard_rows(total, rank, world)
    xf = lambda b, e: synth.make_inputs(config, e, seed=0)[0][b:e]
    cf = lambda b, e: synth.make_inputs(config, e, seed=0)[1][b:e]
    if b1 > b0:
        out = mdist.sample_sharded(_sampler, net, xf, cf, total, config=config)
    else:                                     # more ranks than rows: an empty shard still takes part in the gather
        c = unet_ref.hot_cfg(config)
        out = mdist.gather_rows(torch.zeros(0, c.channels * c.num_frames, c.image_size, c.image_size), total)
    mine = torch.tensor([rank, b0, b1], dtype=torch.int64)
    seen = [torch.zeros_like(mine) for _ in range(world)]
    dist.all_gather(seen, mine)
    if rank == 0:
        torch.save(dict(frames=out, ranks=[v.tolist() for v in seen]), os.path.join(out_dir, "gathered.pt"))
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()

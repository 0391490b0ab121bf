"""CPU, world_size 2, gloo: the N>1 path (row sharding, one weight broadcast, one final gather, noise keyed by GLOBAL
sample index) gives exactly the single-process result.  The compute stand-in is the CPU oracle (checker code)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from mcvd_pytorch_amd import dist as mdist
from oracle import sampler_ref, synth, unet_ref


def test_shard_rows_partition():
    for total in (1, 5, 8, 64, 257):
        for world in (1, 2, 3, 8):
            spans = [mdist.shard_rows(total, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == total
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
            sizes = [e - b for b, e in spans]
            assert max(sizes) - min(sizes) <= 1


class FakeNet:
    """Blob protocol of HipScoreNet on CPU tensors (broadcast_weights only needs these four members)."""

    def __init__(self, values):
        self.device = torch.device("cpu")
        self.blob = values.clone()

    def blob_numel(self):
        return self.blob.numel()

    def export_blob(self):
        return self.blob.clone()

    def import_blob(self, b):
        self.blob = b.clone()

    def sync_parameters(self):
        pass


def _row_noise(config, row, n_steps):
    c = unet_ref.hot_cfg(config)
    g = torch.Generator().manual_seed(9000 + row)
    return torch.randn(n_steps, c.channels * c.num_frames, c.image_size, c.image_size, generator=g)


def _sampler(x, net, cond=None, sample_offset=0, config=None, **kw):
    rows = x.shape[0]
    noise = torch.stack([_row_noise(config, sample_offset + r, 11) for r in range(rows)], dim=1)   # keyed by global row
    k = [0]

    def fn(i, like):
        k[0] += 1
        return noise[k[0] - 1]
    return sampler_ref.sample(x, net, cond=cond, kind="ddpm", final_only=True, denoise=True, subsample_steps=10,
                              noise_fn=fn)


def _worker(rank, world, port, total, tmp):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(2)
    config = synth.make_config("tiny")
    # weights live on rank 0 only; ONE broadcast of the flat blob
    sd0 = synth.make_state_dict(config, seed=123)
    names = list(sd0.keys())
    flat0 = torch.cat([sd0[k].reshape(-1) for k in names])
    fake = FakeNet(flat0 if rank == 0 else torch.zeros_like(flat0))
    mdist.broadcast_weights(fake, src=0)
    assert torch.equal(fake.blob, flat0)
    sd, off = {}, 0
    for k in names:
        n = sd0[k].numel()
        sd[k] = fake.blob[off:off + n].view_as(sd0[k])
        off += n
    net = unet_ref.OracleScoreNet(config, sd)
    xf = lambda b, e: synth.make_inputs(config, e, seed=0)[0][b:e]
    cf = lambda b, e: synth.make_inputs(config, e, seed=0)[1][b:e]
    out = mdist.sample_sharded(_sampler, net, xf, cf, total, config=config)
    if rank == 0:
        torch.save(out, os.path.join(tmp, "sharded.pt"))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_sharded_sampling_matches_single_process(tmp_path):
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    total = 3                                   # uneven shards: 2 + 1 rows
    mp.spawn(_worker, args=(2, port, total, str(tmp_path)), nprocs=2, join=True)
    got = torch.load(os.path.join(str(tmp_path), "sharded.pt"))
    config = synth.make_config("tiny")
    net = unet_ref.OracleScoreNet(config, synth.make_state_dict(config, seed=123))
    x, cond = synth.make_inputs(config, total, seed=0)
    want = _sampler(x, net, cond=cond, sample_offset=0, config=config)[0]
    assert got.shape == want.shape
    # oneDNN picks batch-/thread-dependent blockings: fp32 noise floor between CPU runs (1.7e-6 per forward)
    assert (got - want).abs().max().item() <= 2e-5


def test_eight_rank_launch_plan_with_uneven_shards(tmp_path):
    """The launcher command bench.py builds for `--gpus 8` (bench.plan_launch: python -m torch.distributed.run, one rank per GPU,
    127.0.0.1 rendezvous) started for real with 8 gloo ranks on the CPU, tests/dist_worker_cpu.py in place of bench.py: 11 rows over 8
    ranks (shards of 2, 2, 2, 1, 1, 1, 1, 1), one weight broadcast, one gather -- the gathered frames must equal the single-process
    result and every rank must report the shard shard_rows assigns it (VERDICT r3 task 7: the N = 8 rank / offset plumbing has to have
    run somewhere before the driver's 8-GPU node does)."""
    import importlib.util
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(root, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    total, world = 11, 8
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env.update(MASTER_PORT=str(port), MCVD_DIST_BACKEND="gloo", OMP_NUM_THREADS="1")
    cmd = bench.plan_launch(world, env, argv=[str(total), str(tmp_path)])
    assert cmd is not None and cmd[-3].endswith("bench.py")
    cmd[-3] = os.path.join(root, "tests", "dist_worker_cpu.py")                 # the CPU stand-in for the per-rank body of bench.py
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    got = torch.load(os.path.join(str(tmp_path), "gathered.pt"))
    assert [tuple(v) for v in got["ranks"]] == [(r_,) + mdist.shard_rows(total, r_, world) for r_ in range(world)]
    config = synth.make_config("tiny")
    net = unet_ref.OracleScoreNet(config, synth.make_state_dict(config, seed=123))
    x, cond = synth.make_inputs(config, total, seed=0)
    want = _sampler(x, net, cond=cond, sample_offset=0, config=config)[0]
    assert got["frames"].shape == want.shape
    assert (got["frames"] - want).abs().max().item() <= 2e-5

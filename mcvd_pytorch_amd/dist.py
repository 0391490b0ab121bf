"""Multi-GPU sampling: independent video samples shard across ranks.

One process per GPU (torch.distributed; backend "nccl" is RCCL on ROCm, "gloo" in CPU tests).  The data path has NO
per-step collective: one broadcast of the packed weight blob over xGMI at load time, every rank runs the whole L-step
loop on its contiguous slice of rows, one all_gather of the final frames.  (The reference's nn.DataParallel
re-broadcasts all weights and gathers eps on every forward -- runners/ncsn_runner.py:924 -- which this replaces.)
"""
import torch
import torch.distributed as dist


def shard_rows(total_rows, rank, world):
    """Contiguous [begin, end) slice of rank `rank`; remainders go to the first ranks."""
    base, rem = divmod(total_rows, world)
    begin = rank * base + min(rank, rem)
    return begin, begin + base + (1 if rank < rem else 0)


def _host_staged():
    """gloo moves GPU tensors through the host (it has no device all_gather); RCCL ("nccl") works on device memory."""
    return dist.get_backend() == "gloo"


def broadcast_weights(net, src=0):
    """ONE collective: rank `src` exports its parameter blob, everyone imports it."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        net.sync_parameters()
        return
    if dist.get_rank() == src:
        blob = net.export_blob()
    else:
        blob = torch.empty(net.blob_numel(), dtype=torch.float32, device=net.device)
    if _host_staged() and blob.is_cuda:
        host = blob.cpu()
        dist.broadcast(host, src=src)
        blob = host.to(net.device)
    else:
        dist.broadcast(blob, src=src)
    if dist.get_rank() != src:
        net.import_blob(blob)


def gather_rows(local, total_rows):
    """all_gather of per-rank row blocks [rows_r, ...] -> [total_rows, ...] (uneven shards are padded then trimmed)."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return local
    world = dist.get_world_size()
    sizes = [shard_rows(total_rows, r, world) for r in range(world)]
    mx = max(e - b for b, e in sizes)
    dev = local.device
    pad = torch.zeros((mx,) + tuple(local.shape[1:]), dtype=local.dtype, device=dev)
    pad[:local.shape[0]] = local
    if _host_staged() and pad.is_cuda:
        pad = pad.cpu()
    out = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(out, pad)
    return torch.cat([o[:e - b] for o, (b, e) in zip(out, sizes)], dim=0).to(dev)


def sample_sharded(sampler, scorenet, x_full_fn, cond_full_fn, total_rows, **sampler_kwargs):
    """Run `sampler` on this rank's rows and gather.  `x_full_fn(begin, end)` / `cond_full_fn(begin, end)` produce the
    rows [begin, end) of the global batch (so no rank ever materialises the full inputs); the on-device noise stream is
    keyed by global row through `sample_offset`."""
    rank = dist.get_rank() if dist.is_initialized() else 0
    world = dist.get_world_size() if dist.is_initialized() else 1
    b, e = shard_rows(total_rows, rank, world)
    out = sampler(x_full_fn(b, e), scorenet, cond=cond_full_fn(b, e), sample_offset=b, **sampler_kwargs)
    frames = out[-1] if out.dim() == 5 else out
    return gather_rows(frames.contiguous(), total_rows)

"""Synthetic weights / inputs for benchmarking (no dataset, no checkpoint: SURVEY 8d).

Parameters are re-randomised with variance-preserving scales -- never the reference's default init, which zeroes
217/376 tensors and would make any kernel "correct".  Rows of the inputs are keyed by GLOBAL sample
index so a rank's shard equals the same rows of the unsharded batch.
"""
import zlib

import torch


def _gen(seed, name):
    g = torch.Generator()
    g.manual_seed((seed * 1000003 + zlib.crc32(name.encode())) & 0x7FFFFFFF)
    return g


def random_state_dict(net, seed=123):
    """name -> CPU fp32 tensor for every parameter of a HipScoreNet: N(0, 1/fan_in) matrices/filters,
    1 + 0.1 N(0,1) norm gains, 0.1 N(0,1) biases."""
    sd = {}
    for name, p in net.named_parameters():
        shape = tuple(p.shape)
        g = _gen(seed, name)
        if len(shape) >= 2:
            fan_in = shape[0] if name.endswith(".W") else int(torch.tensor(shape[1:]).prod())
            t = torch.randn(shape, generator=g) * (1.0 / fan_in) ** 0.5
        elif name.endswith(".weight"):
            t = 1.0 + 0.1 * torch.randn(shape, generator=g)
        else:
            t = 0.1 * torch.randn(shape, generator=g)
        sd[name] = t.float()
    return sd


def random_inputs(config, first_row, rows, seed=0):
    """(x_init, cond) for global rows [first_row, first_row + rows): x ~ N(0,1), cond ~ clamp(N(0,1), -1, 1)."""
    d = config.data
    S, C = d.image_size, d.channels
    nc = d.num_frames_cond + getattr(d, "num_frames_future", 0)
    xs, cs = [], []
    for b in range(first_row, first_row + rows):
        xs.append(torch.randn(C * d.num_frames, S, S, generator=_gen(seed, f"x{b}")))
        cs.append(torch.randn(C * nc, S, S, generator=_gen(seed + 1, f"c{b}")).clamp(-1, 1))
    return torch.stack(xs), torch.stack(cs)

"""CPU: the caller-side glue restated in mcvd_pytorch_amd/runner.py (conditioning layout, data transforms)."""
import torch

from oracle import synth


def _runner():
    from mcvd_pytorch_amd import runner
    return runner


def test_conditioning_fn_layout_is_frame_major():
    """runners/ncsn_runner.py:115-118: channel index = t*C + c; cond = [past..., future...]."""
    r = _runner()
    cfg = synth.make_config("tiny_spade")           # C=3, nf=2, past=1, future=1
    d = cfg.data
    B, C, S = 2, d.channels, d.image_size
    T = d.num_frames_cond + d.num_frames + d.num_frames_future
    X = torch.arange(B * T * C).float().reshape(B, T, C, 1, 1).expand(B, T, C, S, S).contiguous()
    pred, cond, mask = r.conditioning_fn(cfg, X, num_frames_pred=d.num_frames)
    assert mask is None
    assert pred.shape == (B, C * d.num_frames, S, S) and cond.shape == (B, C * (d.num_frames_cond + d.num_frames_future), S, S)
    for t in range(d.num_frames):
        for c in range(C):
            assert torch.equal(pred[:, t * C + c], X[:, d.num_frames_cond + t, c])
    assert torch.equal(cond[:, :C], X[:, 0])                                    # past frame
    assert torch.equal(cond[:, C:2 * C], X[:, d.num_frames_cond + d.num_frames])  # future frame
    _, cond0, _ = r.conditioning_fn(cfg, X, num_frames_pred=d.num_frames, prob_mask_future=1.0)
    assert cond0[:, C:].abs().max() == 0                                        # :130-131 all-zero future block


def test_data_transform_round_trip():
    r = _runner()
    cfg = synth.make_config("tiny")
    X = torch.rand(2, 4, 1, 8, 8)
    Y = r.data_transform(cfg, X)
    assert Y.min() >= -1 and Y.max() <= 1 and torch.allclose(Y, 2 * X - 1)
    assert torch.allclose(r.inverse_data_transform(cfg, Y), X, atol=1e-6)
    assert r.inverse_data_transform(cfg, Y * 3).max() <= 1.0                    # clamp (datasets/__init__.py:261)


def test_save_video_pred_format(tmp_path):
    """videos_pred_<ckpt>.pt: a dict of CPU tensors with the reference's keys (ncsn_runner.py:2106-2112)."""
    r = _runner()
    cond, pred, real = torch.rand(2, 4, 8, 8), torch.rand(2, 6, 8, 8), torch.rand(2, 6, 8, 8)
    p = r.save_video_pred(str(tmp_path / "videos_pred_1000.pt"), cond, pred, real)
    d = torch.load(p, weights_only=False)
    assert sorted(d) == ["cond", "pred", "real"] and torch.equal(d["pred"], pred) and d["cond"].device.type == "cpu"

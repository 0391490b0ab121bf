"""Golden vectors from the REAL `NCSNRunner` -- TEST INFRASTRUCTURE ONLY (build container only: needs /root/reference).

    PYTHONDONTWRITEBYTECODE=1 python -m oracle.gen_runner_golden

`runners/ncsn_runner.py` does not import in this image as it stands (imageio, cv2, skimage, torchvision, h5py, progressbar, lpips ...
are absent: SURVEY 8c).  None of those is on the sampling path -- they serve dataset readers, metrics and plots -- so this script
installs an import hook that hands out inert stand-in MODULES for exactly those third-party names, imports the REAL runner module, and
drives the REAL code, unmodified, through the lines the drop-in boundary is about:

    get_model(config)                                   runners/ncsn_runner.py:180-195
    NCSNRunner(args, config, None).get_sampler()        :2702-2714
    NCSNRunner.video_gen(scorenet=..., ckpt=0)          :1304-1569  (the prediction path: data_transform, conditioning_fn, init z, the
                                                        autoregressive block loop with its cond shift, the crop to num_frames_pred)

`video_gen` goes on to metrics and plots after :1569; the run is cut there by a sentinel raised from the wrapper around the call
`inverse_data_transform(self.config, pred)` (:1570), which receives the finished `pred`.  What is replaced, and only from the outside:
`get_dataset` (a synthetic in-memory clip set: there is no dataset on disk), `eval_models.PerceptualLoss` (never reached),
`torch.randn` / `torch.randn_like` (recorded / injected so a GPU run can consume the same draws).  No reference source is copied:
the fixture holds tensors.

Fixture tests/golden/tiny_runner_videogen.pt:
    clips [N, T, C, H, W] in [0, 1] (the dataset), order (the rows the shuffling DataLoader served), real_t (after data_transform),
    real / cond / cond_mask from conditioning_fn, z_init [blocks, B, C*nf, S, S], step noise per block, pred_raw [B, C*nfp, S, S],
    pred01 (after the real inverse_data_transform), real01, cond01, sampler kwargs the runner passed, partial types get_sampler returned.
"""
import argparse
import importlib.abc
import importlib.machinery
import os
import sys
import tempfile
from unittest import mock

os.environ.setdefault("PYTHONDONTWRITEBYTECODE", "1")
sys.dont_write_bytecode = True

import torch

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(os.path.dirname(HERE), "tests", "golden")

from oracle import synth  # noqa: E402

# third-party packages the reference imports for dataset readers / metrics / plots and this image lacks
ABSENT = {"imageio", "cv2", "skimage", "torchvision", "h5py", "lpips", "lmdb", "tensorflow", "prdc", "seaborn", "kornia", "gdown",
          "progressbar"}


class _StandIn(importlib.abc.MetaPathFinder, importlib.abc.Loader):
    def find_spec(self, name, path, target=None):
        if name.split(".")[0] in ABSENT:
            return importlib.machinery.ModuleSpec(name, self, is_package=True)

    def create_module(self, spec):
        m = mock.MagicMock(name=spec.name)
        m.__path__, m.__name__, m.__spec__, m.__loader__ = [], spec.name, spec, self
        return m

    def exec_module(self, module):
        pass


def import_real_runner():
    if not any(isinstance(f, _StandIn) for f in sys.meta_path):
        sys.meta_path.insert(0, _StandIn())
    if REF not in sys.path:
        sys.path.insert(0, REF)
    for attempt in range(20):
        try:
            import runners.ncsn_runner as R
            return R
        except ModuleNotFoundError as e:                    # another absent third-party name: stand-in, retry from a clean slate
            ABSENT.add(e.name.split(".")[0])
            for k in [k for k in sys.modules if k.split(".")[0] in ("runners", "datasets", "models", "evaluation", "losses")]:
                del sys.modules[k]
    raise RuntimeError("could not import runners.ncsn_runner")


class _Cut(Exception):
    pass


def runner_config(name, batch, nfp, subsample, version="DDPM"):
    """The hot-path config of oracle/synth.py plus the fields NCSNRunner.video_gen reads on its way to :1569."""
    config = synth.make_config(name)
    config.device = torch.device("cpu")
    config.model.version = version
    config.model.ema = False
    config.data.dataset = "StochasticMovingMNIST"
    config.data.num_workers = 0
    config.data.prob_mask_cond = 0.0
    config.data.prob_mask_future = 0.0
    config.data.prob_mask_sync = False
    s = config.sampling
    s.batch_size, s.max_data_iter, s.preds_per_test, s.ckpt_id = batch, 1, 1, 0
    s.data_init, s.ssim, s.fvd, s.train = False, True, False, False
    s.num_frames_pred, s.subsample = nfp, subsample
    s.consistent = False
    return config


def gen_videogen(name="tiny", batch=3, nfp=5, subsample=10, n_clips=5, tag="tiny_runner_videogen", overrides=None, version="DDPM"):
    """overrides: `sampling.*` switches of the block loop (init_prev_t: blocks restart from the previous block's frames and the sampler re-noises
    them, :1513 + models/__init__.py:269-280; one_frame_at_a_time: one frame kept per block, :1501-1504, :1530-1531)."""
    R = import_real_runner()
    import models as ref_models
    config = runner_config(name, batch, nfp, subsample, version=version)          # model.version picks the sampler (get_sampler :2702-2714)
    for k_, v_ in (overrides or {}).items():
        setattr(config.sampling, k_, v_)
    C, nf, nc, S = config.data.channels, config.data.num_frames, config.data.num_frames_cond, config.data.image_size
    T = nc + nfp
    g = torch.Generator().manual_seed(31)
    clips = torch.rand(n_clips, T, C, S, S, generator=g)                      # [0, 1] frames, as the dataset readers yield
    ds = torch.utils.data.TensorDataset(clips, torch.zeros(n_clips))
    tmp = tempfile.mkdtemp(prefix="mcvd_runner_")
    args = argparse.Namespace(log_path=tmp, data_path=tmp, start_at=0, image_folder=tmp, video_folder=tmp)

    # ---- the REAL factory + the REAL runner object + the REAL sampler binding
    net = R.get_model(config)                                                 # :180-195 -> UNetMore_DDPM(config).to(device)
    net.load_state_dict(synth.make_state_dict(config, seed=123), strict=False)
    net.eval()
    runner = R.NCSNRunner(args, config, None)
    bound = runner.get_sampler()                                              # :2702-2714
    sampler_name = {"DDPM": "ddpm_sampler", "DDIM": "ddim_sampler", "FPNDM": "FPNDM_sampler"}[version]
    assert bound.func is getattr(ref_models, sampler_name) and bound.keywords == {"config": config}

    n_blocks = nfp if getattr(config.sampling, "one_frame_at_a_time", False) else -(-nfp // nf)
    t_min = getattr(config.sampling, "init_prev_t", -1)
    # draws of one ddpm_sampler call: L - 1 step draws of the executed steps but the last, + 1 for the t_min re-noise (:272-279).  With
    # subsample 10 the t_min test `step < t_min * 10` (:269) skips step 0 only (steps are 0, 100, ...): 8 + 1 = 9 draws, as without t_min.
    per_call = (subsample - 1) if t_min <= 0 else (subsample - 2) + 1
    if version != "DDPM":
        per_call = 0                                                          # DDIM / F-PNDM are deterministic (DDIM draws for the t_min re-noise only)
        assert t_min <= 0
    step_noise = torch.randn(n_blocks, subsample + 1, batch, C * nf, S, S, generator=torch.Generator().manual_seed(77))
    rec = dict(z=[], sampler_kwargs=[], real_t=None, cf=None)
    k = [0]

    def randn_like(like, *a, **kw):                                           # the sampler's step draws, in call order across the blocks
        assert per_call > 0, "a deterministic sampler drew noise"
        blk, i = divmod(k[0], per_call)
        k[0] += 1
        z = step_noise[blk, i].to(like)
        assert z.shape == like.shape
        return z

    real_randn = torch.randn

    def randn(*a, **kw):
        z = real_randn(*a, **kw)
        if z.dim() == 4 and tuple(z.shape) == (batch, C * nf, S, S):          # the block inits, :1476 / :1551
            rec["z"].append(z.clone())
        return z

    real_dt, real_cf, real_idt = R.data_transform, R.conditioning_fn, R.inverse_data_transform
    calls = [0]

    def data_transform(cfg, X):
        out = real_dt(cfg, X)
        rec["real_t"] = out.clone()
        return out

    def conditioning_fn(*a, **kw):
        out = real_cf(*a, **kw)
        rec["cf"] = tuple(None if t is None else t.clone() for t in out)
        return out

    def inverse_data_transform(cfg, X):
        calls[0] += 1
        out = real_idt(cfg, X)
        if calls[0] == 1:
            rec["real01"] = out.clone()
        elif calls[0] == 2:
            rec["cond01"] = out.clone()
        else:                                                                 # :1570: `pred` is complete -- everything after is metrics / plots
            rec["pred_raw"], rec["pred01"] = X.clone(), out.clone()
            raise _Cut()
        return out

    sampler_calls = []
    real_sampler = getattr(ref_models, sampler_name)

    def spy_sampler(x_mod, scorenet, **kw):
        sampler_calls.append({kk: (vv if not torch.is_tensor(vv) else "tensor") for kk, vv in kw.items() if kk not in ("cond", "config")})
        return real_sampler(x_mod, scorenet, **kw)

    import contextlib
    import io
    printed = io.StringIO()
    torch.manual_seed(1234)                                                   # the DataLoader's shuffle and the block inits
    with contextlib.redirect_stdout(printed), mock.patch.object(R, "get_dataset", lambda *a, **kw: (ds, ds)), \
            mock.patch.object(R.eval_models, "PerceptualLoss", mock.MagicMock()), \
            mock.patch.object(R, "data_transform", data_transform), \
            mock.patch.object(R, "conditioning_fn", conditioning_fn), \
            mock.patch.object(R, "inverse_data_transform", inverse_data_transform), \
            mock.patch.object(R, sampler_name, spy_sampler), \
            mock.patch.object(torch, "randn", randn), \
            mock.patch.object(torch, "randn_like", randn_like):
        try:
            runner.video_gen(scorenet=net, ckpt=0, train=False)
            raise RuntimeError("video_gen returned before :1570")
        except _Cut:
            pass
    assert len(rec["z"]) == n_blocks and len(sampler_calls) == n_blocks and k[0] == n_blocks * per_call, (len(rec["z"]), k[0])
    real_t = rec["real_t"]
    # which dataset rows the shuffling loader served (clips are distinct): row r of the batch is clip order[r]
    order = [int(((real_dt(config, clips) - real_t[r]).flatten(1).abs().max(dim=1).values).argmin()) for r in range(batch)]
    real, cond, cond_mask = rec["cf"]
    log_lines = [ln for ln in printed.getvalue().splitlines() if ln.startswith(version + ": ")]      # the sampler's `verbose` lines (:304-306, :181-183)
    assert len(log_lines) == (0 if version == "FPNDM" else n_blocks * (10 if t_min <= 0 else 9)), len(log_lines)
    # the same chain in float64 on the restatement (same inits, same noise): the noise floor a tolerance on pred_raw stands on
    from oracle import sampler_ref, unet_ref
    net64 = unet_ref.OracleScoreNet(config, synth.make_state_dict(config, seed=123), dtype=torch.float64)
    cond64, preds64 = cond.double(), []
    one_at = bool(getattr(config.sampling, "one_frame_at_a_time", False))
    g64 = None
    for b in range(n_blocks):
        kk = [0]

        def fn(i, like, b=b, kk=kk):
            kk[0] += 1
            return step_noise[b, kk[0] - 1].to(like.dtype)
        x_in = rec["z"][b].double() if (b == 0 or t_min <= 0) else g64
        if version == "FPNDM":
            g64 = sampler_ref.fpndm_sample(x_in, net64, cond=cond64, final_only=True, subsample_steps=subsample,
                                           clip_before=bool(getattr(config.sampling, "clip_before", True)))[-1]
        else:
            g64 = sampler_ref.sample(x_in, net64, cond=cond64, kind=version.lower(), final_only=True, denoise=bool(getattr(config.sampling, "denoise", True)),
                                     subsample_steps=subsample, clip_before=bool(getattr(config.sampling, "clip_before", True)), noise_fn=fn, t_min=t_min)[-1]
        preds64.append(g64)
        if b != n_blocks - 1:
            cond64 = torch.cat([cond64[:, C:], g64[:, :C]], dim=1) if one_at else \
                torch.cat([cond64[:, C * nf:], g64[:, C * max(0, nf - nc):]], dim=1)
    drift = float((rec["pred_raw"].double() - torch.cat(preds64, dim=1)[:, :C * nfp]).abs().max())
    print(f"  reference fp32 vs fp64 restatement of the {n_blocks}-block chain: {drift:.3e}")
    out = dict(version=version, log_lines=log_lines, ref32_vs_ref64_max_abs=drift, overrides=dict(overrides or {}), config_name=name, batch=batch, nfp=nfp, subsample=subsample, clips=clips, order=order, real_t=real_t, real=real, cond=cond,
               cond_mask=cond_mask, z_init=torch.stack(rec["z"]), step_noise=step_noise, pred_raw=rec["pred_raw"], pred01=rec["pred01"],
               real01=rec["real01"], cond01=rec["cond01"], sampler_kwargs=sampler_calls,
               stood_in=sorted(ABSENT))
    os.makedirs(OUT, exist_ok=True)
    torch.save(out, os.path.join(OUT, f"{tag}.pt"))
    sys.stdout.write(f"wrote {tag}.pt: {n_blocks} blocks, pred {tuple(rec['pred_raw'].shape)}, range [{rec['pred_raw'].min():.3f}, {rec['pred_raw'].max():.3f}], "
          f"rows served {order}, sampler kwargs {sampler_calls[0]}\n")


if __name__ == "__main__":
    torch.set_num_threads(2)
    which = sys.argv[1:] or ["default", "prevt", "oneframe"]
    if "default" in which:
        gen_videogen()
    if "prevt" in which:
        gen_videogen(tag="tiny_runner_videogen_prevt", overrides=dict(init_prev_t=0.5))
    if "plain" in which:      # the config switches the block loop forwards to the sampler (:1516-1518): no final denoise forward, no clip of x0
        gen_videogen(tag="tiny_runner_videogen_plain", overrides=dict(denoise=False, clip_before=False))
    if "ddim" in which:       # model.version = "DDIM": get_sampler binds ddim_sampler, the same block loop
        gen_videogen(tag="tiny_runner_videogen_ddim", version="DDIM")
    if "fpndm" in which:
        gen_videogen(tag="tiny_runner_videogen_fpndm", version="FPNDM")
    if "oneframe" in which:
        gen_videogen(nfp=3, tag="tiny_runner_videogen_oneframe", overrides=dict(one_frame_at_a_time=True))

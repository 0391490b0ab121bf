"""HipScoreNet: drop-in for the reference's `UNetMore_DDPM` score network on the sampling path.

It offers exactly what the reference's samplers and runner touch:
  * `scorenet(x, labels, cond=cond)` -> eps, same shape/device as x     (models/__init__.py:284)
  * attributes `alphas`, `alphas_prev`, `betas` (1-D device tensors), `type`   (:221, :226)
  * `load_state_dict(sd, strict=False)` with reference key names, `module.`-prefixed or bare
    (runners/ncsn_runner.py:923-932), `eval()`, `named_parameters()`/`parameters()` so that
    `EMAHelper.register/ema` (models/ema.py:10-29) work unchanged, `to(device)`.
All arithmetic happens in libmcvd_hip.so (hand-written gfx950 kernels) through the C ABI; torch tensors are
storage only.  There is no CPU / eager fallback: a non-GPU device or a missing library raises.
"""
import ctypes as C
import math
from collections import OrderedDict, namedtuple

import numpy as np
import torch

from . import _lib
from .config import desc_from_config

_IncompatibleKeys = namedtuple("IncompatibleKeys", ["missing_keys", "unexpected_keys"])
_BUFFER_KEYS = ("betas", "alphas", "alphas_prev", "unet.sigmas", "k", "k_cum", "theta_t")


def _get_sigmas(config):
    """Noise schedule (reference: models/__init__.py:16-35), CPU fp32."""
    T = getattr(config.model, "num_classes")
    dist = getattr(config.model, "sigma_dist", "linear")
    if dist == "linear":
        return torch.linspace(config.model.sigma_begin, config.model.sigma_end, T)
    if dist == "cosine":
        t = torch.linspace(T, 0, T + 1) / T
        s = 0.008
        f = torch.cos((t + s) / (1 + s) * np.pi / 2) ** 2
        return f[:-1] / f[-1]
    raise NotImplementedError("sigma distribution not supported")


def _schedule(config):
    """betas / alphas / alphas_prev as UNetMore_DDPM registers them (ncsnpp_more.py:735-743)."""
    if getattr(config.model, "sigma_dist", "linear") == "linear":
        betas = _get_sigmas(config)
        alphas = torch.cumprod(1 - betas.flip(0), 0).flip(0)
        alphas_prev = torch.cat([alphas[1:], torch.tensor([1.0]).to(alphas)])
    else:
        alphas = _get_sigmas(config)
        alphas_prev = torch.cat([alphas[1:], torch.tensor([1.0]).to(alphas)])
        betas = 1 - alphas / alphas_prev
    return betas.float().contiguous(), alphas.float().contiguous(), alphas_prev.float().contiguous()


def _fptr(t):
    return C.c_void_p(t.data_ptr())


class HipScoreNet:
    def __init__(self, config, device=None, plan_only=False):
        """`plan_only=True` builds the parameter table / schedule without a GPU (host-side protocol tests: state_dict names,
        `EMAHelper` round trip); such an object cannot compute -- every forward / upload raises."""
        self.config = config
        self.plan_only = bool(plan_only)
        self.version = getattr(config.model, "version", "DDPM").upper()
        self.type = getattr(config.model, "type", None) if isinstance(getattr(config.model, "type", None), str) else None
        self.training = False
        self._desc = desc_from_config(config)
        self._ctx = C.c_void_p()
        self._model = C.c_void_p()
        if self.plan_only:
            self.device = torch.device("cpu")
            _lib.check(_lib.lib.mcvd_model_create(None, C.byref(self._desc), C.byref(self._model)), "model_create")
        else:
            if not torch.cuda.is_available():
                raise RuntimeError("HipScoreNet needs a ROCm GPU (MI355X); there is no CPU fallback")
            dev = torch.device(device if device is not None else getattr(config, "device", "cuda:0"))
            if dev.type != "cuda":
                raise RuntimeError(f"HipScoreNet cannot run on device {dev}")
            self.device = torch.device("cuda", dev.index if dev.index is not None else torch.cuda.current_device())
            with torch.cuda.device(self.device):
                stream = torch.cuda.current_stream(self.device).cuda_stream
                _lib.check(_lib.lib.mcvd_ctx_create(self.device.index, C.c_void_p(stream), C.byref(self._ctx)), "ctx_create")
                _lib.check(_lib.lib.mcvd_model_create(self._ctx, C.byref(self._desc), C.byref(self._model)), "model_create")
        # python-visible parameters (torch storage; uploaded to the library's blob by sync_parameters).  requires_grad=True
        # although no autograd ever runs: the reference's EMAHelper.register / .ema only touch parameters that require grad
        # (models/ema.py:12-13, 26-28), and the documented drop-in path goes through it (runners/ncsn_runner.py:928-932).
        self._params = OrderedDict()
        n = _lib.lib.mcvd_model_num_params(self._model)
        name, shape, ndim, off = C.c_char_p(), (C.c_int64 * 4)(), C.c_int(), C.c_int64()
        for i in range(n):
            _lib.check(_lib.lib.mcvd_model_param_info(self._model, i, C.byref(name), shape, C.byref(ndim), C.byref(off)))
            shp = tuple(shape[k] for k in range(ndim.value))
            self._params[name.value.decode()] = torch.nn.Parameter(
                torch.zeros(shp, dtype=torch.float32, device=self.device), requires_grad=True)
        self._loaded = False
        self._dirty = True
        self._cond_z = None
        self._cond_key = None          # (data_ptr, _version, B) of the cond whose SPADE maps are cached in the library
        # schedule buffers + sinusoid table, computed exactly like the reference and handed to the library
        betas, alphas, alphas_prev = _schedule(config)
        self._set_schedule(betas, alphas, alphas_prev)
        self.noise_in_cond = bool(getattr(config.model, "noise_in_cond", False))            # ncsnpp_more.py:751
        self.output_all_frames = bool(getattr(config.model, "output_all_frames", False))
        self.gamma = bool(getattr(config.model, "gamma", False))                            # ncsnpp_more.py:744-749
        if self.gamma:
            # computed on the CPU like the schedule itself: g ~ k theta +- sqrt(k) theta with k up to 1e9 is standardised by a
            # catastrophically cancelling (g - k theta), so the tables must be the reference's bit for bit (a device cumsum is not)
            self.theta_0 = 0.001
            k = betas / (alphas * (self.theta_0 ** 2))
            kc = torch.cumsum(k.flip(0), 0).flip(0).float().contiguous()
            th = (torch.sqrt(alphas) * self.theta_0).float().contiguous()
            self.k, self.k_cum, self.theta_t = k.to(self.device), kc.to(self.device), th.to(self.device)
            _lib.check(_lib.lib.mcvd_model_set_gamma_tables(self._model, _fptr(kc), _fptr(th), self._desc.num_classes), "set_gamma_tables")
        half = self._desc.ngf // 2
        emb = math.log(10000) / (half - 1)                                   # layers.py:505-510
        freqs = torch.exp(torch.arange(half, dtype=torch.float32) * -emb).contiguous()
        _lib.check(_lib.lib.mcvd_model_set_temb_freqs(self._model, _fptr(freqs), half), "set_temb_freqs")


    # ---------------------------------------------------------------- nn.Module-ish surface
    def _set_schedule(self, betas, alphas, alphas_prev):
        T = self._desc.num_classes
        b, a, ap = (t.detach().float().cpu().contiguous() for t in (betas, alphas, alphas_prev))
        assert b.numel() == T and a.numel() == T and ap.numel() == T
        _lib.check(_lib.lib.mcvd_model_set_schedule(self._model, _fptr(b), _fptr(a), _fptr(ap), T), "set_schedule")
        self.betas, self.alphas, self.alphas_prev = b.to(self.device), a.to(self.device), ap.to(self.device)

    def named_parameters(self, prefix="", recurse=True):
        for k, v in self._params.items():
            yield (prefix + k, v)

    def parameters(self, recurse=True):
        return iter(self._params.values())

    def state_dict(self):
        sd = OrderedDict((k, v.data) for k, v in self._params.items())
        sd["betas"], sd["alphas"], sd["alphas_prev"] = self.betas, self.alphas, self.alphas_prev
        if self.gamma:
            sd["k"], sd["k_cum"], sd["theta_t"] = self.k, self.k_cum, self.theta_t
        return sd

    def load_state_dict(self, state_dict, strict=True):
        """Reference key names; a leading 'module.' (DataParallel) is ignored."""
        seen = set()
        unexpected = []
        sched = {}
        for k, v in state_dict.items():
            key = k[7:] if k.startswith("module.") else k
            if key in self._params:
                p = self._params[key]
                if tuple(v.shape) != tuple(p.shape):
                    raise RuntimeError(f"size mismatch for {key}: {tuple(v.shape)} vs {tuple(p.shape)}")
                with torch.no_grad():
                    p.data.copy_(v.detach().to(dtype=torch.float32))
                seen.add(key)
            elif key in _BUFFER_KEYS:
                sched[key] = v
            else:
                unexpected.append(k)
        if all(k in sched for k in ("betas", "alphas", "alphas_prev")):
            self._set_schedule(sched["betas"], sched["alphas"], sched["alphas_prev"])
        if self.gamma and all(k in sched for k in ("k_cum", "theta_t")):          # buffers of a model.gamma checkpoint (ncsnpp_more.py:747-749)
            self.k_cum, self.theta_t = sched["k_cum"].float().to(self.device), sched["theta_t"].float().to(self.device)
            if "k" in sched:
                self.k = sched["k"].float().to(self.device)
            kc, th = self.k_cum.cpu().contiguous(), self.theta_t.cpu().contiguous()
            _lib.check(_lib.lib.mcvd_model_set_gamma_tables(self._model, _fptr(kc), _fptr(th), self._desc.num_classes), "set_gamma_tables")
        missing = [k for k in self._params if k not in seen]
        if strict and (missing or unexpected):
            raise RuntimeError(f"load_state_dict: missing {missing[:5]}... unexpected {unexpected[:5]}...")
        if not missing:
            self._loaded = True
        self._dirty = True
        return _IncompatibleKeys(missing, unexpected)

    @torch.no_grad()
    def reset_parameters(self, generator=None):
        """The reference's construction-time initialisation, so that `get_model(config)` without a checkpoint holds what
        `UNetMore_DDPM(config)` holds (runners/ncsn_runner.py:180-195 hands the fresh module straight to the trainer /
        sampler).  Same distributions per tensor, not the same draws (the reference draws in module-construction order from the
        global generator; `generator` here is an optional CPU torch.Generator):
          * `default_init(scale)` = variance_scaling(scale, 'fan_avg', 'uniform'), scale 0 -> 1e-10 (models/better/layers.py:43-80):
            U(+-sqrt(3 scale / fan_avg)), fan = shape[1 | 0] * receptive field -- scale 1 for the two time-MLP Linears
            (ncsnpp_more.py:89-95), every Dense_0 (layerspp.py:505-507), the stem, Conv_0 and the 1x1 Conv_2 (layers.py:89-113);
            scale 0 (init_scale = 0., ncsnpp_more.py:67) for every Conv_1, NIN_3 and the last conv (layerspp.py:586, :219;
            ncsnpp_more.py:247, :586); scale 0.1 for NIN_0..2 (layers.py:536);
          * every bias and NIN.b zero, GroupNorm / final-norm gains 1 (torch defaults, layerspp.py:215, :477);
          * SPADE mlp_shared / mlp_gamma / mlp_beta are `ddpm_conv3x3` too (MySPADE is given `conv=conv3x3`, layerspp.py:103, :148-150;
            ncsnpp_more.py:433-440): scale 1, zero bias;
          * cond_emb: nn.Embedding default N(0, 1) (ncsnpp_more.py:98)."""
        names = list(self._params.keys())
        last_conv = max((int(k.split(".")[2]) for k in names if self._params[k].dim() == 4), default=-1)
        for k, p in self._params.items():
            shape = tuple(p.shape)
            mod = int(k.split(".")[2])
            if p.dim() == 1:
                if k.endswith(".weight"):                                # GroupNorm_0.weight / Norm_0.weight
                    t = torch.ones(shape)
                else:                                                      # .bias, NIN .b
                    t = torch.zeros(shape)
            elif self._desc.cond_emb and p.dim() == 2 and mod == 2 and k.endswith(".weight"):
                t = torch.randn(shape, generator=generator)                # nn.Embedding(2, nf // 2)
            else:
                if k.endswith(".W"):
                    scale = 1e-10 if ".NIN_3." in k else 0.1
                elif ".Conv_1." in k or (p.dim() == 4 and mod == last_conv):
                    scale = 1e-10
                else:
                    scale = 1.0
                rf = 1
                for s_ in shape[2:]:
                    rf *= s_
                fan_avg = (shape[0] + shape[1]) * rf / 2.0
                t = (torch.rand(shape, generator=generator) * 2 - 1) * math.sqrt(3.0 * scale / fan_avg)
            p.data.copy_(t.to(dtype=torch.float32))
        self._loaded = True
        self._dirty = True
        return self

    def mark_dirty(self):
        """Call after modifying parameter tensors in place (EMAHelper.ema does this through .data.copy_)."""
        self._dirty = True

    def sync_parameters(self, force=False):
        """Upload the python-visible parameters into the library blob and repack (~1 ms).  The samplers call
        this (forced) at the start of every sampling call, because in-place edits through `.data` cannot be observed."""
        if not (self._dirty or force):
            return
        if self.plan_only:
            raise RuntimeError("HipScoreNet(plan_only=True) has no device: it cannot upload parameters or compute")
        if not self._loaded:
            raise RuntimeError("HipScoreNet: parameters were never loaded (load_state_dict first)")
        with torch.cuda.device(self.device):
            self._bind_stream()
            shape = (C.c_int64 * 4)()
            for k, p in self._params.items():
                for i, s in enumerate(p.shape):
                    shape[i] = s
                _lib.check(_lib.lib.mcvd_model_set_param(self._model, k.encode(), _fptr(p.data), shape, p.dim(), 1),
                           f"set_param({k})")
            _lib.check(_lib.lib.mcvd_model_finalize(self._model), "finalize")
            self._report_selftest()
            _lib.check(_lib.lib.mcvd_model_invalidate_cond(self._model))
        self._cond_key = None
        self._dirty = False

    def _report_selftest(self):
        """mcvd_model_finalize runs the library's self-test of its hand-scheduled Winograd kernels once per context (include/mcvd_hip.h:
        mcvd_ctx_selftest).  A failure is not fatal -- the library falls back to its fp32-MFMA kernels -- but it must not be silent."""
        rc = _lib.lib.mcvd_ctx_selftest(self._ctx)
        if rc == -6:               # MCVD_ESELFTEST: the test RAN and the kernels disagreed -> the library switched bf16x3 off
            import warnings
            warnings.warn("mcvd_hip: " + _lib.last_error() + " (results stay within the fp32 contract; throughput is lower)", RuntimeWarning)
        elif rc < 0:               # the test could not run (allocation / launch error): nothing was switched off, nothing was verified
            raise RuntimeError("mcvd_hip: the Winograd self-test could not run: " + _lib.last_error())

    def eval(self):
        self.training = False
        return self

    def train(self, mode=True):
        if mode:
            raise NotImplementedError("HipScoreNet is inference-only (dropout must be inactive, SURVEY 9.6-9)")
        return self

    def to(self, device=None, *args, **kwargs):
        if device is not None and torch.device(device).type == "cuda" and not self.plan_only:
            d = torch.device(device)
            if d.index is None or d.index == self.device.index:
                return self
        raise RuntimeError(f"HipScoreNet is bound to {self.device}; cannot move to {device}")

    def set_option(self, key, value):
        _lib.check(_lib.lib.mcvd_ctx_set_option(self._ctx, key.encode(), int(value)), f"set_option({key})")

    # ---------------------------------------------------------------- kernel-selection table (autotuner output)
    def get_tuning(self, B):
        """[(shape id, cout tile)] per plan op for batch size B (after a forward at B)."""
        n = _lib.lib.mcvd_model_get_tuning(self._model, int(B), None, None, 0)
        if n < 0:
            raise RuntimeError(f"get_tuning: {_lib.last_error()}")
        sh, ct = (C.c_int * n)(), (C.c_int * n)()
        if _lib.lib.mcvd_model_get_tuning(self._model, int(B), sh, ct, n) != n:
            raise RuntimeError(f"get_tuning: {_lib.last_error()}")
        return [(sh[i], ct[i]) for i in range(n)]

    def set_tuning(self, B, table):
        n = len(table)
        sh, ct = (C.c_int * n)(*[int(t[0]) for t in table]), (C.c_int * n)(*[int(t[1]) for t in table])
        _lib.check(_lib.lib.mcvd_model_set_tuning(self._model, int(B), sh, ct, n), "set_tuning")

    def save_tuning(self, path, batch_sizes):
        import json
        with open(path, "w") as f:
            json.dump({str(int(b)): self.get_tuning(b) for b in batch_sizes}, f)

    def load_tuning(self, path):
        import json
        with open(path) as f:
            for b, table in json.load(f).items():
                self.set_tuning(int(b), table)

    # ---------------------------------------------------------------- blob (one-shot weight broadcast)
    def blob_numel(self):
        n = C.c_int64()
        _lib.check(_lib.lib.mcvd_model_blob_floats(self._model, C.byref(n)))
        return n.value

    def export_blob(self):
        self.sync_parameters()
        t = torch.empty(self.blob_numel(), dtype=torch.float32, device=self.device)
        _lib.check(_lib.lib.mcvd_model_export_blob(self._model, _fptr(t)), "export_blob")
        return t

    def import_blob(self, blob):
        assert blob.is_cuda and blob.dtype == torch.float32 and blob.numel() == self.blob_numel()
        with torch.cuda.device(self.device):
            self._bind_stream()
            _lib.check(_lib.lib.mcvd_model_import_blob(self._model, _fptr(blob.contiguous())), "import_blob")
            _lib.check(_lib.lib.mcvd_model_finalize(self._model), "finalize")
            self._report_selftest()
            # keep the python-visible copies coherent
            name, shape, ndim, off = C.c_char_p(), (C.c_int64 * 4)(), C.c_int(), C.c_int64()
            for i, (k, p) in enumerate(self._params.items()):
                _lib.check(_lib.lib.mcvd_model_param_info(self._model, i, C.byref(name), shape, C.byref(ndim), C.byref(off)))
                p.data.copy_(blob[off.value:off.value + p.numel()].view_as(p))
        self._loaded = True
        self._dirty = False

    # ---------------------------------------------------------------- forward
    def _bind_stream(self):
        _lib.check(_lib.lib.mcvd_ctx_set_stream(self._ctx, C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)))

    def _prep(self, t, name):
        if t is None:
            return None
        if not t.is_cuda or t.device != self.device:
            raise RuntimeError(f"{name} must live on {self.device} (got {t.device})")
        return t.to(dtype=torch.float32).contiguous()

    @torch.no_grad()
    def __call__(self, x, y, cond=None, cond_mask=None):
        """eps = UNet(x, y, cond)   (reference: UNetMore_DDPM.forward, ncsnpp_more.py:753-770)."""
        if self.plan_only:
            raise RuntimeError("HipScoreNet(plan_only=True) cannot compute (no GPU context)")
        self.sync_parameters()
        x = self._prep(x, "x")
        cond = self._prep(cond, "cond")
        d = self._desc
        B = x.shape[0]
        if tuple(x.shape[1:]) != (d.channels * d.num_frames, d.image_size, d.image_size):
            raise RuntimeError(f"x has shape {tuple(x.shape)}")
        if d.num_frames_cond > 0:
            if cond is None or tuple(cond.shape) != (B, d.channels * d.num_frames_cond, d.image_size, d.image_size):
                raise RuntimeError("cond missing or mis-shaped")
        elif cond is not None:      # the reference concatenates whatever it is given and fails in the stem conv (ncsnpp_more.py:256-257)
            raise RuntimeError("this model takes no conditioning frames (num_frames_cond == 0) but cond was passed")
        # integer labels as the DDPM/DDIM samplers pass them; float (possibly fractional) timesteps for F-PNDM's midpoints:
        # the reference's embedding takes timesteps.float() either way (layers.py:504-518)
        float_t = y.is_floating_point()
        y = y.to(device=self.device, dtype=torch.float32 if float_t else torch.int64).contiguous()
        if y.shape != (B,):
            raise RuntimeError(f"labels have shape {tuple(y.shape)}")
        if self.output_all_frames and cond is not None:
            # ncsnpp_more.py:384-385: torch.split(h, [C*nc, C*nf]) on the C*nf-channel output of conv3x3_last cannot succeed
            raise RuntimeError(f"split_with_sizes expects split_sizes to sum exactly to {d.channels * d.num_frames} (input tensor's size "
                               f"at dimension 1), but got split_sizes=[{d.channels * d.num_frames_cond}, {d.channels * d.num_frames}] "
                               "-- model.output_all_frames is broken for arch 'unetmore' in the reference as well")
        mask = None
        if self._desc.cond_emb and cond_mask is not None:
            mask = cond_mask.to(device=self.device, dtype=torch.int32).contiguous()
            if mask.shape != (B,):
                raise RuntimeError(f"cond_mask has shape {tuple(mask.shape)}")
        out = torch.empty_like(x)
        with torch.cuda.device(self.device):
            self._bind_stream()
            if self.noise_in_cond and cond is not None:                        # ncsnpp_more.py:755-768: fresh noise every forward
                if float_t:
                    raise IndexError("tensors used as indices must be long, int, byte or bool tensors (noise_in_cond indexes alphas "
                                     "with the labels)")
                z = self._cond_z
                self._cond_z = None
                if z is None:
                    if self.gamma:                                              # :761-765, torch's device gamma sampler like the reference
                        ua = self.alphas[y].reshape(B, 1, 1, 1)
                        uk = self.k_cum[y].reshape(B, 1, 1, 1).repeat(1, *cond.shape[1:])
                        ut = self.theta_t[y].reshape(B, 1, 1, 1).repeat(1, *cond.shape[1:])
                        z = torch.distributions.gamma.Gamma(uk, 1 / ut).sample()
                        z = (z - uk * ut) / (1 - ua).sqrt()
                    else:
                        z = torch.randn_like(cond)
                z = z.to(device=self.device, dtype=torch.float32).contiguous()
                if z.shape != cond.shape:
                    raise RuntimeError(f"conditioning noise has shape {tuple(z.shape)}, cond {tuple(cond.shape)}")
                _lib.check(_lib.lib.mcvd_model_set_cond_noise(self._model, _fptr(z), 0, 0, 0), "set_cond_noise")
            if self._desc.spade and not self.noise_in_cond:
                # SPADE gamma/beta depend only on cond: recompute only when the tensor (or its content version) changes
                key = (cond.data_ptr(), cond._version, B)
                if key != self._cond_key:
                    _lib.check(_lib.lib.mcvd_model_prepare_cond(self._model, _fptr(cond), B), "prepare_cond")
                    self._cond_key = key
                    self._cond_keepalive = cond
            cptr = _fptr(cond) if cond is not None else None
            if mask is not None and not float_t:
                rc = _lib.lib.mcvd_unet_forward_masked(self._model, _fptr(x), C.c_void_p(y.data_ptr()), cptr, C.c_void_p(mask.data_ptr()),
                                                       _fptr(out), B)
            elif mask is not None:
                raise RuntimeError("cond_mask with float timesteps is not supported")
            else:
                fwd = _lib.lib.mcvd_unet_forward_ft if float_t else _lib.lib.mcvd_unet_forward
                rc = fwd(self._model, _fptr(x), C.c_void_p(y.data_ptr()), cptr, _fptr(out), B)
            if self.noise_in_cond:
                _lib.lib.mcvd_model_set_cond_noise(self._model, None, 0, 0, 0)          # never leave a pointer to a dead tensor behind
            _lib.check(rc, "unet_forward")
        return out

    def set_next_cond_noise(self, z):
        """noise_in_cond: use `z` (shaped like cond) instead of a fresh device draw in the NEXT forward (parity runs)."""
        self._cond_z = z

    forward = __call__

    def num_launches(self, B=1):
        return _lib.lib.mcvd_model_num_launches(self._model, B)

    def __del__(self):
        try:
            if getattr(self, "_model", None):
                _lib.lib.mcvd_model_destroy(self._model)
                self._model = None
            if getattr(self, "_ctx", None):
                _lib.lib.mcvd_ctx_destroy(self._ctx)
                self._ctx = None
        except Exception:
            pass


def get_model(config):
    """Factory mirroring runners/ncsn_runner.py:180-195 for arch == 'unetmore': like `UNetMore_DDPM(config).to(config.device)`, the
    returned net carries the reference's construction-time initialisation (`reset_parameters`) until a checkpoint is loaded."""
    return HipScoreNet(config, getattr(config, "device", None)).reset_parameters()

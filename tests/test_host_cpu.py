"""CPU (no GPU): the C-ABI library loads, exports every symbol of include/mcvd_hip.h, fails loudly without a
device, and its host-side plan (parameter table, schedule) agrees with the reference-pinned oracle."""
import ctypes as C
import os
import re

import pytest
import torch

from oracle import synth, unet_ref

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    from mcvd_pytorch_amd import _lib
    hdr = open(os.path.join(ROOT, "include", "mcvd_hip.h")).read()
    declared = set(re.findall(r"\b(mcvd_[a-z0-9_]+)\s*\(", hdr))
    declared -= {"mcvd_ctx", "mcvd_model", "mcvd_unet_desc"}
    assert declared, "no declarations parsed"
    raw = C.CDLL(_lib.LIB_PATH)
    for name in sorted(declared):
        assert hasattr(raw, name), f"{name} declared in mcvd_hip.h but not exported"
    assert declared == set(_lib.EXPORTS), declared ^ set(_lib.EXPORTS)


def test_no_gpu_fails_loudly():
    from mcvd_pytorch_amd import _lib
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    ctx = C.c_void_p()
    rc = _lib.lib.mcvd_ctx_create(0, None, C.byref(ctx))
    assert rc != 0 and "device" in _lib.last_error().lower()
    from mcvd_pytorch_amd.scorenet import HipScoreNet
    with pytest.raises(RuntimeError):
        HipScoreNet(synth.make_config("tiny"))


def test_samplers_refuse_foreign_scorenet():
    from mcvd_pytorch_amd.samplers import ddpm_sampler
    with pytest.raises(TypeError):
        ddpm_sampler(torch.zeros(1, 2, 32, 32), lambda x, y, cond=None: x)


def _plan(name):
    from mcvd_pytorch_amd import _lib
    from mcvd_pytorch_amd.config import desc_from_config
    config = synth.make_config(name)
    desc = desc_from_config(config)
    m = C.c_void_p()
    _lib.check(_lib.lib.mcvd_model_create(None, C.byref(desc), C.byref(m)), "model_create(plan only)")
    return _lib, config, m


@pytest.mark.parametrize("name", ["tiny", "tiny_spade", "smmnist_big5", "smmnist_big5_ngf96", "kth64_big_ngf128",
                                  "bair_big_spade", "cityscapes_big", "cityscapes_big_variant"])
def test_plan_parameter_table_matches_reference_names(name):
    """Names, shapes and ORDER equal the reference state_dict (oracle.param_shapes is pinned to it by gen_golden)."""
    _lib, config, m = _plan(name)
    want = unet_ref.param_shapes(unet_ref.hot_cfg(config))
    n = _lib.lib.mcvd_model_num_params(m)
    pname, shape, ndim, off = C.c_char_p(), (C.c_int64 * 4)(), C.c_int(), C.c_int64()
    have = []
    for i in range(n):
        _lib.check(_lib.lib.mcvd_model_param_info(m, i, C.byref(pname), shape, C.byref(ndim), C.byref(off)))
        have.append((pname.value.decode(), tuple(shape[k] for k in range(ndim.value))))
        assert off.value % 4 == 0
    assert have == list(want.items())
    _lib.lib.mcvd_model_destroy(m)


@pytest.mark.parametrize("name", ["tiny", "smmnist_big5"])
def test_library_default_schedule_close_to_torch(name):
    """The C restatement of linspace/cumprod (used when no host overrides it) vs torch: a few ulp at most."""
    _lib, config, m = _plan(name)
    T = config.model.num_classes
    bufs = [torch.empty(T) for _ in range(3)]
    _lib.check(_lib.lib.mcvd_model_get_schedule(m, *[C.c_void_p(b.data_ptr()) for b in bufs], T))
    ref = unet_ref.make_schedule(unet_ref.hot_cfg(config))
    for got, want in zip(bufs, ref):
        torch.testing.assert_close(got, want, rtol=2e-6, atol=0)
    _lib.lib.mcvd_model_destroy(m)


def test_unsupported_configs_raise():
    from mcvd_pytorch_amd.config import desc_from_config
    cfg = synth.make_config("tiny")
    cfg.model.gamma = True
    with pytest.raises(NotImplementedError):
        desc_from_config(cfg)
    cfg = synth.make_config("tiny")
    cfg.model.arch = "unetmore3d"
    with pytest.raises(NotImplementedError):
        desc_from_config(cfg)

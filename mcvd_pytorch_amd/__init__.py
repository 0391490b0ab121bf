"""mcvd_pytorch_amd: MI355X (gfx950) native MCVD sampling hot path.

Python host code keeping the reference's call surface (`scorenet(x, t, cond=)`, `ddpm_sampler`, `ddim_sampler`, the YAML
config schema) on top of hand-written HIP kernels reached through the C ABI of libmcvd_hip.so (include/mcvd_hip.h).
Importing this package loads the shared library and raises if it is missing -- there is no CPU or eager fallback.
"""
from . import _lib  # noqa: F401  (fails loudly when the HIP extension is not built)
from .config import desc_from_config, dict2namespace, load_config  # noqa: F401
from .samplers import ddim_sampler, ddpm_sampler, fpndm_sampler, get_sampler  # noqa: F401
from .scorenet import HipScoreNet, get_model  # noqa: F401
from .checkpoint import load_model, load_states_into  # noqa: F401
from .runner import conditioning_fn, data_transform, frames_to_uint8, inverse_data_transform, save_video_pred, video_gen  # noqa: F401

__all__ = ["HipScoreNet", "get_model", "ddpm_sampler", "ddim_sampler", "fpndm_sampler", "get_sampler", "dict2namespace", "load_config",
           "desc_from_config", "load_model", "load_states_into", "conditioning_fn", "data_transform",
           "inverse_data_transform", "video_gen", "save_video_pred", "frames_to_uint8"]

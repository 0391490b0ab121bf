"""Checkpoint ingestion with the reference's on-disk format (SURVEY section 5 'checkpoint / resume').

`checkpoint.pt` = `[model_state_dict ('module.'-prefixed), optimizer_state, epoch, step, ema_state_dict (bare keys)]`
(runners/ncsn_runner.py:425-433).  Sampling loads `states[0]` and then, if `config.model.ema`, overwrites the parameters
with the EMA shadow `states[-1]` (runners/ncsn_runner.py:926-932, models/ema.py:24-29) -- reproduced here without the
EMAHelper round trip.
"""
import torch

from .scorenet import HipScoreNet


def load_states_into(scorenet, states, use_ema=True):
    """Load `states` (the list stored in checkpoint.pt) into a HipScoreNet.  Returns the (missing, unexpected) report of
    the model state_dict load."""
    if not isinstance(states, (list, tuple)) or len(states) < 1:
        raise ValueError("checkpoint must be the reference's list [model_sd, optim_sd, epoch, step, (ema_sd)]")
    report = scorenet.load_state_dict(states[0], strict=False)
    if use_ema and len(states) >= 5 and isinstance(states[-1], dict):
        shadow = states[-1]
        own = dict(scorenet.named_parameters())
        for name, value in shadow.items():          # EMAHelper.ema: param.data.copy_(shadow[name])
            key = name[7:] if name.startswith("module.") else name
            if key in own:
                own[key].data.copy_(value.to(own[key].device, dtype=torch.float32))
        scorenet.mark_dirty()
    return report


def load_model(ckpt_path, config, device="cuda:0"):
    """Counterpart of runners/ncsn_runner.py:155-177 `load_model` for arch == 'unetmore' (config passed explicitly)."""
    config.device = device
    net = HipScoreNet(config, device)
    states = torch.load(ckpt_path, map_location="cpu", weights_only=False)
    load_states_into(net, states, use_ema=bool(getattr(config.model, "ema", False)))
    return net.eval()

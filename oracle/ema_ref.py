"""CPU ORACLE -- TEST INFRASTRUCTURE ONLY (see oracle/unet_ref.py header).

Restatement of the reference's `EMAHelper` (models/ema.py:4-29, 43-47): the object `NCSNRunner.sample` / `video_gen` push the
sampling weights through (runners/ncsn_runner.py:928-932: register -> load_state_dict(states[-1]) -> ema(scorenet)).  Only the
sampling-side methods are restated; `update` / `ema_copy` belong to training.

Parity status: PINNED by tests/test_host_cpu.py::test_ema_helper_protocol, which runs the real class from /root/reference (when
present: the build container) and this one against the same HipScoreNet parameter table and compares the outcomes.
"""
import torch.nn as nn


class EMAHelper(object):
    def __init__(self, mu=0.999):                     # ema.py:5-7
        self.mu = mu
        self.shadow = {}

    def register(self, module):                        # ema.py:9-14
        if isinstance(module, nn.DataParallel):
            module = module.module
        for name, param in module.named_parameters():
            if param.requires_grad:
                self.shadow[name] = param.data.clone()

    def ema(self, module):                             # ema.py:23-28
        if isinstance(module, nn.DataParallel):
            module = module.module
        for name, param in module.named_parameters():
            if param.requires_grad:
                param.data.copy_(self.shadow[name].data)

    def state_dict(self):                              # ema.py:43-44
        return self.shadow

    def load_state_dict(self, state_dict):             # ema.py:46-47
        self.shadow = state_dict

"""``EmpiricalNormalization`` (upstream ``rsl_rl/modules/normalizer.py`` @ v2.3.1 -- third-party, absent from the reference
tree; PARITY UNPINNED, restated in oracle/rsl_rl_oracle.py).  Enabled by ``empirical_normalization=True``
(reference isaaclab_rl/rsl_rl/rl_cfg.py).  Running mean / variance update + normalisation are ``imx_empirical_normalization``."""

from __future__ import annotations

import torch
import torch.nn as nn

from .. import _lib
from .._lib import check, lib


class EmpiricalNormalization(nn.Module):
    def __init__(self, shape, eps: float = 1e-2, until: int | None = None):
        super().__init__()
        self.eps, self.until = eps, until
        shape = list(shape) if isinstance(shape, (list, tuple)) else [shape]
        self.register_buffer("_mean", torch.zeros(shape).unsqueeze(0))
        self.register_buffer("_var", torch.ones(shape).unsqueeze(0))
        self.register_buffer("_std", torch.ones(shape).unsqueeze(0))
        self.register_buffer("_count_f", torch.zeros(1))  # float on the device: no host sync in forward
        self._host_count = 0

    @property
    def mean(self):
        return self._mean.squeeze(0).clone()

    @property
    def std(self):
        return self._std.squeeze(0).clone()

    @property
    def count(self):
        return int(self._count_f.item())

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        x = x.contiguous()
        N, D = x.shape
        update = self.training and (self.until is None or self._host_count < self.until)
        if update:
            self._host_count += N
        out = torch.empty_like(x)
        check(lib().imx_empirical_normalization(N, D, x.data_ptr(), int(update), float(self.eps), self._mean.data_ptr(),
                                                self._var.data_ptr(), self._std.data_ptr(), self._count_f.data_ptr(),
                                                out.data_ptr(), _lib.current_stream(x.device)))
        return out

    @torch.jit.unused
    def inverse(self, y):
        return y * (self._std + self.eps) + self._mean

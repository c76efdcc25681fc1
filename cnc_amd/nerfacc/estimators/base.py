"""Common base of the transmittance estimators (reference: nerfacc/estimators/base.py:7-22): an nn.Module
that knows its device and declares the two hooks a sampler provides."""
from typing import Any

import torch
from torch import nn


class AbstractEstimator(nn.Module):
    def __init__(self) -> None:
        super().__init__()
        # an empty, non-persistent buffer that follows .to(): the estimator's device without a parameter
        self.register_buffer("_dummy", torch.empty(0), persistent=False)

    @property
    def device(self) -> torch.device:
        return self._dummy.device

    def sampling(self, *args, **kwargs) -> Any:
        """-> (ray_indices, t_starts, t_ends)"""
        raise NotImplementedError(f"{type(self).__name__} does not sample")

    def update_every_n_steps(self, *args, **kwargs) -> None:
        raise NotImplementedError(f"{type(self).__name__} has nothing to update")

"""Backward pass of one propagation timestep (filled in by the training milestone)."""
from __future__ import annotations

import torch


class PropagationStepFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, *args):
        raise NotImplementedError("training path not built yet")

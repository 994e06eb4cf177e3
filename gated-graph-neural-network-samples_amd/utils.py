"""Host-side mirror of the reference's utils.py pieces the hot path touches:
SMALL_NUMBER (utils.py:8), glorot_init (:11-13) and MLP (:39-70).  ThreadedIterator (:16-36) is not
mirrored: batches are packed once and kept resident in HBM instead of being prefetched by a thread.
"""
from __future__ import annotations

import numpy as np
import torch

SMALL_NUMBER = 1e-7


def glorot_init(shape):
    """utils.py:11-13 -- uniform(+-sqrt(6/(shape[-2]+shape[-1]))) from the global NumPy RNG."""
    initialization_range = np.sqrt(6.0 / (shape[-2] + shape[-1]))
    return np.random.uniform(low=-initialization_range, high=initialization_range, size=shape).astype(np.float32)


def tf_dropout(x: torch.Tensor, keep_prob: float, generator=None) -> torch.Tensor:
    """tf.nn.dropout: x / keep * floor(keep + U[0,1)); identity at keep_prob == 1."""
    if keep_prob >= 1.0:
        return x
    u = torch.rand(x.shape, device=x.device, dtype=x.dtype, generator=generator)
    return x / keep_prob * torch.floor(keep_prob + u)


class MLP(object):
    """utils.py:39-70.  With hid_sizes=[] (the only use, chem_tensorflow.py:153-157) the call returns
    the PRE-activation of the single layer: inputs @ dropout(W) + b  (utils.py:64-70)."""

    def __init__(self, in_size, out_size, hid_sizes, dropout_keep_prob, device="cpu"):
        self.in_size = in_size
        self.out_size = out_size
        self.hid_sizes = list(hid_sizes)
        self.dropout_keep_prob = dropout_keep_prob
        self.device = device
        self.params = self.make_network_params()

    def make_network_params(self):
        dims = [self.in_size] + self.hid_sizes + [self.out_size]
        weight_sizes = list(zip(dims[:-1], dims[1:]))
        weights = [torch.from_numpy(self.init_weights(s)).to(self.device) for s in weight_sizes]
        biases = [torch.zeros(s[-1], dtype=torch.float32, device=self.device) for s in weight_sizes]
        return {"weights": weights, "biases": biases}

    def init_weights(self, shape):
        # numpy 1.13 (the reference's pin) keeps float32 here (value-based scalar casting); numpy 2 would
        # promote to float64 through the np.float64 scalar, so cast explicitly
        return (np.sqrt(6.0 / (shape[-2] + shape[-1])) * (2 * np.random.rand(*shape).astype(np.float32) - 1)).astype(np.float32)

    def __call__(self, inputs):
        acts = inputs
        keep = self.dropout_keep_prob() if callable(self.dropout_keep_prob) else self.dropout_keep_prob
        hid = acts
        for W, b in zip(self.params["weights"], self.params["biases"]):
            hid = acts.matmul(tf_dropout(W, keep)) + b
            acts = torch.relu(hid)
        return hid

"""Optimiser-side arithmetic of the reference's train step (chem_tensorflow.py:172-193), in torch:
tf.train.AdamOptimizer(lr) with TF-1.3's update rule, per-VARIABLE tf.clip_by_norm (not global norm),
and the data-parallel gradient reduction between them when a DataParallelContext is present.
"""
from __future__ import annotations

import math
from typing import Dict, List

import numpy as np
import torch


class TFAdam:
    """tf.train.AdamOptimizer (TF-1.3): lr_t = lr*sqrt(1-b2^t)/(1-b1^t); m,v EMA;
    var -= lr_t * m / (sqrt(v) + eps)   -- epsilon is applied to the UN-corrected sqrt(v)."""

    def __init__(self, variables: List[torch.Tensor], lr=0.001, beta1=0.9, beta2=0.999, epsilon=1e-8):
        self.vars = variables
        self.lr, self.b1, self.b2, self.eps = lr, beta1, beta2, epsilon
        self.m = [torch.zeros_like(v) for v in variables]
        self.v = [torch.zeros_like(v) for v in variables]
        self.t = 0

    @torch.no_grad()
    def apply_gradients(self, grads: List[torch.Tensor]) -> None:
        self.t += 1
        lr_t = self.lr * math.sqrt(1.0 - self.b2 ** self.t) / (1.0 - self.b1 ** self.t)
        live = [(p, g, m, v) for p, g, m, v in zip(self.vars, grads, self.m, self.v) if g is not None]
        if not live:
            return
        ps, gs, ms, vs = map(list, zip(*live))
        torch._foreach_mul_(ms, self.b1); torch._foreach_add_(ms, gs, alpha=1.0 - self.b1)
        torch._foreach_mul_(vs, self.b2); torch._foreach_addcmul_(vs, gs, gs, value=1.0 - self.b2)
        den = torch._foreach_sqrt(vs)
        torch._foreach_add_(den, self.eps)
        torch._foreach_addcdiv_(ps, ms, den, value=-lr_t)

    # checkpoint schema: the reference pickles ALL global variables incl. Adam slots and beta powers
    # (chem_tensorflow.py:311-313); names follow TF's '<var>/Adam:0', '<var>/Adam_1:0'.
    def state_variables(self, named: Dict[str, torch.Tensor]) -> Dict[str, np.ndarray]:
        out = {"beta1_power:0": np.float32(self.b1 ** (self.t + 1)), "beta2_power:0": np.float32(self.b2 ** (self.t + 1)),
               "ggnn_amd/adam_step:0": np.int64(self.t)}
        for (name, _), m, v in zip(named.items(), self.m, self.v):
            base = name[:-2] if name.endswith(":0") else name
            out[base + "/Adam:0"] = m.detach().cpu().numpy()
            out[base + "/Adam_1:0"] = v.detach().cpu().numpy()
        return out

    def _step_from_beta_powers(self, weights: Dict[str, np.ndarray]) -> int:
        """Adam step of a checkpoint written by the reference (no explicit counter): TF keeps beta^(t+1) in float32.
        0.9^(t+1) underflows to 0 after ~980 steps (a few dozen QM9 epochs) and is denormal -- i.e. wrong -- well before,
        so the step is read from beta2_power (0.999^(t+1) stays a normal float32 up to t ~ 87,000); beta1_power is only
        used while it is comfortably normal, and a power that is no longer representable means "many steps": both bias
        corrections are 1 to float32 precision from there on."""
        def power(name):
            if name not in weights:
                return None
            return float(np.asarray(weights[name], dtype=np.float64).reshape(-1)[0])
        b2p, b1p = power("beta2_power:0"), power("beta1_power:0")
        tiny = float(np.finfo(np.float32).tiny)
        if b2p is not None and tiny * 1e3 < b2p < 1.0:
            return max(int(round(math.log(b2p) / math.log(self.b2))) - 1, 0)
        if b2p is not None and b2p >= 1.0:
            return 0
        if b1p is not None and tiny * 1e3 < b1p < 1.0:
            return max(int(round(math.log(b1p) / math.log(self.b1))) - 1, 0)
        if b1p is not None and b1p >= 1.0:
            return 0
        if b1p is None and b2p is None:
            return self.t
        return 1 << 20            # powers underflowed: the bias corrections have converged

    def load_state_variables(self, named: Dict[str, torch.Tensor], weights: Dict[str, np.ndarray]) -> set:
        used = set()
        if "ggnn_amd/adam_step:0" in weights:
            self.t = int(weights["ggnn_amd/adam_step:0"]); used.add("ggnn_amd/adam_step:0")
        else:
            self.t = self._step_from_beta_powers(weights)
        used |= {"beta1_power:0", "beta2_power:0"} & set(weights)
        for i, (name, _) in enumerate(named.items()):
            base = name[:-2] if name.endswith(":0") else name
            for slot, dst in ((base + "/Adam:0", self.m), (base + "/Adam_1:0", self.v)):
                if slot in weights:
                    dst[i].copy_(torch.from_numpy(np.asarray(weights[slot])).to(dst[i].device).reshape(dst[i].shape))
                    used.add(slot)
        return used


@torch.no_grad()
def clip_by_norm_(grads: List[torch.Tensor], clip_norm: float) -> None:
    """tf.clip_by_norm applied to each gradient separately (chem_tensorflow.py:186-190):
    g * clip / max(||g||_2, clip)."""
    live = [g for g in grads if g is not None]
    if not live:
        return
    norms = torch._foreach_norm(live)
    scales = [clip_norm / torch.clamp(n, min=clip_norm) for n in norms]
    torch._foreach_mul_(live, scales)


def train_step(model, batch_data) -> torch.Tensor:
    """loss -> gradients -> (data-parallel all-reduce) -> per-variable clip -> Adam.

    Under data parallelism each rank holds a different shard of graphs.  The reference's loss is
    sum(0.5*diff^2) / (sum(mask) + 1e-7) over the whole batch (chem_tensorflow.py:161-169), so shards
    are combined by all-reducing the loss numerator-gradients and the mask count, not by averaging
    per-rank losses (see parallel.DataParallelContext.reduce_gradients)."""
    variables = list(model.trainable_variables.values())
    for v in variables:
        v.requires_grad_(True)
        v.grad = None
    model.training = True
    try:
        loss = model.forward_batch(batch_data)
        dist = getattr(model, "dist", None)
        if dist is not None and dist.active:
            loss_for_grad = dist.global_loss(model)
        else:
            loss_for_grad = loss
        loss_for_grad.backward()
    finally:
        model.training = False
    grads = [v.grad for v in variables]
    if dist is not None and dist.active:
        dist.reduce_gradients(variables, grads)
        loss = loss_for_grad
    for v in variables:
        v.requires_grad_(False)
    clip_by_norm_(grads, model.params['clamp_gradient_norm'])
    model.optimizer.apply_gradients(grads)
    for v in variables:
        v.grad = None
    return loss.detach()

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
        self.t = 0
        self._flat = None
        if variables and all(v.is_cuda and v.dtype == torch.float32 and v.is_contiguous() and v.data_ptr() % 16 == 0 for v in variables):
            self._make_flat()
        else:                                                   # CPU tensors (host-side tests): per-variable torch slots
            self.m = [torch.zeros_like(v) for v in variables]
            self.v = [torch.zeros_like(v) for v in variables]

    # ---- fused path (GPU): gradients and Adam slots live in flat buffers, clip + Adam are two HIP launches for all variables ----
    def _make_flat(self) -> None:
        from . import _lib
        lib = _lib.load()
        B = lib.ggnn_optim_block_floats()
        dev = self.vars[0].device
        first, block_var = [0], []
        for i, v in enumerate(self.vars):
            nb = (v.numel() + B - 1) // B
            block_var += [i] * nb
            first.append(first[-1] + nb)
        nblocks = first[-1]
        i32 = lambda xs: torch.tensor(xs, dtype=torch.int32, device=dev)
        f = {"B": B, "nblocks": nblocks, "block_var": i32(block_var), "var_first": i32(first),
             "var_numel": i32([v.numel() for v in self.vars]),
             "p_ptr": torch.tensor([v.data_ptr() for v in self.vars], dtype=torch.int64, device=dev),
             "p_ptr_host": [v.data_ptr() for v in self.vars],
             "g": torch.zeros(nblocks * B, dtype=torch.float32, device=dev),
             "m": torch.zeros(nblocks * B, dtype=torch.float32, device=dev),
             "v": torch.zeros(nblocks * B, dtype=torch.float32, device=dev),
             "partial": torch.empty(nblocks, dtype=torch.float32, device=dev),
             "active_host": None, "active": torch.ones(len(self.vars), dtype=torch.int32, device=dev)}
        view = lambda buf, i, v: buf[first[i] * B:first[i] * B + v.numel()].view(v.shape)
        self.m = [view(f["m"], i, v) for i, v in enumerate(self.vars)]
        self.v = [view(f["v"], i, v) for i, v in enumerate(self.vars)]
        f["g_views"] = [view(f["g"], i, v) for i, v in enumerate(self.vars)]
        self._flat = f

    @property
    def fused(self) -> bool:
        return self._flat is not None

    @torch.no_grad()
    def sink_targets(self) -> Dict[int, torch.Tensor]:
        """{variable address: its view of the flat gradient buffer}: where backward.weight_gradient_sink accumulates."""
        return {v.data_ptr(): w for v, w in zip(self.vars, self._flat["g_views"])}

    @torch.no_grad()
    def zero_gradient_views(self, ptrs=None) -> None:
        views = [w for v, w in zip(self.vars, self._flat["g_views"]) if ptrs is None or v.data_ptr() in ptrs]
        if views:
            torch._foreach_zero_(views)

    @torch.no_grad()
    def load_gradients(self, grads: List[torch.Tensor], in_place=()) -> torch.Tensor:
        """Pack this step's gradients into the flat gradient buffer (ONE multi-tensor copy; variables without a gradient keep
        zeros and are marked inactive) and return the buffer -- the operand of the data-parallel all-reduce.
        in_place: addresses of variables whose gradient was accumulated straight into their view of the buffer (the
        weight-gradient sink of the propagation steps); an autograd gradient for such a variable is ADDED to it."""
        f = self._flat
        mask = [0 if (g is None and v.data_ptr() not in in_place) else 1 for g, v in zip(grads, self.vars)]
        if in_place:
            extra = [(w, g) for w, g, v in zip(f["g_views"], grads, self.vars) if g is not None and v.data_ptr() in in_place]
            for w, g in extra:
                w.add_(g.reshape(w.shape))
            grads = [None if v.data_ptr() in in_place else g for g, v in zip(grads, self.vars)]
            keep = [v.data_ptr() in in_place for v in self.vars]
        else:
            keep = [False] * len(self.vars)
        if mask != f["active_host"]:
            f["active"].copy_(torch.tensor(mask, dtype=torch.int32), non_blocking=False)
            f["active_host"] = mask
        if 0 in mask:
            for w, g, k in zip(f["g_views"], grads, keep):
                if g is None and not k:
                    w.zero_()
        dst = [w for w, g in zip(f["g_views"], grads) if g is not None]
        src = [g.reshape(w.shape) for w, g in zip(f["g_views"], grads) if g is not None]
        if src:
            torch._foreach_copy_(dst, src)
        return f["g"]

    def mark_all_active(self) -> None:
        """After a data-parallel all-reduce every variable has a (possibly zero) reduced gradient on every rank."""
        f = self._flat
        mask = [1] * len(self.vars)
        if mask != f["active_host"]:
            f["active"].fill_(1)
            f["active_host"] = mask

    @torch.no_grad()
    def clip_and_apply(self, clip_norm: float) -> None:
        """Per-variable clip_by_norm (chem_tensorflow.py:186-190) + the Adam update on the packed gradients: two launches."""
        from . import _lib
        f = self._flat
        if [v.data_ptr() for v in self.vars] != f["p_ptr_host"]:
            raise RuntimeError("a trainable variable was re-allocated after the optimizer was built")
        self.t += 1
        lr_t = self.lr * math.sqrt(1.0 - self.b2 ** self.t) / (1.0 - self.b1 ** self.t)
        lib = _lib.load()
        _lib.check(lib.ggnn_clip_adam_f32(f["p_ptr"].data_ptr(), f["var_numel"].data_ptr(), f["g"].data_ptr(), f["m"].data_ptr(),
                                          f["v"].data_ptr(), f["partial"].data_ptr(), f["block_var"].data_ptr(),
                                          f["var_first"].data_ptr(), f["active"].data_ptr(), f["nblocks"], float(clip_norm or 0.0),
                                          lr_t, self.b1, self.b2, self.eps, torch.cuda.current_stream().cuda_stream))
        # The kernel wrote the weights through raw pointers: bump their version counters, which the caches of packed LDS
        # weight images (ops.PackedWeights) and of transposed weights are keyed on.
        bump = getattr(torch.autograd.graph, "increment_version", None)
        if bump is not None:
            for v in self.vars:
                bump(v)
        else:
            torch._foreach_add_(self.vars, 0.0)

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
    from . import train_native
    if train_native.eligible(model, batch_data):
        # the default model: forward, backward and weight-gradient products as native launch sequences (csrc/ggnn_train.hip)
        return train_native.native_train_step(model, batch_data)
    variables = list(model.trainable_variables.values())
    for v in variables:
        v.requires_grad_(True)
        v.grad = None
    model.training = True
    opt = model.optimizer
    fused = opt.fused and len(opt.vars) == len(variables) and all(a is b for a, b in zip(opt.vars, variables))
    in_place = ()
    try:
        loss = model.forward_batch(batch_data)
        dist = getattr(model, "dist", None)
        if dist is not None and dist.active:
            loss_for_grad = dist.global_loss(model)
        else:
            loss_for_grad = loss
        if fused:
            # the propagation steps add their weight gradients straight into the flat gradient buffer, on a side stream
            # (backward.weight_gradient_sink); everything else arrives through autograd and is packed below
            from .backward import weight_gradient_sink
            opt.zero_gradient_views()
            with weight_gradient_sink(opt.sink_targets()) as sink:
                loss_for_grad.backward()
                if sink is not None:
                    torch.cuda.current_stream().wait_stream(sink.stream)
                    sink.finish()                              # pending weight-dropout masks on the accumulated gradients
                    in_place = frozenset(sink.used)
        else:
            loss_for_grad.backward()
    finally:
        model.training = False
    grads = [v.grad for v in variables]
    for v in variables:
        v.requires_grad_(False)
    if fused:
        # GPU: one multi-tensor copy packs the gradients, ONE all-reduce of the flat buffer under data parallelism, then
        # per-variable clip + Adam in two launches for all variables (ggnn_clip_adam_f32)
        flat = opt.load_gradients(grads, in_place)
        if dist is not None and dist.active:
            dist.all_reduce_sum_(flat)
            opt.mark_all_active()
            loss = loss_for_grad
        opt.clip_and_apply(model.params['clamp_gradient_norm'])
    else:
        if dist is not None and dist.active:
            dist.reduce_gradients(variables, grads)
            loss = loss_for_grad
        clip_by_norm_(grads, model.params['clamp_gradient_norm'])
        opt.apply_gradients(grads)
    for v in variables:
        v.grad = None
    return loss.detach()

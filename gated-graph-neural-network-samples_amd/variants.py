"""Training support for the NON-DEFAULT switches of chem_tensorflow_sparse.py: propagation attention (:147-149,
170-196) and the BasicRNNCell / CudnnCompatibleGRUCell cells (:105-110).

Forward of a timestep is the HIP path (ops.msg_transform -> ops.gather_segment_sum[_attn] -> ops.rnn / ops.cudnn_gru /
ops.gru), exactly what inference runs.  The default model's backward is hand-written (backward.PropagationStepFn); for
these rarely used switches the backward is DERIVED instead of written: `VariantStepFn.backward` re-evaluates the
timestep with differentiable torch ops on the same GPU tensors and lets torch autograd produce the gradients -- what
TF autodiff does for the reference (chem_tensorflow.py:184).  The re-evaluation is used for gradients only; the state
that flows forward (and the loss) comes from the HIP kernels.  Gradients are tested against the reference's own
training run (tests/golden/reference_sparse_{attention,relu_rnn,cudnn_gru}.npz).
"""
from __future__ import annotations

from typing import Optional, Sequence

import torch

from . import ops
from .utils import SMALL_NUMBER


def _activation(name: str):
    return torch.tanh if name.lower() == "tanh" else torch.relu


def _step_torch(h, index: "ops.MessageIndex", nin, edge_weights, edge_biases, attention_weights, use_avg: bool,
                residuals: Sequence[torch.Tensor], cell_type: str, cell: Sequence[torch.Tensor], activation: str):
    """One timestep of chem_tensorflow_sparse.py:153-216 in differentiable torch ops (reference op order)."""
    V, D = h.shape
    T = edge_weights.shape[0]
    src, dst = index.adj[:, 0].long(), index.adj[:, 1].long()
    off = index.type_off
    etype = torch.cat([torch.full((off[t + 1] - off[t],), t, dtype=torch.long, device=h.device) for t in range(T)]) \
        if index.num_messages else torch.zeros(0, dtype=torch.long, device=h.device)
    H = torch.einsum('vd,tde->vte', h, edge_weights)                      # :160-164 for every type at once
    messages = H[src, etype]                                              # [M, D], type-major like :168
    if attention_weights is not None:                                     # :147-149, 170-196
        scores = (h[src] * h[dst]).sum(-1) * attention_weights[etype]
        smax = torch.full((V,), torch.finfo(h.dtype).min, dtype=h.dtype, device=h.device)
        smax = smax.scatter_reduce(0, dst, scores, reduce="amax", include_self=True)
        exped = torch.exp(scores - smax[dst])
        ssum = torch.zeros(V, dtype=h.dtype, device=h.device).index_add(0, dst, exped)
        messages = messages * (exped / (ssum[dst] + SMALL_NUMBER)).unsqueeze(-1)
    incoming = torch.zeros_like(h).index_add(0, dst, messages)            # :198-200
    if edge_biases is not None:
        incoming = incoming + nin.matmul(edge_biases)                     # :202-204
    if use_avg:
        incoming = incoming / (nin.sum(dim=-1, keepdim=True) + SMALL_NUMBER)   # :206-209
    x = torch.cat(list(residuals) + [incoming], dim=-1)                   # :211-212
    if cell_type == 'rnn':                                                # BasicRNNCell: act([x,h] W + b)
        kernel, bias = cell
        return _activation(activation)(torch.cat([x, h], dim=1).matmul(kernel) + bias)
    gates = torch.sigmoid(torch.cat([x, h], dim=1).matmul(cell[0]) + cell[1])
    r, u = gates[:, :D], gates[:, D:]                                     # r first, then u
    if cell_type == 'gru':
        c = _activation(activation)(torch.cat([x, r * h], dim=1).matmul(cell[2]) + cell[3])
    else:                                                                 # CudnnCompatibleGRUCell
        c = torch.tanh(x.matmul(cell[2]) + cell[3] + r * (h.matmul(cell[4]) + cell[5]))
    return u * h + (1 - u) * c


class VariantStepFn(torch.autograd.Function):
    """apply(h, index, nin, use_avg, cell_type, activation, num_cell, num_res, edge_weights, edge_biases|None,
             attention_weights|None, *cell tensors, *residual states)"""

    @staticmethod
    def forward(ctx, h, index, nin, use_avg, cell_type, activation, num_cell, num_res, edge_weights, edge_biases,
                attention_weights, *rest):
        cell, residuals = rest[:num_cell], rest[num_cell:num_cell + num_res]
        h = h.contiguous()
        W = edge_weights.contiguous()
        H = ops.msg_transform(h, W)
        if attention_weights is not None:
            incoming = ops.gather_segment_sum_attn(H, h, index, attention_weights, nin, edge_biases, use_avg)
        else:
            incoming = ops.gather_segment_sum(H, index, nin, edge_biases, use_avg)
        xs = list(residuals) + [incoming]
        if cell_type == 'gru':
            out = ops.gru(xs, h, cell[0], cell[1], cell[2], cell[3], activation)
        elif cell_type == 'rnn':
            out = ops.rnn(xs, h, cell[0], cell[1], activation)
        else:
            out = ops.cudnn_gru(xs, h, *cell)
        ctx.index, ctx.use_avg, ctx.cell_type, ctx.activation = index, use_avg, cell_type, activation
        ctx.num_cell, ctx.num_res = num_cell, num_res
        ctx.has_bias, ctx.has_attn = edge_biases is not None, attention_weights is not None
        tensors = [h, nin, edge_weights] + ([edge_biases] if ctx.has_bias else []) + \
                  ([attention_weights] if ctx.has_attn else []) + list(cell) + list(residuals)
        ctx.save_for_backward(*tensors)
        return out

    @staticmethod
    def backward(ctx, g):
        saved = list(ctx.saved_tensors)
        h, nin, W = saved[:3]
        k = 3
        bias = saved[k] if ctx.has_bias else None
        k += int(ctx.has_bias)
        attn = saved[k] if ctx.has_attn else None
        k += int(ctx.has_attn)
        cell = saved[k:k + ctx.num_cell]
        residuals = saved[k + ctx.num_cell:k + ctx.num_cell + ctx.num_res]
        leaves = [t.detach().requires_grad_(True) for t in [h, W] + ([bias] if bias is not None else []) +
                  ([attn] if attn is not None else []) + list(cell) + list(residuals)]
        it = iter(leaves)
        h_, W_ = next(it), next(it)
        bias_ = next(it) if bias is not None else None
        attn_ = next(it) if attn is not None else None
        cell_ = [next(it) for _ in range(ctx.num_cell)]
        res_ = [next(it) for _ in range(ctx.num_res)]
        with torch.enable_grad():
            out = _step_torch(h_, ctx.index, nin, W_, bias_, attn_, ctx.use_avg, res_, ctx.cell_type, cell_, ctx.activation)
        grads = list(torch.autograd.grad(out, leaves, g.contiguous(), allow_unused=True))
        it = iter(grads)
        dh, dW = next(it), next(it)
        dbias = next(it) if bias is not None else None
        dattn = next(it) if attn is not None else None
        dcell = [next(it) for _ in range(ctx.num_cell)]
        dres = [next(it) for _ in range(ctx.num_res)]
        return (dh, None, None, None, None, None, None, None, dW, dbias, dattn, *dcell, *dres)


def variant_step(h, index, nin, edge_weights, edge_biases: Optional[torch.Tensor], attention_weights: Optional[torch.Tensor],
                 use_avg: bool, residual_states: Sequence[torch.Tensor], cell_type: str, cell: Sequence[torch.Tensor],
                 activation: str) -> torch.Tensor:
    return VariantStepFn.apply(h, index, nin, use_avg, cell_type, activation, len(cell), len(residual_states), edge_weights,
                               edge_biases, attention_weights, *cell, *residual_states)

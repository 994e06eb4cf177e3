"""Training support for the NON-DEFAULT switches of chem_tensorflow_sparse.py: propagation attention (:147-149,
170-196) and the BasicRNNCell / CudnnCompatibleGRUCell cells (:105-110).

Forward of a timestep is the HIP path (ops.msg_transform -> ops.gather_segment_sum[_attn] -> ops.rnn / ops.cudnn_gru /
ops.gru), exactly what inference runs.  The backward is hand-written on the same kernel families as the default model's
(backward.py):
  cell      GRU: ggnn_gru_bwd_fused_f32;  BasicRNNCell: ggnn_act_bwd_f32 + ggnn_bwd_dx_f32;  CudnnCompatibleGRUCell:
            ggnn_cudnn_gru_bwd_stage_f32 + three ggnn_bwd_dx_f32 products;  every weight / bias gradient: ggnn_xty_f32
  attention ggnn_attn_bwd_target_f32 (softmax backward per target node, no [M,D] tensor), ggnn_weighted_segment_sum_f32
            (transpose gathers with the per-message coefficients), ggnn_range_sum_f32 (d attention factor per type)
  transform backward.transform_backward (compact rows; the compacted transform kernel on W^T)
Hidden sizes beyond the fused kernels' (128 / 192 / 256 and what pads to them) run the same backward on the generic kernels:
the un-fused GRU backward, ggnn_gemm_tn_f32 / ggnn_colsum_f32 for the weight and bias gradients.  The timestep restated in
differentiable torch ops lives in tests/variant_oracle.py (a test oracle; BACKWARD_ORACLE below is its hook).
"""
from __future__ import annotations

from typing import Optional, Sequence

import torch

from . import ops
from .utils import SMALL_NUMBER

# Test hook: tests/variant_oracle.py installs `autograd_backward` here -- the timestep restated in differentiable torch ops and
# differentiated by torch autograd -- to hold the hand-written backward against it.  The product never sets it.
BACKWARD_ORACLE = None


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
        D = h.shape[1]       # (any kernel width: sizes without a compacted transform kernel take the per-type GEMM in transform_backward)
        ctx.hip_backward = BACKWARD_ORACLE is None
        extra = []                                              # what the hand-written backward needs beyond the inputs
        if cell_type == 'gru':
            save = {} if ctx.hip_backward else None
            out = ops.gru(xs, h, cell[0], cell[1], cell[2], cell[3], activation, save=save)
            if save is not None:
                extra = [incoming, save["r"], save["u"], save["c"]]
        elif cell_type == 'rnn':
            out = ops.rnn(xs, h, cell[0], cell[1], activation)
            if ctx.hip_backward:
                extra = [incoming, out]
        else:
            if ctx.hip_backward:
                out, r, u, c, hc = ops.cudnn_gru_train(xs, h, *cell)
                extra = [incoming, r, u, c, hc]
            else:
                out = ops.cudnn_gru(xs, h, *cell)
        ctx.index, ctx.use_avg, ctx.cell_type, ctx.activation = index, use_avg, cell_type, activation
        ctx.num_cell, ctx.num_res = num_cell, num_res
        ctx.has_bias, ctx.has_attn = edge_biases is not None, attention_weights is not None
        ctx.num_extra = len(extra)
        tensors = [h, nin, W] + ([edge_biases] if ctx.has_bias else []) + \
                  ([attention_weights] if ctx.has_attn else []) + list(cell) + list(residuals) + extra
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
        if ctx.hip_backward:
            extra = saved[k + ctx.num_cell + ctx.num_res:]
            return _hip_backward(ctx, g.contiguous(), h, nin, W, bias, attn, cell, residuals, extra)
        return BACKWARD_ORACLE(ctx, g.contiguous(), h, nin, W, bias, attn, cell, residuals)


def _hip_backward(ctx, g, h, nin, W, bias, attn, cell, residuals, extra):
    """The timestep's backward on HIP kernels only (see the module docstring)."""
    from .autograd import _PACKED
    from .backward import _TRANSPOSED, transform_backward, weight_grad
    index = ctx.index
    V, D = h.shape
    T = W.shape[0]
    nx = len(residuals) + 1
    Kx = (nx + 1) * D
    incoming = extra[0]
    xs = list(residuals) + [incoming]
    dev = h.device

    # ---- cell: d state, d inputs (the last input = aggregated messages, mean aggregation undone), weight / bias gradients ------
    if ctx.cell_type == 'gru':
        r, u, c = extra[1:4]
        if (ops.gru_bwd_is_fused(D) and nx <= ops.GRU_FUSED_MAX_INPUTS):
            dpc, dpg, rh, dh, dxs = ops.gru_bwd_fused(g, h, r, u, c, _PACKED.gru_bwd(cell[0], cell[2], nx, D), nin, ctx.use_avg, nx,
                                                      ctx.activation)
            dinc, d_res = dxs[-1], dxs[:-1]
            wc = ops.xty(xs + [rh], dpc, ones_row=True)
            wg = ops.xty(xs + [h], dpg, ones_row=True)
            dcell = [wg[:Kx], wg[Kx], wc[:Kx], wc[Kx]]
        else:
            from .backward import _gru_backward_unfused
            from . import _lib
            _, _, dh, dinc, d_res, dWc, dbc, dWg, dbg = _gru_backward_unfused(
                _lib.load(), g, h, r, u, c, cell[0], cell[2], nin, xs, nx, T, ops.ACT_IDS[ctx.activation.lower()], ctx.use_avg,
                torch.cuda.current_stream().cuda_stream)
            dcell = [dWg, dbg, dWc, dbc]
    elif ctx.cell_type == 'rnn':                                       # h' = act([x|h] W + b)
        out = extra[1]
        dP = ops.act_bwd(g, out, ctx.activation)
        dcell = list(weight_grad(xs + [h], dP))
        dx = torch.empty((V, nx * D), dtype=torch.float32, device=dev)
        dinc = torch.empty_like(h); dh = torch.empty_like(h)
        ops.bwd_dx(dP, 1, _TRANSPOSED.get(cell[0]), nx * D, True, dx, dinc, nin, ctx.use_avg, dh, False, False, D)
        d_res = [dx[:, i * D:(i + 1) * D] for i in range(nx - 1)]
    else:                                                              # CudnnCompatibleGRUCell (:105-108)
        r, u, c, hc = extra[1:5]
        Wg, _, Wcx, _, Wch, _ = cell
        dpc, dpg, dh, dhc = ops.cudnn_gru_bwd_stage(g, h, r, u, c, hc)
        dcell = [*weight_grad(xs + [h], dpg), *weight_grad(xs, dpc), *weight_grad([h], dhc)]
        dx = torch.empty((V, nx * D), dtype=torch.float32, device=dev)
        dinc = torch.empty_like(h)
        ops.bwd_dx(dpc, 1, _TRANSPOSED.get(Wcx), nx * D, False, dx, None, None, False, None, False, False, D)       # dx = dpc Wcx^T
        ops.bwd_dx(dhc, 1, _TRANSPOSED.get(Wch), 0, False, None, None, None, False, dh, False, True, D)              # dh += dhc Wch^T
        ops.bwd_dx(dpg, 2, _TRANSPOSED.get(Wg), nx * D, True, dx, dinc, nin, ctx.use_avg, dh, True, True, D)         # += dpg Wg^T
        d_res = [dx[:, i * D:(i + 1) * D] for i in range(nx - 1)]

    # ---- aggregation: dinc is dL/d(sum of (attention-weighted) messages + nin @ bias) --------------------------------------------
    dbias = ops.gemm_tn(nin, dinc) if bias is not None else None                                 # :202-204  nin^T dinc  [T, D]
    comp = getattr(index, "_compact", None)
    if comp is None:
        comp = index._compact = ops.build_compact_sources(index)
    dattn = None
    weights = None
    if attn is not None and not index.num_messages:                     # no message: nothing attends, no gradient
        dattn = torch.zeros_like(attn)
    elif attn is not None:                                              # :170-196
        H = ops.msg_transform(h, W)                                     # the messages' values, recomputed (dense form)
        weights, coef_s, dfac = ops.attn_backward_target(H.view(-1, D), h, dinc, index, attn, dh)
        del H
        dattn = ops.range_sum(dfac, index.type_off)
        bwd = ops.compact_backward(index, comp)
        ops.weighted_segment_sum(h, bwd.source_node_index, bwd.source_node_index.msg, coef_s, out=dh, accumulate=True)
    dW = transform_backward(index, comp, h, W, dinc, dh, message_weights=weights)
    return (dh, None, None, None, None, None, None, None, dW, dbias, dattn, *dcell, *d_res)


def variant_step(h, index, nin, edge_weights, edge_biases: Optional[torch.Tensor], attention_weights: Optional[torch.Tensor],
                 use_avg: bool, residual_states: Sequence[torch.Tensor], cell_type: str, cell: Sequence[torch.Tensor],
                 activation: str) -> torch.Tensor:
    return VariantStepFn.apply(h, index, nin, use_avg, cell_type, activation, len(cell), len(residual_states), edge_weights,
                               edge_biases, attention_weights, *cell, *residual_states)

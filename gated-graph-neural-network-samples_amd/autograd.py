"""One propagation timestep as a unit (forward on the HIP kernels; backward added for training), plus
the per-graph segment sum of the readout.

propagation_step == chem_tensorflow_sparse.py:153-216 for one timestep:
    H        = h @ [W_0 | W_1 | .. | W_{T-1}]                    (msg_transform,      :160-164)
    incoming = segment_sum(H rows) [+ nin @ b] [/ (deg + 1e-7)]   (gather_segment_sum, :198-209)
    h'       = GRU([residuals.. | incoming], h)                   (gru,                :211-216)
"""
from __future__ import annotations

from typing import List, Optional, Sequence

import torch

from . import ops


# Inference path: source-compacted message transform (ops.msg_transform_compact) where the hidden size
# supports it; set False to force the dense [V,D]x[D,T*D] form (the form the training path uses).
USE_COMPACT_TRANSFORM = True

# LDS stage images of the weights, rebuilt only when a weight tensor's version counter changes
_PACKED = ops.PackedWeights()


def propagation_step(h: torch.Tensor, index: "ops.MessageIndex", nin: torch.Tensor, edge_weights: torch.Tensor,
                     edge_biases: Optional[torch.Tensor], use_avg: bool, residual_states: Sequence[torch.Tensor],
                     cell, activation: str, need_grad: bool = False, ew_mask=None, gru_fmt: int = ops.GRU_FMT_EXACT,
                     edge_fmt: int = ops.GRU_FMT_EXACT) -> torch.Tensor:
    """ew_mask (training only): (keep_prob, seed) of the layer's edge-weight dropout (chem_tensorflow_sparse.py:91) -- `edge_weights`
    is then the VARIABLE (viewed [T,D,D]); the step multiplies by the masked weights and routes the gradient back through the mask.
    gru_fmt / edge_fmt: operand format of the fused GRU forward / of the compacted message transform of this layer (formats.py; the
    model proves the range or passes BF16X3); the training step (need_grad) keeps its transforms in the exact format."""
    if need_grad:
        from .backward import PropagationStepFn
        return PropagationStepFn.apply(h, index, nin, edge_weights, edge_biases, use_avg, activation,
                                       cell.gates_kernel, cell.gates_bias, cell.candidate_kernel, cell.candidate_bias,
                                       ew_mask, int(gru_fmt), *residual_states)
    assert ew_mask is None
    D = h.shape[1]
    # same choice as the native driver (ggnn_propagate.hip): segment sum gathered inside the GRU kernel
    nx = len(residual_states) + 1
    gather_in_gru = nx <= min(int(ops.FUSE_GATHER), ops.GRU_FUSED_MAX_INPUTS) and ops.gru_gather_fused(D) and edge_biases is None
    if USE_COMPACT_TRANSFORM and ops.compact_supported(D):
        # transform only the (node, type) pairs that emit a message; the pair list is built once per batch
        comp = getattr(index, "_compact", None)
        if comp is None:
            comp = index._compact = ops.build_compact_sources(index)
        ew = edge_weights.contiguous()
        Hc = ops.msg_transform_compact_packed(h, _PACKED.edge(ew, edge_fmt), ew.shape[0], comp, fmt=edge_fmt)
        if gather_in_gru:
            Hrows, gather_row = Hc, comp.gather_row
        else:
            incoming = ops.gather_segment_sum_compact(Hc, index, comp, nin, edge_biases, use_avg)
    else:
        H = ops.msg_transform(h, edge_weights.contiguous())
        if gather_in_gru:
            Hrows, gather_row = H.view(-1, D), None
        else:
            incoming = ops.gather_segment_sum(H, index, nin, edge_biases, use_avg)
    if gather_in_gru:
        packed = _PACKED.gru(cell.gates_kernel, cell.candidate_kernel, len(residual_states) + 1, D, gru_fmt)
        return ops.gru_packed_gather(list(residual_states), h, packed, cell.gates_bias, cell.candidate_bias, Hrows, index,
                                     gather_row, nin if use_avg else None, activation, fmt=gru_fmt)
    xs = list(residual_states) + [incoming]
    if ops.gru_is_fused(D) and nx <= ops.GRU_FUSED_MAX_INPUTS:     # (more inputs: the generic two-launch GRU below)
        packed = _PACKED.gru(cell.gates_kernel, cell.candidate_kernel, len(xs), D, gru_fmt)
        return ops.gru_packed(xs, h, packed, cell.gates_bias, cell.candidate_bias, activation, fmt=gru_fmt)
    return ops.gru(xs, h, cell.gates_kernel, cell.gates_bias, cell.candidate_kernel, cell.candidate_bias, activation)


class _SegmentSumRows(torch.autograd.Function):
    """tf.unsorted_segment_sum for the readout (chem_tensorflow_sparse.py:226-228); backward = gather."""

    @staticmethod
    def forward(ctx, data, ids, num_segments):
        ctx.save_for_backward(ids)
        return ops.unsorted_segment_sum(data.contiguous(), ids, int(num_segments))

    @staticmethod
    def backward(ctx, grad_out):
        (ids,) = ctx.saved_tensors
        return grad_out.index_select(0, ids.long()), None, None


def segment_sum_rows(data: torch.Tensor, ids: torch.Tensor, num_segments: int) -> torch.Tensor:
    return _SegmentSumRows.apply(data, ids, num_segments)


class _ReadoutLoss(torch.autograd.Function):
    """gated_regression + the masked loss sums of one task as ONE differentiable unit on the fused HIP kernels
    (ggnn_readout_loss_{fwd,bwd}_f32; chem_tensorflow_sparse.py:220-231, chem_tensorflow.py:158-170): no [V,2D] concat,
    no per-node intermediates, deterministic segmented sums.  Returns (out [G], sum 0.5 diff^2, sum |diff|, sum mask)."""

    @staticmethod
    def forward(ctx, last_h, h0, graph_nodes_list, graph_ptr, node_mask, num_graphs, gate_W, gate_b, transform_W, transform_b,
                target, mask):
        last_h = last_h.contiguous()
        gW, tW = gate_W.reshape(-1).contiguous(), transform_W.reshape(-1).contiguous()
        out, gate, val, stats = ops.readout_loss_fwd(last_h, h0, graph_nodes_list, graph_ptr, node_mask, num_graphs, gW,
                                                     gate_b.reshape(-1), tW, transform_b.reshape(-1), target, mask)
        ctx.save_for_backward(last_h, h0, graph_nodes_list, node_mask, gW, tW, gate, val, out, target, mask)
        ctx.num_graphs = int(num_graphs)
        ctx.shapes = (gate_W.shape, gate_b.shape, transform_W.shape, transform_b.shape)
        num, ab, ms = stats[0], stats[1], stats[2]
        ctx.mark_non_differentiable(ms)
        return out, num, ab, ms

    @staticmethod
    def backward(ctx, d_out, d_num, d_abs, d_ms):
        last_h, h0, gnl, node_mask, gW, tW, gate, val, out, target, mask = ctx.saved_tensors
        zero = None
        if d_num is not None or d_abs is not None:
            zero = torch.zeros((), dtype=torch.float32, device=last_h.device)
        d_stats = None if zero is None else torch.stack([zero if d_num is None else d_num.reshape(()),
                                                         zero if d_abs is None else d_abs.reshape(())]).contiguous()
        d_h, dgW, dgb, dtW, dtb = ops.readout_loss_bwd(last_h, h0, gnl, node_mask, ctx.num_graphs, gW, tW, gate, val, out, target, mask,
                                                       None if d_out is None else d_out.contiguous(), d_stats)
        sg, sgb, st, stb = ctx.shapes
        return (d_h, None, None, None, None, None, dgW.reshape(sg), dgb.reshape(sgb), dtW.reshape(st), dtb.reshape(stb), None, None)


def readout_loss(last_h, h0, graph_nodes_list, graph_ptr, node_mask, num_graphs, gate_W, gate_b, transform_W, transform_b, target, mask):
    return _ReadoutLoss.apply(last_h, h0, graph_nodes_list, graph_ptr, node_mask, num_graphs, gate_W, gate_b, transform_W, transform_b,
                              target, mask)

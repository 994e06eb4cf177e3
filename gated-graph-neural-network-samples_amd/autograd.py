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
                     cell, activation: str, need_grad: bool = False) -> torch.Tensor:
    if need_grad:
        from .backward import PropagationStepFn
        return PropagationStepFn.apply(h, index, nin, edge_weights, edge_biases, use_avg, activation,
                                       cell.gates_kernel, cell.gates_bias, cell.candidate_kernel, cell.candidate_bias,
                                       *residual_states)
    D = h.shape[1]
    # same choice as the native driver (ggnn_propagate.hip): segment sum gathered inside the GRU kernel
    gather_in_gru = len(residual_states) + 1 <= int(ops.FUSE_GATHER) and ops.gru_gather_fused(D) and edge_biases is None
    if USE_COMPACT_TRANSFORM and ops.compact_supported(D):
        # transform only the (node, type) pairs that emit a message; the pair list is built once per batch
        comp = getattr(index, "_compact", None)
        if comp is None:
            comp = index._compact = ops.build_compact_sources(index)
        ew = edge_weights.contiguous()
        Hc = ops.msg_transform_compact_packed(h, _PACKED.edge(ew), ew.shape[0], comp)
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
        packed = _PACKED.gru(cell.gates_kernel, cell.candidate_kernel, len(residual_states) + 1, D)
        return ops.gru_packed_gather(list(residual_states), h, packed, cell.gates_bias, cell.candidate_bias, Hrows, index,
                                     gather_row, nin if use_avg else None, activation)
    xs = list(residual_states) + [incoming]
    if ops.gru_is_fused(D):
        packed = _PACKED.gru(cell.gates_kernel, cell.candidate_kernel, len(xs), D)
        return ops.gru_packed(xs, h, packed, cell.gates_bias, cell.candidate_bias, activation)
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

"""Backward pass of one propagation timestep (what TF autodiff derives from
chem_tensorflow_sparse.py:153-216 via optimizer.compute_gradients, chem_tensorflow.py:184).

Forward runs on the hand-written HIP kernels and saves r, u, c.  Backward:
  * GRU gate algebra: elementwise torch ops (bandwidth-bound passes over [V,D]);
  * d(gather/segment-sum) = the SAME HIP gather/segment-sum kernel driven by the transpose index
    (messages bucketed by (src,type), gathering d_incoming[dst]) -- atomics-free and deterministic;
  * the dense contractions (dX = dY W^T, dW = X^T dY) go to the vendor BLAS through torch.matmul: plain
    library GEMMs, two of them reductions over all V nodes.
"""
from __future__ import annotations

import torch

from . import ops
from .utils import SMALL_NUMBER


def _source_index(index: "ops.MessageIndex", num_nodes: int) -> "ops.MessageIndex":
    src_index = getattr(index, "_source_index", None)
    if src_index is None:
        lists = [index.adj[index.type_off[t]:index.type_off[t + 1]] for t in range(index.num_edge_types)]
        src_index = ops.build_source_index(lists, num_nodes)
        index._source_index = src_index
    return src_index


class PropagationStepFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, h, index, nin, edge_weights, edge_biases, use_avg, activation, Wg, bg, Wc, bc, *residuals):
        h = h.contiguous()
        edge_weights = edge_weights.contiguous()
        H = ops.msg_transform(h, edge_weights)
        incoming = ops.gather_segment_sum(H, index, nin, edge_biases, use_avg)
        del H
        save = {}
        h_new = ops.gru(list(residuals) + [incoming], h, Wg, bg, Wc, bc, activation, save=save)
        ctx.index, ctx.use_avg, ctx.activation, ctx.has_bias = index, use_avg, activation.lower(), edge_biases is not None
        ctx.save_for_backward(h, nin, edge_weights, Wg, Wc, incoming, save["r"], save["u"], save["c"], *residuals)
        return h_new

    @staticmethod
    def backward(ctx, g):
        h, nin, W, Wg, Wc, incoming, r, u, c, *residuals = ctx.saved_tensors
        V, D = h.shape
        T = W.shape[0]
        nx = len(residuals) + 1
        g = g.contiguous()
        xs = list(residuals) + [incoming]

        # ---- GRU blend and candidate:  h' = u*h + (1-u)*c,  c = act([x | r*h] Wc + bc)
        one_minus_u = 1.0 - u
        dpc = g * one_minus_u
        dpc = dpc * (1.0 - c * c) if ctx.activation == "tanh" else dpc * (c > 0).to(c.dtype)
        dpu = g * (h - c) * u * one_minus_u
        dh = g * u
        rh = r * h
        a_c = torch.cat(xs + [rh], dim=1)                      # [V, (nx+1)D]
        dWc = a_c.t().matmul(dpc)
        dbc = dpc.sum(0)
        dxrh = dpc.matmul(Wc.t())                              # [V, (nx+1)D]
        drh = dxrh[:, nx * D:]
        dh = dh + drh * r
        dpr = drh * h * r * (1.0 - r)
        # ---- gates: [r|u] = sigmoid([x | h] Wg + bg)
        dpg = torch.cat([dpr, dpu], dim=1)                     # [V, 2D]
        a_c[:, nx * D:] = h                                    # reuse the buffer as [x | h]
        dWg = a_c.t().matmul(dpg)
        dbg = dpg.sum(0)
        dxh = dpg.matmul(Wg.t())
        dh = dh + dxh[:, nx * D:]
        dx = dxrh[:, :nx * D] + dxh[:, :nx * D]
        d_res = [dx[:, i * D:(i + 1) * D] for i in range(nx - 1)]
        dinc = dx[:, (nx - 1) * D:]

        # ---- mean / bias / segment sum (chem_tensorflow_sparse.py:198-209)
        if ctx.use_avg:
            dinc = dinc / (nin.sum(dim=-1, keepdim=True) + SMALL_NUMBER)
        dinc = dinc.contiguous()
        dbias = nin.t().matmul(dinc) if ctx.has_bias else None
        dH = ops.segment_sum_rows_by_index(dinc, _source_index(ctx.index, V)).view(V, T * D)

        # ---- message transform H = h [W_0 | .. | W_{T-1}]
        dh = dh + dH.matmul(W.transpose(1, 2).reshape(T * D, D))
        dW = h.t().matmul(dH).view(D, T, D).transpose(0, 1)

        return (dh, None, None, dW, dbias, None, None, dWg, dbg, dWc, dbc, *d_res)

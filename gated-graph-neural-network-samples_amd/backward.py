"""Backward pass of one propagation timestep (what TF autodiff derives from
chem_tensorflow_sparse.py:153-216 via optimizer.compute_gradients, chem_tensorflow.py:184).

Forward runs on the hand-written HIP kernels and saves r, u, c.  Backward:
  * GRU gate algebra: two fused HIP element-wise passes (ggnn_gru_bwd_stage{1,2}_f32);
  * d(gather/segment-sum) = the SAME HIP gather/segment-sum kernel driven by the transpose index
    (messages bucketed by (src,type), gathering d_incoming[dst]) -- atomics-free and deterministic;
  * optionally (USE_COMPACT_TRANSFORM) the message transform runs in its compacted form in both directions (active
    (node,type) pairs only): the transpose gather lands on compact rows, dHc W_t^T is the forward kernel on
    transposed weights, the per-node sum over types is one more segment sum;
  * dX = dY W^T runs on the package's own FP32-MFMA GEMM (ggnn_gemm_f32 on the transposed weights; ~2.5x the vendor
    BLAS at these skinny shapes: M = 1e5, K and N = 100..400);
  * the weight gradients dW = X^T dY -- plain tall-skinny library GEMMs -- go to the vendor BLAS through torch,
    batched along the 1e5-long reduction (tn_matmul).
"""
from __future__ import annotations

import os

import torch

from . import _lib, ops
from ._lib import check
from .utils import SMALL_NUMBER, tn_matmul

# Training forward/backward on the compacted message transform (GGNN_TRAIN_COMPACT=1) or on the dense [V, T*D] form
# (default).  The compacted form does 3.3x fewer transform flops in both directions, but its backward needs two more
# segment sums, a row gather and per-type weight-gradient products; measured on MI355X at QM9 shapes the training
# step is 12.4 ms compacted vs 12.1 ms dense, so dense stays the default (inference always uses the compacted form).
USE_COMPACT_TRANSFORM = os.environ.get("GGNN_TRAIN_COMPACT", "0") != "0"


# Weight gradients X^T dY: vendor BLAS batched along the rows (utils.tn_matmul; default) or the package's row-split TN
# kernel ggnn_gemm_tn_f32 (GGNN_TN_KERNEL=1).  Measured on MI355X at V = 1e5: [V,400]^T [V,200] 195 us (82 TF) batched
# BLAS vs 297 us (54 TF) own kernel; [V,200]^T [V,100] 89 vs 138 us -- the own kernel (deterministic, tested) is not
# yet competitive, so the library product keeps this plain GEMM.
USE_TN_KERNEL = os.environ.get("GGNN_TN_KERNEL", "0") != "0"


def _tn(x: torch.Tensor, dy: torch.Tensor) -> torch.Tensor:
    if USE_TN_KERNEL and dy.shape[1] % 4 == 0 and dy.shape[1] <= 512:
        return ops.gemm_tn(x, dy)
    return tn_matmul(x, dy)


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
        ctx.comp = None
        if USE_COMPACT_TRANSFORM and ops.compact_supported(h.shape[1]):
            # transform only the (node, type) pairs that emit a message (~1.2 V rows instead of T V)
            comp = getattr(index, "_compact", None)
            if comp is None:
                comp = index._compact = ops.build_compact_sources(index)
            ctx.comp = comp
            H = ops.msg_transform_compact(h, edge_weights, comp)
            incoming = ops.gather_segment_sum_compact(H, index, comp, nin, edge_biases, use_avg)
        else:
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
        lib = _lib.load()
        h, nin, W, Wg, Wc, incoming, r, u, c, *residuals = ctx.saved_tensors
        V, D = h.shape
        T = W.shape[0]
        nx = len(residuals) + 1
        K = (nx + 1) * D
        g = g.contiguous()
        st = torch.cuda.current_stream().cuda_stream
        act = ops.ACT_IDS[ctx.activation]

        # ---- GRU blend and candidate:  h' = u*h + (1-u)*c,  c = act([x | r*h] Wc + bc)
        a_c = torch.empty((V, K), dtype=torch.float32, device=h.device)        # [x_0 | .. | incoming | r*h]
        for i, x in enumerate(list(residuals) + [incoming]):
            a_c[:, i * D:(i + 1) * D] = x
        dpc = torch.empty_like(h)
        dpg = torch.empty((V, 2 * D), dtype=torch.float32, device=h.device)    # [d pre-r | d pre-u]
        dh = torch.empty_like(h)
        check(lib.ggnn_gru_bwd_stage1_f32(g.data_ptr(), h.data_ptr(), r.data_ptr(), u.data_ptr(), c.data_ptr(), act,
                                          dpc.data_ptr(), dpg.data_ptr(), dh.data_ptr(), a_c.data_ptr(), K, nx * D, V, D, st))
        dWc = _tn(a_c, dpc)
        dbc = dpc.sum(0)
        dxrh = ops.gemm([dpc], Wc.t().contiguous())                            # [V, (nx+1)D] = dpc Wc^T
        # ---- gates: [r|u] = sigmoid([x | h] Wg + bg)
        drh = dxrh[:, nx * D:]
        check(lib.ggnn_gru_bwd_stage2_f32(drh.data_ptr(), K, h.data_ptr(), r.data_ptr(), dh.data_ptr(), dpg.data_ptr(), V, D, st))
        a_c[:, nx * D:] = h                                                    # reuse the buffer as [x | h]
        dWg = _tn(a_c, dpg)
        dbg = dpg.sum(0)
        dxh = ops.gemm([dpg[:, :D], dpg[:, D:]], Wg.t().contiguous())           # [V, (nx+1)D] = dpg Wg^T
        dh += dxh[:, nx * D:]
        dx = dxrh[:, :nx * D] + dxh[:, :nx * D]
        d_res = [dx[:, i * D:(i + 1) * D] for i in range(nx - 1)]
        dinc = dx[:, (nx - 1) * D:]

        # ---- mean / bias / segment sum (chem_tensorflow_sparse.py:198-209)
        if ctx.use_avg:
            dinc = dinc / (nin.sum(dim=-1, keepdim=True) + SMALL_NUMBER)
        dinc = dinc.contiguous()
        dbias = nin.t().matmul(dinc) if ctx.has_bias else None
        if ctx.comp is not None:
            # ---- compacted transform Hc[r] = h[node(r)] W_type(r): its backward on the same R rows
            comp = ctx.comp
            bwd = ops.compact_backward(ctx.index, comp)
            R = comp.num_rows
            dW = torch.zeros_like(W)
            if R:
                dHc = ops.segment_sum_rows_by_index(dinc, bwd.rows_index)                      # [R,D], transpose gather
                Z = ops.msg_transform_compact(dHc, W.transpose(1, 2).contiguous(), bwd.identity)   # dHc W_t^T, same kernel
                dh += ops.segment_sum_rows_by_index(Z[:R], bwd.node_index)                   # sum over a node's types
                hg = h.index_select(0, comp.pair_node[:R].long())
                for t in range(T):
                    a, b = comp.type_row_off[t], comp.type_row_off[t + 1]
                    if b > a:
                        dW[t] = tn_matmul(hg[a:b], dHc[a:b], chunk=512)    # (rare types have only a few thousand rows)
            return (dh, None, None, dW, dbias, None, None, dWg, dbg, dWc, dbc, *d_res)
        dH = ops.segment_sum_rows_by_index(dinc, _source_index(ctx.index, V)).view(V, T * D)

        # ---- message transform H = h [W_0 | .. | W_{T-1}]
        WT = W.transpose(1, 2).reshape(T * D, D).contiguous()                  # rows t*D..: W_t^T
        for t0 in range(0, T, 4):                                              # (the GEMM takes <= 4 K segments)
            t1 = min(t0 + 4, T)
            dh += ops.gemm([dH[:, t * D:(t + 1) * D] for t in range(t0, t1)], WT[t0 * D:t1 * D])
        dW = _tn(h, dH).view(D, T, D).transpose(0, 1)

        return (dh, None, None, dW, dbias, None, None, dWg, dbg, dWc, dbc, *d_res)

"""The optimisation step of the default sparse model on the native launch sequences of csrc/ggnn_train.hip
(ggnn_sparse_train_forward_f32 / ggnn_sparse_train_backward_f32): what chem_tensorflow.py:183-191 runs through sess.run --
forward (chem_tensorflow_sparse.py:117-218), gated regression + masked loss (:220-231, chem_tensorflow.py:158-170),
compute_gradients, per-variable clip_by_norm, Adam -- without torch.autograd in between.

train.train_step takes this path when the model is the default one (GRU cell, no attention, no edge bias, state keep-prob 1, a
hidden size with the gather-fused kernels, at most two residual inputs per layer, every graph variable trainable); everything
else keeps the autograd path (backward.PropagationStepFn, variants.py), which is also what this one is tested against.

Per step the host makes two C calls for the propagation, one ops call per task for the readout forward and one for its backward, a
handful of launches for the weight images and masks, and the optimizer's two launches: ~1 ms of host time instead of ~4.4 ms.
"""
from __future__ import annotations

import ctypes
from typing import Any, Dict, Optional

import torch

from . import _lib, backward, ops
from ._lib import check
from .utils import SMALL_NUMBER


import os

# all of a step's weight images in one launch (GGNN_FUSED_PREPARE=0: mask / transpose / pack layer by layer through the caches of ops.py)
USE_FUSED_PREPARE = os.environ.get("GGNN_FUSED_PREPARE", "1") != "0"


def _ptrs(tensors):
    return (ctypes.c_void_p * len(tensors))(*[None if t is None else t.data_ptr() for t in tensors])


def _i32(xs):
    return (ctypes.c_int32 * max(len(xs), 1))(*[int(x) for x in xs])


def model_eligible(model) -> bool:
    """The part of `eligible` that depends on the model only (run_epoch uses it to choose how batches are prefetched)."""
    p = getattr(model, "params", None)
    if p is None or not hasattr(model, "_edge_weight_vars") or not torch.cuda.is_available():
        return False
    if not backward.USE_NATIVE_STEP or not backward.USE_COMPACT_TRANSFORM or not backward.TRAIN_GATHER_IN_GRU:
        return False
    D = p['hidden_size']
    if getattr(model, "cell_type", None) != 'gru' or p['use_propagation_attention'] or p['use_edge_bias'] or not p['use_graph']:
        return False
    if torch.device(model.device).type != 'cuda':
        return False
    if model._kw != D or not ops.gru_gather_fused(D) or not ops.compact_supported(D) or D > 104 or not ops.gru_bwd_is_fused(D):
        return False
    if float(p.get('graph_state_dropout_keep_prob', 1.0)) < 1.0:
        return False
    L = len(p['layer_timesteps'])
    if L > 60 or any(int(s) < 1 for s in p['layer_timesteps']):
        return False
    if any(len(p['residual_connections'].get(str(l)) or []) + 1 > ops.GRU_FUSED_MAX_INPUTS for l in range(L)):
        return False
    opt = model.optimizer
    variables = list(model.trainable_variables.values())
    if not (opt.fused and len(opt.vars) == len(variables) and all(a is b for a, b in zip(opt.vars, variables))):
        return False
    have = {v.data_ptr() for v in variables}
    if any(v.data_ptr() not in have for v in model.named_variables().values()):        # (--freeze-graph-model)
        return False
    for task_id in p['task_ids']:
        if len(model.weights['regression_gate_task%i' % task_id].params["weights"]) != 1:
            return False
    return True


def eligible(model, batch_data: Dict[str, Any]) -> bool:
    """True when this step can run on the native sequences (see the module docstring)."""
    if ops._timing is not None or not model_eligible(model):
        return False
    D = model.params['hidden_size']
    if float(batch_data.get('graph_state_keep_prob', 1.0)) < 1.0:
        return False
    h0 = batch_data.get('initial_node_representation')
    index = batch_data.get('message_index')
    if h0 is None or index is None or not h0.is_cuda or h0.shape[0] == 0 or h0.shape[1] != D or index.num_messages == 0:
        return False
    if batch_data.get('graph_nodes_sorted') is not True or int(batch_data['num_graphs']) == 0:
        return False
    return True


class _Workspace:
    """One growing device buffer per model: the saved tensors and temporaries of a step (ggnn_sparse_train_workspace_bytes) and the
    layer-input gradient accumulators.  Allocated on the training stream, reused by every step."""

    def __init__(self):
        self.buf: Optional[torch.Tensor] = None
        self.dstate: Optional[torch.Tensor] = None

    def get(self, nbytes: int, L: int, V: int, D: int, device):
        if self.buf is None or self.buf.numel() < nbytes or self.buf.device != device:
            self.buf = torch.empty(int(nbytes * 1.05) + 256, dtype=torch.uint8, device=device)
        if self.dstate is None or self.dstate.numel() < L * V * D or self.dstate.device != device:
            self.dstate = torch.empty(int(L * V * D * 1.05) + 64, dtype=torch.float32, device=device)
        return self.buf, [self.dstate[l * V * D:(l + 1) * V * D].view(V, D) for l in range(L)]


def native_train_step(model, batch_data: Dict[str, Any]) -> torch.Tensor:
    from .autograd import _PACKED
    lib = _lib.load()
    p = model.params
    opt = model.optimizer
    model.feed(batch_data)
    ph = model.placeholders
    h0 = ph['initial_node_representation'].contiguous()
    V, D = h0.shape
    T = model.num_edge_types
    L = len(p['layer_timesteps'])
    steps = int(sum(p['layer_timesteps']))
    index = ph['message_index']
    comp = getattr(index, "_compact", None)
    if comp is None or getattr(comp, "_bwd", None) is None:
        ops.prepare_message_index(index, D, True, training=True)
        comp = index._compact
    bwd = ops.compact_backward(index, comp)
    R = comp.num_rows
    use_avg = bool(p['use_edge_msg_avg_aggregation'])
    nin = ph['num_incoming_edges_per_type']
    act = ops.ACT_IDS[p['graph_rnn_activation'].lower()]
    residuals = [[int(i) for i in (p['residual_connections'].get(str(l)) or [])] for l in range(L)]
    res_ptr, res_idx = [0], []
    for r in residuals:
        res_idx.extend(r)
        res_ptr.append(len(res_idx))
    st = torch.cuda.current_stream()
    if getattr(model, "_native_ws", None) is None:
        model._native_ws = _Workspace()
    side = backward.side_stream(h0.device)
    dev = h0.device

    with torch.no_grad():
        # ---- this step's weights: masked edge weights (:91, one mask per layer and step) and the kernels' stage images ----
        ew_keep = float(ph.get('edge_weight_dropout_keep_prob', 1.0))
        cells = model.gnn_weights.rnn_cells
        masks = [(ew_keep, model.dropout_seed('edge_weights', l)) for l in range(L)] if ew_keep < 1.0 else []
        nxs = [len(residuals[l]) + 1 for l in range(L)]
        # operand format of each layer's GRU forward: two-piece f16 only where this step's operands are provably in its range
        # (formats.py: max|h0|, the weights' maxima tracked across optimizer steps, tanh cell, mean aggregation), else exact bf16x3
        gru_fmts = model.gru_formats(h0, ew_keep, 1.0, training=True)
        fm = _i32(gru_fmts)
        if L <= 16 and USE_FUSED_PREPARE:
            # all ~120 images of the step in ONE launch, the weight-dropout mask applied on the fly (ggnn_sparse_train_prepare_f32)
            imgs = getattr(model, "_native_images", None)
            if imgs is None:
                f32 = lambda nbytes: torch.empty(nbytes // 4, dtype=torch.float32, device=dev)
                eb = lib.ggnn_msg_transform_compact_workspace_bytes(D, T)
                imgs = model._native_images = (
                    [f32(eb) for _ in range(L)], [f32(eb) for _ in range(L)],
                    [f32(lib.ggnn_gru_packed_bytes(D, nxs[l])) for l in range(L)],
                    [f32(lib.ggnn_gru_bwd_packed_bytes(D, nxs[l])) for l in range(L)])
            edge_packed, edge_packed_t, gru_packed, gru_bwd_packed = imgs
            seeds = (ctypes.c_uint64 * L)(*[m[1] for m in masks]) if masks else None
            check(lib.ggnn_sparse_train_prepare_f32(
                L, T, D, _i32(nxs), _ptrs(model._edge_weight_vars), ew_keep if masks else 1.0, seeds,
                _ptrs([c.gates_kernel for c in cells]), _ptrs([c.candidate_kernel for c in cells]), fm, _ptrs(edge_packed),
                _ptrs(edge_packed_t), _ptrs(gru_packed), _ptrs(gru_bwd_packed), st.cuda_stream))
        else:
            edge_packed, edge_packed_t, gru_packed, gru_bwd_packed = [], [], [], []
            for l in range(L):
                W = model._edge_weight_vars[l].view(T, D, D)
                if masks:
                    W = backward._MASKED.get(W, masks[l][0], masks[l][1])
                edge_packed.append(_PACKED.edge(W))
                edge_packed_t.append(_PACKED.edge(backward._TRANSPOSED.get(W, (1, 2))))
                gru_packed.append(_PACKED.gru(cells[l].gates_kernel, cells[l].candidate_kernel, nxs[l], D, gru_fmts[l]))
                gru_bwd_packed.append(_PACKED.gru_bwd(cells[l].gates_kernel, cells[l].candidate_kernel, nxs[l], D))

        # ---- forward ------------------------------------------------------------------------------------------------------
        ws_bytes = lib.ggnn_sparse_train_workspace_bytes(V, D, T, R, steps)
        ws, dstate = model._native_ws.get(ws_bytes, L, V, D, dev)
        tro = (ctypes.c_int64 * (T + 1))(*comp.type_row_off)
        lt, rp, ri = _i32(p['layer_timesteps']), _i32(res_ptr), _i32(res_idx)
        bg = _ptrs([c.gates_bias for c in cells]); bc = _ptrs([c.candidate_bias for c in cells])
        final_off = ctypes.c_int64(0)
        check(lib.ggnn_sparse_train_forward_f32(
            h0.data_ptr(), V, D, T, index.row_ptr.data_ptr(), comp.gather_row.data_ptr(), comp.pair_node.data_ptr(), tro,
            nin.data_ptr(), 1 if use_avg else 0, L, lt, rp, ri, _ptrs(edge_packed), bg, bc, _ptrs(gru_packed), fm, act,
            ws.data_ptr(), ws.numel(), ctypes.byref(final_off), st.cuda_stream))
        off = int(final_off.value)
        final = ws[off:off + V * D * 4].view(torch.float32).view(V, D)
        model.ops['final_node_representations'] = final

        # ---- gated regression + masked loss per task (:220-231, chem_tensorflow.py:158-170) ---------------------------------
        G = int(ph['num_graphs'])
        gnl, gptr = ph['graph_nodes_list'], ph.get('graph_ptr')
        saved, losses = [], []
        for internal_id, task_id in enumerate(p['task_ids']):
            gate, tr = model.weights['regression_gate_task%i' % task_id], model.weights['regression_transform_task%i' % task_id]
            gW, tW = gate.dropped_weight(0).reshape(-1).contiguous(), tr.dropped_weight(0).reshape(-1).contiguous()    # utils.py:68
            target = ph['target_values'][internal_id, :].contiguous()
            mask = ph['target_mask'][internal_id, :].contiguous()
            out, ngate, nval, stats = ops.readout_loss_fwd(final, h0, gnl, gptr, None, G, gW, gate.params["biases"][0].reshape(-1), tW,
                                                           tr.params["biases"][0].reshape(-1), target, mask)
            saved.append((gate, tr, gW, tW, ngate, nval, out, target, mask))
            num, ab, ms = stats[0], stats[1], stats[2]
            den = ms + SMALL_NUMBER
            ratio = 1.0 / (p['task_sample_ratios'].get(task_id) or 1.0)                 # chem_tensorflow.py:168
            model.ops['accuracy_task%i' % task_id] = ab / den
            model.ops['loss_numerator_task%i' % task_id] = num
            model.ops['abs_error_sum_task%i' % task_id] = ab
            model.ops['loss_denominator_task%i' % task_id] = ms
            losses.append(num / den * ratio)
            model.output = out
        model.ops['losses'] = losses
        loss = torch.stack(losses).sum()
        model.ops['loss'] = loss

        # data parallelism: the loss is normalised by the mask count of the WHOLE step (parallel.DataParallelContext.global_loss)
        dist = getattr(model, "dist", None)
        sharded = dist is not None and dist.active
        dens = torch.stack([model.ops['loss_denominator_task%i' % t] for t in p['task_ids']]).to(torch.float32)
        if sharded:
            dist.all_reduce_sum_(dens)
            loss = torch.stack([model.ops['loss_numerator_task%i' % t] / (dens[i] + SMALL_NUMBER) *
                                (1.0 / (p['task_sample_ratios'].get(t) or 1.0)) for i, t in enumerate(p['task_ids'])]).sum()

        # ---- backward -----------------------------------------------------------------------------------------------------
        opt._flat["g"].zero_()
        gviews = opt.sink_targets()
        d_final = None
        for internal_id, task_id in enumerate(p['task_ids']):
            gate, tr, gW, tW, ngate, nval, out, target, mask = saved[internal_id]
            ratio = 1.0 / (p['task_sample_ratios'].get(task_id) or 1.0)
            d_stats = torch.stack([ratio / (dens[internal_id] + SMALL_NUMBER), torch.zeros((), dtype=torch.float32, device=dev)]).contiguous()
            dst = [gviews[gate.params["weights"][0].data_ptr()], gviews[gate.params["biases"][0].data_ptr()],
                   gviews[tr.params["weights"][0].data_ptr()], gviews[tr.params["biases"][0].data_ptr()]]
            d_final = ops.readout_loss_bwd(final, h0, gnl, None, G, gW, tW, ngate, nval, out, target, mask, None, d_stats,
                                           d_last_h=d_final, grad_out=dst)[0]
            keep = float(ph.get('out_layer_dropout_keep_prob', 1.0))
            if keep < 1.0:                                                              # utils.py:68: the masks of the forward
                ops.dropout(dst[0], keep, gate.dropout_seed(0), out=dst[0])
                ops.dropout(dst[2], keep, tr.dropout_seed(0), out=dst[2])
        gv = lambda t: gviews[t.data_ptr()]
        check(lib.ggnn_sparse_train_backward_f32(
            h0.data_ptr(), V, D, T, comp.pair_node.data_ptr(), tro, nin.data_ptr(), 1 if use_avg else 0, L, lt, rp, ri,
            bwd.rows_index.row_ptr.data_ptr(), bwd.rows_index.gather_row.data_ptr(),
            ops._ptr(ops.slot_heads(bwd.rows_index, bwd.rows_index.row_ptr, bwd.rows_index.gather_row, R)),
            bwd.node_index.row_ptr.data_ptr(), bwd.node_index.gather_row.data_ptr(),
            ops._ptr(ops.slot_heads(bwd.node_index, bwd.node_index.row_ptr, bwd.node_index.gather_row, V)),
            bwd.identity.pair_node.data_ptr(), _ptrs(edge_packed_t), _ptrs(gru_bwd_packed), act,
            _ptrs([gv(model._edge_weight_vars[l]) for l in range(L)]), _ptrs([gv(c.gates_kernel) for c in cells]),
            _ptrs([gv(c.gates_bias) for c in cells]), _ptrs([gv(c.candidate_kernel) for c in cells]),
            _ptrs([gv(c.candidate_bias) for c in cells]), d_final.data_ptr(), _ptrs(dstate), ws.data_ptr(), ws.numel(),
            st.cuda_stream, side.cuda_stream))
        for l, (keep, seed) in enumerate(masks):             # :91 d variable = mask / keep * d masked weights, once per layer
            g = gv(model._edge_weight_vars[l])
            ops.dropout(g, keep, seed, out=g)

        # ---- (all-reduce) -> per-variable clip -> Adam ---------------------------------------------------------------------
        if sharded:
            dist.all_reduce_sum_(opt._flat["g"])
        opt.mark_all_active()
        opt.clip_and_apply(p['clamp_gradient_norm'])
    return loss.detach()

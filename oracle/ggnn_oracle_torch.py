"""CPU ORACLE #2 (test infrastructure, NOT product code): torch-CPU restatement of the sparse and
dense GGNN propagation in the reference's op order.  Parity status: held to the reference-run vectors
(tests/golden/reference_*.npz); TensorFlow's kernels themselves unpinned -- see the ggnn_oracle.py header.

Independent of ggnn_oracle.py on purpose (torch ops, index_add_, F.linear-free explicit matmuls) so
the two restatements can be checked against each other; differentiable, so torch autograd of this
file is the gradient oracle for the hand-written backward kernels; and it is the `cpu_baseline`
("kind": "port") that bench.py times next to the GPU path: TF-1.x cannot run here, and this is
the closest stand-in (multi-threaded BLAS matmuls + serial index-ordered segment sum, like
TF-1.3's CPU kernels).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
from __future__ import annotations

import torch

SMALL_NUMBER = 1e-7  # utils.py:8


def _act(name):
    name = name.lower()
    if name == "tanh":
        return torch.tanh
    if name == "relu":
        return torch.relu
    raise Exception("Unknown activation function type '%s'." % name)


def gru(x, h, Wg, bg, Wc, bc, act=torch.tanh):
    """TF-1.3 GRUCell.call as used at chem_tensorflow_sparse.py:104,215-216 (r first, then u)."""
    D = h.shape[1]
    gates = torch.sigmoid(torch.cat([x, h], dim=1).matmul(Wg) + bg)
    r, u = gates[:, :D], gates[:, D:]
    c = act(torch.cat([x, r * h], dim=1).matmul(Wc) + bc)
    return u * h + (1 - u) * c


def rnn_cell(x, h, cell, kind, act):
    """The three cell types of chem_tensorflow_sparse.py:102-112 (TF-1.3 semantics)."""
    if kind == "gru":
        return gru(x, h, cell["Wg"], cell["bg"], cell["Wc"], cell["bc"], act)
    if kind == "rnn":                       # BasicRNNCell: act([x,h] W + b)
        return act(torch.cat([x, h], dim=1).matmul(cell["W"]) + cell["b"])
    if kind == "cudnncompatiblegrucell":    # c = tanh(x Wcx + bcx + r * (h Wch + bch))
        D = h.shape[1]
        g = torch.sigmoid(torch.cat([x, h], dim=1).matmul(cell["Wg"]) + cell["bg"])
        r, u = g[:, :D], g[:, D:]
        c = torch.tanh(x.matmul(cell["Wcx"]) + cell["bcx"] + r * (h.matmul(cell["Wch"]) + cell["bch"]))
        return u * h + (1 - u) * c
    raise Exception("Unknown RNN cell type '%s'." % kind)


def sparse_step(h, adjacency_lists, nin, edge_weights, cell, residual_states=(), edge_biases=None,
                avg=True, act=torch.tanh, kind="gru", attention_weights=None):
    """chem_tensorflow_sparse.py:153-216, one timestep."""
    V = h.shape[0]
    msgs, tgts, srcs, types = [], [], [], []
    for t, adj in enumerate(adjacency_lists):                          # :159
        src = adj[:, 0].long()
        msgs.append(h.index_select(0, src).matmul(edge_weights[t]))    # :161-164
        tgts.append(adj[:, 1].long())
        srcs.append(src)
        types.append(torch.full((adj.shape[0],), t, dtype=torch.long, device=h.device))
    msgs = torch.cat(msgs, 0)                                          # :168
    tgts = torch.cat(tgts, 0)                                          # :128
    if attention_weights is not None:                                  # :170-196, written with per-target loops
        srcs = torch.cat(srcs, 0); types = torch.cat(types, 0)
        scores = (h[srcs] * h[tgts]).sum(-1) * attention_weights[types]
        att = torch.zeros_like(scores)
        for v in tgts.unique().tolist():
            sel = (tgts == v).nonzero().flatten()
            e = torch.exp(scores[sel] - scores[sel].max())
            att[sel] = e / (e.sum() + SMALL_NUMBER)
        msgs = msgs * att[:, None]
    incoming = torch.zeros(V, h.shape[1], dtype=h.dtype, device=h.device).index_add_(0, tgts, msgs)  # :198-200
    if edge_biases is not None:
        incoming = incoming + nin.matmul(edge_biases)                  # :202-204
    if avg:
        incoming = incoming / (nin.sum(dim=-1, keepdim=True) + SMALL_NUMBER)  # :206-209
    x = torch.cat(list(residual_states) + [incoming], dim=-1)          # :211-212
    return rnn_cell(x, h, cell, kind, act)                             # :215-216


def sparse_propagate(h0, adjacency_lists, nin, layers, params, return_all_layers=False):
    """chem_tensorflow_sparse.py:117-218 with keep-probs = 1.  All inputs torch CPU tensors;
    `layers` as in ggnn_oracle.make_sparse_layers (values already torch tensors)."""
    act = _act(params.get("graph_rnn_activation", "tanh"))
    states = [h0]
    for li, nts in enumerate(params["layer_timesteps"]):
        res_ids = params.get("residual_connections", {}).get(str(li))
        res = [] if res_ids is None else [states[i] for i in res_ids]
        L = layers[li]
        eb = L.get("edge_biases") if params.get("use_edge_bias", False) else None
        aw = L.get("edge_type_attention_weights") if params.get("use_propagation_attention", False) else None
        kind = params.get("graph_rnn_cell", "GRU").lower()
        cur = states[-1]
        for _ in range(nts):
            cur = sparse_step(cur, adjacency_lists, nin, L["edge_weights"], L, res, eb,
                              params.get("use_edge_msg_avg_aggregation", True), act, kind, aw)
        states.append(cur)
    return states if return_all_layers else states[-1]


def gated_regression(last_h, h0, graph_nodes_list, num_graphs, gate_W, gate_b, tr_W, tr_b):
    """chem_tensorflow_sparse.py:220-231 with utils.MLP(hid_sizes=[]) = x@W+b (utils.py:64-70)."""
    gate = torch.sigmoid(torch.cat([last_h, h0], dim=-1).matmul(gate_W) + gate_b)
    gated = gate * (last_h.matmul(tr_W) + tr_b)
    out = torch.zeros(num_graphs, 1, dtype=last_h.dtype, device=last_h.device).index_add_(0, graph_nodes_list.long(), gated)
    return out[:, 0]


def task_loss(pred, target_values, target_mask):
    """chem_tensorflow.py:161-169: (loss, mae) for one task."""
    num = target_mask.sum() + SMALL_NUMBER
    diff = (pred - target_values) * target_mask
    return (0.5 * diff * diff).sum() / num, diff.abs().sum() / num


def dense_propagate(h0, adjacency, edge_weights, edge_biases, cell, num_timesteps):
    """chem_tensorflow_dense.py:93-117; h0 [b,v,D], adjacency [b,e,v,v]."""
    b, v, D = h0.shape
    A = adjacency.permute(1, 0, 2, 3)
    h = h0.reshape(-1, D)
    for _ in range(num_timesteps):
        acts = None
        for e in range(edge_weights.shape[0]):
            m = h.matmul(edge_weights[e]).reshape(b, v, D)
            if edge_biases is not None:
                m = m + edge_biases[e]
            c = torch.bmm(A[e], m)
            acts = c if acts is None else acts + c
        h = gru(acts.reshape(-1, D), h, cell["Wg"], cell["bg"], cell["Wc"], cell["bc"])
    return h.reshape(b, v, D)


def to_torch(obj, dtype=torch.float32):
    """Recursively convert NumPy weights/feeds to torch CPU tensors (ints stay integer)."""
    import numpy as np
    if isinstance(obj, dict):
        return {k: to_torch(v, dtype) for k, v in obj.items()}
    if isinstance(obj, (list, tuple)):
        return [to_torch(v, dtype) for v in obj]
    if isinstance(obj, np.ndarray):
        t = torch.from_numpy(np.ascontiguousarray(obj))
        return t.to(dtype) if t.is_floating_point() else t
    return obj

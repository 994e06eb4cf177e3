"""CPU ORACLE (test infrastructure, NOT product code) for the GGNN propagation hot path.

PARITY STATUS: pinned to the reference's own PYTHON SOURCE, unpinned below it.  The reference
(microsoft/gated-graph-neural-network-samples @ /root/reference) ships no tests, golden vectors or fixtures,
and its arithmetic lives in the third-party wheel ``tensorflow==1.3.0`` (requirements.txt:2), which cannot be
installed or run here.  What can run here is the reference's Python: tests/golden/make_reference_golden.py
imports chem_tensorflow{,_sparse,_dense}.py + utils.py from /root/reference unmodified and executes them
(constructor, data pipeline, graph construction, loss, train op) over oracle/tf13_shim -- a deferred-execution
restatement of the ~40 TF-1.3 ops they call -- and commits the vectors as tests/golden/reference_*.npz
(8 configurations: default 5-layer residual model, edge bias, sum aggregation, attention, RNN/ReLU and
CudnnCompatibleGRU cells, dense tied/untied; forward + Adam steps).  tests/test_reference_golden.py holds this
file, the torch restatement and the package's host side to those vectors; tests/test_gpu_reference_golden.py
the HIP path.  So op order, wiring, shapes, packer output, initialisation and the training recipe are pinned to
reference code that actually ran; the arithmetic of TensorFlow's kernels (GRUCell gate order, _linear,
unsorted_segment_sum order, Adam's update) is restated from TF-1.3's published source in the shim and is
NOT pinned by a TensorFlow run -- for that layer parity remains unpinned.

This file is a NumPy restatement of the reference's algorithm, written from the reference call sites cited on
every function plus TF-1.3's published op semantics (GRUCell / _linear / unsorted_segment_sum /
embedding_lookup / dropout).  Besides the reference-run vectors it is held by (a) an fp64 and an fp32
instantiation that must agree, (b) an independent torch restatement (oracle/ggnn_oracle_torch.py), (c) a scalar C
restatement (oracle/ggnn_oracle.c), (d) the sparse == dense cross-formulation identity, and
(e) hand-computed known-answer cases (tests/test_oracle.py).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
The product path (the package next to this directory) never imports anything under oracle/.

All functions take plain NumPy arrays and are written in the REFERENCE'S OP ORDER
(gather -> per-type matmul -> concat -> index-ordered segment sum -> bias -> mean -> concat -> GRU)
so the accumulation order matches TF-CPU's (type ascending, then list order).
"""
from __future__ import annotations

import numpy as np

SMALL_NUMBER = 1e-7  # utils.py:8


# ----------------------------------------------------------------------------------------------
# TF-1.3 primitive semantics
# ----------------------------------------------------------------------------------------------
def sigmoid(x):
    return 1.0 / (1.0 + np.exp(-x))


def activation(name: str):
    """chem_tensorflow_sparse.py:75-81 -- 'tanh' or 'relu' (case-insensitive)."""
    name = name.lower()
    if name == "tanh":
        return np.tanh
    if name == "relu":
        return lambda x: np.maximum(x, 0)
    raise Exception("Unknown activation function type '%s'." % name)


def unsorted_segment_sum(data, segment_ids, num_segments):
    """tf.unsorted_segment_sum (chem_tensorflow_sparse.py:198-200, 226-228).

    Output rows with no contributing id are zero; contributions are accumulated in index order
    (np.add.at is an unbuffered in-order accumulation, like TF-CPU's serial functor).
    Out-of-range ids raise, like TF-CPU's InvalidArgument.
    """
    segment_ids = np.asarray(segment_ids)
    if segment_ids.size and (segment_ids.min() < 0 or segment_ids.max() >= num_segments):
        raise IndexError("segment id out of range")
    out = np.zeros((num_segments,) + data.shape[1:], dtype=data.dtype)
    np.add.at(out, segment_ids, data)
    return out


def embedding_lookup(params, ids):
    """tf.nn.embedding_lookup == row gather (chem_tensorflow_sparse.py:161-162)."""
    ids = np.asarray(ids)
    if ids.size and (ids.min() < 0 or ids.max() >= params.shape[0]):
        raise IndexError("gather index out of range")
    return params[ids]


def gru_cell(x, h, Wg, bg, Wc, bc, act=np.tanh):
    """TF-1.3 tf.nn.rnn_cell.GRUCell.call (used at chem_tensorflow_sparse.py:104, 215-216).

    value = sigmoid(concat([x, h]) @ gates/kernel + gates/bias); r, u = split(value, 2)
    c = act(concat([x, r*h]) @ candidate/kernel + candidate/bias);  h' = u*h + (1-u)*c
    gates/kernel is [(in+D), 2D] (r columns first), gates/bias initialised to 1.0,
    candidate/kernel is [(in+D), D], candidate/bias initialised to 0.
    Returns (h', r, u, c).
    """
    D = h.shape[1]
    g = sigmoid(np.concatenate([x, h], axis=1) @ Wg + bg)
    r, u = g[:, :D], g[:, D:]
    c = act(np.concatenate([x, r * h], axis=1) @ Wc + bc)
    return u * h + (1.0 - u) * c, r, u, c


def cudnn_compatible_gru_cell(x, h, Wg, bg, Wcx, bcx, Wch, bch):
    """tf.contrib.cudnn_rnn.CudnnCompatibleGRUCell (chem_tensorflow_sparse.py:105-108):
    r,u as GRUCell; c = tanh(x@Wcx + bcx + r*(h@Wch + bch)); h' = u*h + (1-u)*c."""
    D = h.shape[1]
    g = sigmoid(np.concatenate([x, h], axis=1) @ Wg + bg)
    r, u = g[:, :D], g[:, D:]
    c = np.tanh(x @ Wcx + bcx + r * (h @ Wch + bch))
    return u * h + (1.0 - u) * c


def basic_rnn_cell(x, h, W, b, act=np.tanh):
    """tf.nn.rnn_cell.BasicRNNCell (chem_tensorflow_sparse.py:109-110): h' = act([x,h]@W + b)."""
    return act(np.concatenate([x, h], axis=1) @ W + b)


# ----------------------------------------------------------------------------------------------
# One sparse propagation step and the layer/timestep driver
# ----------------------------------------------------------------------------------------------
# ---- the split matrix path of the HIP kernels (csrc/ggnn_split.hpp), restated -------------------------------------------
# Not part of the reference: the kernels evaluate the reference's f32 products  x W  as six bf16 products of operands split into
# three bf16 pieces.  This restatement pins the two claims the kernels rest on: the split is exact, and the six-product sum has
# the error of an f32 evaluation.
def bf16_split3(a):
    """a (float32) -> (hi, mid, lo), each exactly representable in bf16 (low 16 bits zero), hi + mid + lo == a exactly.
    Truncation split: every piece takes the next 8 significand bits."""
    a = np.ascontiguousarray(a, dtype=np.float32)
    trunc = lambda x: (x.view(np.uint32) & np.uint32(0xFFFF0000)).view(np.float32)
    hi = trunc(a)
    r1 = (a - hi).astype(np.float32)          # exact: a and hi share their leading bits
    mid = trunc(r1)
    lo = (r1 - mid).astype(np.float32)        # at most 8 significant bits are left: already a bf16 value
    return hi, mid, lo


def split6_matmul(A, W, chunk=32):
    """A [M,K] x W [K,N] the way stage_mma_split accumulates it: per `chunk` of k, six dot products (lo-order terms first)
    whose bf16 x bf16 products are exact, each added to an f32 accumulator (the MFMA's C operand)."""
    a, w = bf16_split3(A), bf16_split3(W)
    order = [(0, 2), (1, 1), (0, 1), (2, 0), (1, 0), (0, 0)]      # (piece of A, piece of W): w_lo a_hi | w_mid a_mid | w_mid a_hi | w_hi a_lo | ...
    acc = np.zeros((A.shape[0], W.shape[1]), np.float32)
    for c in range(0, A.shape[1], chunk):
        for i, j in order:
            d = a[i][:, c:c + chunk].astype(np.float64) @ w[j][c:c + chunk].astype(np.float64)
            acc = (acc.astype(np.float64) + d).astype(np.float32)
    return acc


def f16_split2(a, scale=256.0):
    """a (float32) -> (hi, lo) as float32 arrays holding f16 values: hi = RN_f16(a * scale), lo = RN_f16(a * scale - hi).  The
    residual subtraction is exact; hi + lo carries 22 of a * scale's 24 significand bits (csrc/ggnn_split.hpp, GGNN_SPLIT2 --
    a round-4 EXPERIMENT, not the library's default matrix path).  `scale` is a power of two: exact."""
    t = (np.ascontiguousarray(a, dtype=np.float32) * np.float32(scale)).astype(np.float32)
    hi = t.astype(np.float16).astype(np.float32)
    lo = (t - hi).astype(np.float32).astype(np.float16).astype(np.float32)
    return hi, lo


def split3_f16_matmul(A, W, chunk=32, scale=256.0):
    """A [M,K] x W [K,N] the way stage_mma_split accumulates it under GGNN_SPLIT2: per `chunk` of k, three dot products of f16
    pieces (w_lo a_hi | w_hi a_lo | w_hi a_hi; every f16 x f16 product is exact in f32), each added to an f32 accumulator that
    holds scale^2 x the sum; the consumer multiplies by scale^-2."""
    a, w = f16_split2(A, scale), f16_split2(W, scale)
    acc = np.zeros((A.shape[0], W.shape[1]), np.float32)
    for c in range(0, A.shape[1], chunk):
        for i, j in ((0, 1), (1, 0), (0, 0)):                        # (piece of A, piece of W)
            d = a[i][:, c:c + chunk].astype(np.float64) @ w[j][c:c + chunk].astype(np.float64)
            acc = (acc.astype(np.float64) + d).astype(np.float32)
    return (acc.astype(np.float64) / (float(scale) ** 2)).astype(np.float32)


def philox4x32_10(counter, key):
    """Philox4x32-10 (Salmon et al., "Parallel random numbers: as easy as 1, 2, 3", SC'11; Random123) on uint32 arrays:
    counter [..., 4], key [..., 2] -> [..., 4].  Pinned by Random123's published known-answer vectors (tests/test_oracle.py)."""
    c = [np.asarray(counter[..., i], dtype=np.uint64) for i in range(4)]
    k0 = np.asarray(key[..., 0], dtype=np.uint64)
    k1 = np.asarray(key[..., 1], dtype=np.uint64)
    mask = np.uint64(0xFFFFFFFF)
    for _ in range(10):
        p0 = np.uint64(0xD2511F53) * c[0]
        p1 = np.uint64(0xCD9E8D57) * c[2]
        c = [(p1 >> np.uint64(32)) ^ c[1] ^ k0, p1 & mask, (p0 >> np.uint64(32)) ^ c[3] ^ k1, p0 & mask]
        k0 = (k0 + np.uint64(0x9E3779B9)) & mask
        k1 = (k1 + np.uint64(0xBB67AE85)) & mask
    return np.stack(c, axis=-1).astype(np.uint32)


def counter_dropout(x, keep_prob, seed, row_key=None, row_key_base=0):
    """tf.nn.dropout's arithmetic, x / keep * floor(keep + U) (nn_ops.py @ TF r1.3; reference call sites
    chem_tensorflow_sparse.py:91,113-114, chem_tensorflow_dense.py:104, utils.py:68), with the package's counter-based
    uniform: U[r, c] = (philox(counter=(key_r lo, key_r hi, c // 4, 0), key=(seed lo, seed hi))[c % 4] >> 8) * 2^-24,
    key_r = row_key[r] or row_key_base + r.  x [rows, cols] float32 -> float32 (bit-exact twin of ggnn_dropout_f32)."""
    x = np.asarray(x, dtype=np.float32)
    rows, cols = x.shape
    quads = (cols + 3) // 4
    keys = (np.asarray(row_key, dtype=np.int64) if row_key is not None else np.int64(row_key_base) + np.arange(rows, dtype=np.int64))
    keys = keys.astype(np.uint64)
    ctr = np.zeros((rows, quads, 4), dtype=np.uint64)
    ctr[..., 0] = (keys & np.uint64(0xFFFFFFFF))[:, None]
    ctr[..., 1] = (keys >> np.uint64(32))[:, None]
    ctr[..., 2] = np.arange(quads, dtype=np.uint64)[None, :]
    seed = int(seed) & 0xFFFFFFFFFFFFFFFF
    key = np.broadcast_to(np.array([seed & 0xFFFFFFFF, seed >> 32], dtype=np.uint64), (rows, quads, 2))
    u = philox4x32_10(ctr, key).reshape(rows, quads * 4)[:, :cols]
    U = (u >> np.uint32(8)).astype(np.float32) * np.float32(1.0 / 16777216.0)
    keep = np.float32(keep_prob)
    return (x / keep * np.floor(keep + U)).astype(np.float32)


def unsorted_segment_max(data, segment_ids, num_segments):
    """tf.unsorted_segment_max (chem_tensorflow_sparse.py:180-182): segments without entries hold the lowest
    representable value of the dtype."""
    out = np.full((num_segments,) + data.shape[1:], np.finfo(data.dtype).min, dtype=data.dtype)
    np.maximum.at(out, np.asarray(segment_ids), data)
    return out


def sparse_step(h, adjacency_lists, nin, edge_weights, gru, residual_states=(), edge_biases=None,
                use_edge_msg_avg_aggregation=True, act=np.tanh, cell="gru",
                return_intermediates=False, attention_weights=None):
    """One timestep of chem_tensorflow_sparse.py:153-216 (attention branch off).

    h                [V,D]      current node states
    adjacency_lists  T x [E_t,2] int (src,dst) per edge type          (:67, :159-160)
    nin              [V,T]      incoming-edge counts per type          (:69, :310-313)
    edge_weights     [T,D,D]    this layer's per-type message weights  (:88-92)
    gru              dict Wg,bg,Wc,bc (GRU) / W,b (RNN) / cudnn-compatible set
    residual_states  tuple of [V,D] placed BEFORE the aggregated messages (:211-212)
    edge_biases      [T,D] or None                                     (:98-100, :202-204)
    """
    V = h.shape[0]
    messages, targets, source_states, edge_types = [], [], [], []
    for t, adj in enumerate(adjacency_lists):                     # :159
        adj = np.asarray(adj).reshape(-1, 2)
        edge_source_states = embedding_lookup(h, adj[:, 0])       # :161-162
        messages.append(edge_source_states @ edge_weights[t])     # :163-164
        source_states.append(edge_source_states)                  # :166
        targets.append(adj[:, 1])                                 # :125-126
        edge_types.append(np.full(len(adj), t, dtype=np.int64))   # :127
    messages = np.concatenate(messages, axis=0)                   # :168   [M,D]
    targets = np.concatenate(targets, axis=0)                     # :128   [M]
    if attention_weights is not None:                             # use_propagation_attention (:170-196)
        edge_types = np.concatenate(edge_types, axis=0)
        factors = np.asarray(attention_weights, dtype=h.dtype)[edge_types]            # :148-149
        src_states = np.concatenate(source_states, axis=0)                            # :171
        tgt_states = embedding_lookup(h, targets)                                     # :172-173
        scores = np.einsum('mi,mi->m', src_states, tgt_states) * factors              # :174-175
        smax = unsorted_segment_max(scores, targets, V)                               # :180-182
        scores = scores - smax[targets]                                               # :184-186
        exped = np.exp(scores)                                                        # :188
        ssum = unsorted_segment_sum(exped, targets, V)                                # :189-191
        attention = exped / (ssum[targets] + h.dtype.type(SMALL_NUMBER))              # :192-194
        messages = messages * attention[:, None]                                      # :196
    incoming = unsorted_segment_sum(messages, targets, V)         # :198-200
    if edge_biases is not None:
        incoming = incoming + nin.astype(h.dtype) @ edge_biases   # :202-204
    if use_edge_msg_avg_aggregation:
        num_incoming = nin.astype(h.dtype).sum(axis=-1, keepdims=True)   # :207-208
        incoming = incoming / (num_incoming + h.dtype.type(SMALL_NUMBER))  # :209
    x = np.concatenate(list(residual_states) + [incoming], axis=-1)      # :211-212
    if cell == "gru":
        h_new, r, u, c = gru_cell(x, h, gru["Wg"], gru["bg"], gru["Wc"], gru["bc"], act)
    elif cell == "rnn":
        h_new, r, u, c = basic_rnn_cell(x, h, gru["W"], gru["b"], act), None, None, None
    elif cell == "cudnncompatiblegrucell":
        h_new = cudnn_compatible_gru_cell(x, h, gru["Wg"], gru["bg"], gru["Wcx"], gru["bcx"],
                                          gru["Wch"], gru["bch"])
        r = u = c = None
    else:
        raise Exception("Unknown RNN cell type '%s'." % cell)
    if return_intermediates:
        return h_new, dict(incoming=incoming, x=x, r=r, u=u, c=c)
    return h_new


def sparse_propagate(h0, adjacency_lists, nin, layers, params, dtype=np.float64,
                     return_all_layers=False):
    """chem_tensorflow_sparse.py:117-218  compute_final_node_representations (keep-probs = 1).

    layers: list (one per entry of params['layer_timesteps']) of dicts with
        'edge_weights' [T,D,D], optional 'edge_biases' [T,D], and the cell weights
        ('Wg','bg','Wc','bc' for GRU).
    Residual index k refers to node_states_per_layer[k] (k=0 is h0)  (:140-145).
    """
    cast = lambda a: None if a is None else np.asarray(a, dtype=dtype)
    act = activation(params.get("graph_rnn_activation", "tanh"))
    cell = params.get("graph_rnn_cell", "GRU").lower()
    node_states_per_layer = [cast(h0)]                                   # :118-119
    nin = cast(nin)
    for layer_idx, num_timesteps in enumerate(params["layer_timesteps"]):  # :131
        res_ids = params.get("residual_connections", {}).get(str(layer_idx))  # :140
        residual_states = [] if res_ids is None else [node_states_per_layer[i] for i in res_ids]
        L = layers[layer_idx]
        gru = {k: cast(v) for k, v in L.items() if k not in ("edge_weights", "edge_biases", "edge_type_attention_weights")}
        ew = cast(L["edge_weights"])
        eb = cast(L.get("edge_biases")) if params.get("use_edge_bias", False) else None
        aw = cast(L.get("edge_type_attention_weights")) if params.get("use_propagation_attention", False) else None
        node_states_per_layer.append(node_states_per_layer[-1])          # :152
        for _ in range(num_timesteps):                                   # :153
            node_states_per_layer[-1] = sparse_step(
                node_states_per_layer[-1], adjacency_lists, nin, ew, gru, residual_states, eb,
                params.get("use_edge_msg_avg_aggregation", True), act, cell, attention_weights=aw)
    return node_states_per_layer if return_all_layers else node_states_per_layer[-1]  # :218


# ----------------------------------------------------------------------------------------------
# Readout + loss (boundary-adjacent; chem_tensorflow_sparse.py:220-231, chem_tensorflow.py:158-170)
# ----------------------------------------------------------------------------------------------
def mlp_linear(x, W, b):
    """utils.MLP with hid_sizes=[] returns the PRE-activation of its single layer (utils.py:64-70)."""
    return x @ W + b


def gated_regression(last_h, h0, graph_nodes_list, num_graphs, gate_W, gate_b, tr_W, tr_b):
    """chem_tensorflow_sparse.py:220-231. Returns [G] (tf.squeeze of [G,1])."""
    gate_input = np.concatenate([last_h, h0], axis=-1)                       # :222
    gated = sigmoid(mlp_linear(gate_input, gate_W, gate_b)) * mlp_linear(last_h, tr_W, tr_b)  # :223
    return unsorted_segment_sum(gated, graph_nodes_list, num_graphs)[:, 0]   # :226-229


def task_loss(pred, target_values, target_mask):
    """chem_tensorflow.py:161-169 for one task: returns (loss, mae)."""
    diff = pred - target_values                       # :161
    num = target_mask.sum() + SMALL_NUMBER            # :163
    diff = diff * target_mask                         # :164
    mae = np.abs(diff).sum() / num                    # :165
    loss = (0.5 * diff ** 2).sum() / num              # :166
    return loss, mae


# ----------------------------------------------------------------------------------------------
# Dense variant (chem_tensorflow_dense.py)
# ----------------------------------------------------------------------------------------------
def graph_to_adj_mat(graph, max_n_vertices, num_edge_types, tie_fwd_bkwd=True):
    """chem_tensorflow_dense.py:30-36: A[e, dst, src] = 1 (and the tied/offset reverse)."""
    bwd = 0 if tie_fwd_bkwd else num_edge_types // 2
    amat = np.zeros((num_edge_types, max_n_vertices, max_n_vertices))
    for src, e, dest in graph:
        amat[e - 1, dest, src] = 1
        amat[e - 1 + bwd, src, dest] = 1
    return amat


def dense_propagate(h0, adjacency, edge_weights, edge_biases, gru, num_timesteps, dtype=np.float64):
    """chem_tensorflow_dense.py:93-117 (keep-probs = 1).

    h0 [b,v,D]; adjacency [b,e,v,v] (A[b,e,dst,src]); edge_weights [e,D,D];
    edge_biases [e,1,D] or None; one shared GRU for all timesteps (:101-102).
    The bias is added to EVERY row incl. padded vertices before A_e is applied (:107-108).
    """
    h0 = np.asarray(h0, dtype)
    b, v, D = h0.shape
    A = np.asarray(adjacency, dtype).transpose(1, 0, 2, 3)            # :80  [e,b,v,v]
    W = np.asarray(edge_weights, dtype)
    g = {k: np.asarray(a, dtype) for k, a in gru.items()}
    h = h0.reshape(-1, D)                                             # :97
    for _ in range(num_timesteps):                                    # :100
        acts = None
        for e in range(W.shape[0]):                                   # :103
            m = (h @ W[e]).reshape(b, v, D)                           # :104-106
            if edge_biases is not None:
                m = m + np.asarray(edge_biases, dtype)[e]             # :107-108
            contrib = A[e] @ m                                        # :110-112
            acts = contrib if acts is None else acts + contrib
        h = gru_cell(acts.reshape(-1, D), h, g["Wg"], g["bg"], g["Wc"], g["bc"])[0]  # :115
    return h.reshape(b, v, D)                                         # :116


def dense_gated_regression(last_h, h0, node_mask, gate_W, gate_b, tr_W, tr_b):
    """chem_tensorflow_dense.py:119-129: masked per-graph sum of sigmoid(gate)*transform."""
    b, v, D = last_h.shape
    gi = np.concatenate([last_h, h0], axis=2).reshape(-1, 2 * D)
    go = sigmoid(mlp_linear(gi, gate_W, gate_b)) * mlp_linear(last_h.reshape(-1, D), tr_W, tr_b)
    return (go.reshape(b, v) * node_mask).sum(axis=1)


# ----------------------------------------------------------------------------------------------
# Host-side input construction, restated from the reference packer
# ----------------------------------------------------------------------------------------------
def graph_to_adjacency_lists(graph, num_edge_types, tie_fwd_bkwd=True):
    """chem_tensorflow_sparse.py:254-276.

    Returns ({type: int32 [E,2] sorted (src,dst)}, {type: {node: count}}).
    For tie_fwd_bkwd=False the reference is broken (bwd type = num_edge_types + e
    falls out of range and the in-degree counter increments the wrong endpoint); the INTENDED
    semantics are restated here: bwd type = num_edge_types//2 + e, the reversed edge (y,x)
    increments x.
    """
    from collections import defaultdict
    adj = defaultdict(list)
    nin = defaultdict(lambda: defaultdict(int))
    for src, e, dest in graph:
        t = e - 1                                             # :258
        adj[t].append((src, dest))                            # :259
        nin[t][dest] += 1                                     # :260
        if tie_fwd_bkwd:
            adj[t].append((dest, src))                        # :262
            nin[t][src] += 1                                  # :263
    final = {t: np.array(sorted(lm), dtype=np.int32) for t, lm in adj.items()}  # :265
    if not tie_fwd_bkwd:
        half = num_edge_types // 2
        for t, edges in adj.items():
            bt = half + t
            final[bt] = np.array(sorted((y, x) for (x, y) in edges), dtype=np.int32)  # :272
            for (x, y) in edges:
                nin[bt][x] += 1
    return final, nin


def pack_batch(graphs, num_edge_types, hidden_size, tie_fwd_bkwd=True, task_ids=(0,)):
    """chem_tensorflow_sparse.py:286-350 for ONE batch made of all `graphs` (raw JSON dicts
    {'targets','graph','node_features'}, get_data.py:82-86).  Returns the feed layout as a dict of
    NumPy arrays keyed by the reference placeholder names."""
    feats, gnl, nins, tv, tm = [], [], [], [], []
    adjs = [[] for _ in range(num_edge_types)]
    node_offset = 0
    for gi, g in enumerate(graphs):
        al, nd = graph_to_adjacency_lists(g["graph"], num_edge_types, tie_fwd_bkwd)
        n = len(g["node_features"])
        f = np.asarray(g["node_features"], dtype=np.float64)
        feats.append(np.pad(f, ((0, 0), (0, hidden_size - f.shape[1])), "constant"))  # :300-302
        gnl.append(np.full([n], gi, dtype=np.int32))                                    # :304
        for t in range(num_edge_types):
            if t in al:
                adjs[t].append(al[t] + node_offset)                                     # :307
        ni = np.zeros((n, num_edge_types))
        for t, d in nd.items():
            for node, cnt in d.items():
                ni[node, t] = cnt                                                       # :313
        nins.append(ni)
        labels = [g["targets"][task][0] for task in task_ids]                           # :241
        tv.append([0.0 if v is None else v for v in labels])
        tm.append([0.0 if v is None else 1.0 for v in labels])
        node_offset += n
    return {
        "initial_node_representation": np.concatenate(feats, axis=0),
        "num_incoming_edges_per_type": np.concatenate(nins, axis=0),
        "graph_nodes_list": np.concatenate(gnl),
        "target_values": np.transpose(np.asarray(tv, dtype=np.float64), [1, 0]),
        "target_mask": np.transpose(np.asarray(tm, dtype=np.float64), [1, 0]),
        "num_graphs": len(graphs),
        "adjacency_lists": [np.concatenate(a) if len(a) else np.zeros((0, 2), np.int32) for a in adjs],
    }


# ----------------------------------------------------------------------------------------------
# Weight construction in the reference's shapes
# ----------------------------------------------------------------------------------------------
def glorot_init(rng, shape):
    """utils.py:11-13."""
    r = np.sqrt(6.0 / (shape[-2] + shape[-1]))
    return rng.uniform(low=-r, high=r, size=shape).astype(np.float32)


def make_sparse_layers(rng, params, num_edge_types, random_bias=False):
    """Weights in the shapes of chem_tensorflow_sparse.py:83-115: per layer edge_weights
    glorot over [T*D, D] then reshaped [T,D,D] (:88-90); edge_biases [T,D] zeros (:99);
    GRU kernels glorot-like, gate bias 1.0, candidate bias 0 (TF-1.3 GRUCell).
    random_bias=True perturbs biases so parity tests exercise them."""
    D = params["hidden_size"]
    T = num_edge_types
    layers = []
    for layer_idx in range(len(params["layer_timesteps"])):
        res = params.get("residual_connections", {}).get(str(layer_idx)) or []
        in_dim = D * (len(res) + 1)
        L = {"edge_weights": glorot_init(rng, [T * D, D]).reshape(T, D, D)}
        if params.get("use_edge_bias", False):
            L["edge_biases"] = (rng.normal(0, 0.1, [T, D]) if random_bias else np.zeros([T, D])).astype(np.float32)
        if params.get("use_propagation_attention", False):   # :94-96 ones; perturbed so tests see the per-type factor
            L["edge_type_attention_weights"] = (np.ones(T) + (rng.normal(0, 0.3, T) if random_bias else 0)).astype(np.float32)
        cell = params.get("graph_rnn_cell", "GRU").lower()
        if cell == "gru":
            L["Wg"] = glorot_init(rng, [in_dim + D, 2 * D])
            L["bg"] = (np.ones(2 * D) + (rng.normal(0, 0.1, [2 * D]) if random_bias else 0)).astype(np.float32)
            L["Wc"] = glorot_init(rng, [in_dim + D, D])
            L["bc"] = (rng.normal(0, 0.1, [D]) if random_bias else np.zeros(D)).astype(np.float32)
        elif cell == "rnn":
            L["W"] = glorot_init(rng, [in_dim + D, D])
            L["b"] = (rng.normal(0, 0.1, [D]) if random_bias else np.zeros(D)).astype(np.float32)
        elif cell == "cudnncompatiblegrucell":
            L["Wg"] = glorot_init(rng, [in_dim + D, 2 * D])
            L["bg"] = np.ones(2 * D, np.float32)
            L["Wcx"] = glorot_init(rng, [in_dim, D]); L["bcx"] = np.zeros(D, np.float32)
            L["Wch"] = glorot_init(rng, [D, D]); L["bch"] = np.zeros(D, np.float32)
        else:
            raise Exception("Unknown RNN cell type '%s'." % cell)
        layers.append(L)
    return layers


def default_sparse_params():
    """chem_tensorflow.py:18-37 merged with chem_tensorflow_sparse.py:40-61."""
    return {
        "num_epochs": 3000, "patience": 25, "learning_rate": 0.001, "clamp_gradient_norm": 1.0,
        "out_layer_dropout_keep_prob": 1.0, "hidden_size": 100, "num_timesteps": 4, "use_graph": True,
        "tie_fwd_bkwd": True, "task_ids": [0], "random_seed": 0,
        "train_file": "molecules_train.json", "valid_file": "molecules_valid.json",
        "batch_size": 100000, "use_edge_bias": False, "use_propagation_attention": False,
        "use_edge_msg_avg_aggregation": True,
        "residual_connections": {"2": [0], "4": [0, 2]},
        "layer_timesteps": [2, 2, 1, 2, 1],
        "graph_rnn_cell": "GRU", "graph_rnn_activation": "tanh",
        "graph_state_dropout_keep_prob": 1.0, "task_sample_ratios": {},
        "edge_weight_dropout_keep_prob": 0.8,
    }

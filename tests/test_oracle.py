"""CPU tests that pin the oracle to itself (the reference ships no tests or golden vectors and
TensorFlow 1.3 cannot run here; the vectors of the reference's own source run over a TF-op shim are in
test_reference_golden.py).  Pins: fp64 vs fp32 NumPy, NumPy vs the
independent torch restatement, NumPy vs the scalar C restatement, the sparse == dense cross-formulation
identity, hand-computed known answers, structural properties, and the committed golden fixtures."""
import ctypes
import os

import numpy as np
import pytest
import torch

from conftest import ROOT, random_graph_batch


def small_params(oracle, **over):
    p = oracle.default_sparse_params()
    p.update(over)
    return p


def make_case(oracle, pkg, seed=0, n_graphs=40, D=100, **over):
    p = small_params(oracle, hidden_size=D, **over)
    ms = pkg.synthetic_qm9(n_graphs, mean_nodes=10, seed=seed)
    T = ms.num_fwd_edge_types * (1 if p["tie_fwd_bkwd"] else 2)
    b = pkg.data.pack_batch(ms, np.arange(n_graphs), T, D, p["tie_fwd_bkwd"])
    layers = oracle.make_sparse_layers(np.random.default_rng(seed), p, T, random_bias=True)
    rng = np.random.default_rng(seed + 1)
    h0 = rng.uniform(-1, 1, (b.num_nodes, D)).astype(np.float32)
    return p, b, layers, h0


def test_fp64_vs_fp32(oracle, pkg):
    p, b, layers, h0 = make_case(oracle, pkg)
    a = oracle.sparse_propagate(h0, b.adjacency_lists, b.num_incoming_edges_per_type, layers, p, np.float64)
    c = oracle.sparse_propagate(h0, b.adjacency_lists, b.num_incoming_edges_per_type, layers, p, np.float32)
    assert c.dtype == np.float32
    assert np.abs(a - c).max() < 5e-6


@pytest.mark.parametrize("over", [{}, {"use_edge_bias": True}, {"use_edge_msg_avg_aggregation": False},
                                  {"graph_rnn_activation": "relu"}, {"tie_fwd_bkwd": False},
                                  {"graph_rnn_cell": "RNN"}, {"graph_rnn_cell": "CudnnCompatibleGRUCell"},
                                  {"use_propagation_attention": True}, {"use_propagation_attention": True, "use_edge_bias": True}])
def test_numpy_vs_torch(oracle, oracle_torch, pkg, over):
    p, b, layers, h0 = make_case(oracle, pkg, seed=3, **over)
    a = oracle.sparse_propagate(h0, b.adjacency_lists, b.num_incoming_edges_per_type, layers, p, np.float64)
    tl = oracle_torch.to_torch(layers, torch.float64)
    t = oracle_torch.sparse_propagate(torch.from_numpy(h0).double(), [torch.from_numpy(x) for x in b.adjacency_lists],
                                      torch.from_numpy(b.num_incoming_edges_per_type).double(), tl, p)
    assert np.abs(a - t.numpy()).max() < 1e-12


def test_numpy_vs_c(oracle, pkg):
    so = os.path.join(ROOT, "oracle", "_build", "libggnn_oracle.so")
    if not os.path.exists(so):
        import subprocess
        subprocess.run(["make", "-C", os.path.join(ROOT, "oracle")], check=True)
    lib = ctypes.CDLL(so)
    p, b, layers, h0 = make_case(oracle, pkg, seed=5, n_graphs=15, use_edge_bias=True)
    L = layers[2]                                   # layer with one residual input
    V, D = h0.shape
    T = len(b.adjacency_lists)
    res = [np.ascontiguousarray(np.random.default_rng(9).uniform(-1, 1, (V, D)).astype(np.float32))]
    want = oracle.sparse_step(h0.astype(np.float64), b.adjacency_lists, b.num_incoming_edges_per_type.astype(np.float64),
                              L["edge_weights"].astype(np.float64), {k: L[k].astype(np.float64) for k in ("Wg", "bg", "Wc", "bc")},
                              [res[0].astype(np.float64)], L["edge_biases"].astype(np.float64), True, np.tanh)
    adj = np.ascontiguousarray(np.concatenate(b.adjacency_lists, 0).astype(np.int32))
    off = np.concatenate([[0], np.cumsum([len(a) for a in b.adjacency_lists])]).astype(np.int32)
    out = np.zeros((V, D), np.float32)
    fp = lambda a: np.ascontiguousarray(a, dtype=np.float32).ctypes.data_as(ctypes.c_void_p)
    resp = (ctypes.c_void_p * 1)(res[0].ctypes.data)
    rc = lib.ggnn_oracle_sparse_step_f32(fp(h0), adj.ctypes.data_as(ctypes.c_void_p), off.ctypes.data_as(ctypes.c_void_p),
                                         fp(b.num_incoming_edges_per_type), fp(L["edge_weights"]), fp(L["edge_biases"]), resp, 1,
                                         fp(L["Wg"]), fp(L["bg"]), fp(L["Wc"]), fp(L["bc"]), 1, 0, V, D, T,
                                         out.ctypes.data_as(ctypes.c_void_p))
    assert rc == 0
    assert np.abs(out - want).max() < 5e-6


def test_sparse_equals_dense(oracle, pkg):
    """with use_edge_bias, no mean aggregation, one layer of 4 steps, no residuals the sparse
    model equals the dense model (chem_tensorflow_dense.py:93-117) on the real (unpadded) nodes."""
    D, T = 20, 4
    ms = pkg.synthetic_qm9(8, mean_nodes=7, seed=11)
    p = small_params(oracle, hidden_size=D, use_edge_bias=True, use_edge_msg_avg_aggregation=False,
                     layer_timesteps=[4], residual_connections={})
    layers = oracle.make_sparse_layers(np.random.default_rng(1), p, T, random_bias=True)
    L = layers[0]
    v = int(ms.nodes_per_graph().max()) + 2
    db = pkg.data.pack_dense_batch(ms, np.arange(ms.num_graphs), v, T, D)
    sb = pkg.data.pack_batch(ms, np.arange(ms.num_graphs), T, D)
    sparse = oracle.sparse_propagate(sb.initial_node_representation, sb.adjacency_lists, sb.num_incoming_edges_per_type,
                                     layers, p)
    dense = oracle.dense_propagate(db.initial_node_representation, db.adjacency_matrix, L["edge_weights"],
                                   L["edge_biases"].reshape(T, 1, D), {k: L[k] for k in ("Wg", "bg", "Wc", "bc")}, 4)
    real = dense[db.node_mask.astype(bool)]
    assert real.shape == sparse.shape
    assert np.abs(real - sparse).max() < 1e-12
    # and the dense adjacency equals the reference's graph_to_adj_mat
    raw = ms.to_json()
    for g in range(ms.num_graphs):
        assert np.array_equal(db.adjacency_matrix[g], oracle.graph_to_adj_mat(raw[g]["graph"], v, T))


def test_known_answer_two_nodes(oracle):
    """Hand-written scalar arithmetic for D=2, one tied bond 0-1 of type 0, one isolated node 2."""
    D = 2
    h = np.array([[0.5, -0.25], [0.1, 0.3], [0.7, 0.2]])
    adj = [np.array([[0, 1], [1, 0]], np.int32)]
    nin = np.array([[1.0], [1.0], [0.0]])
    W = np.array([[[1.0, 2.0], [3.0, 4.0]]])
    Wg = np.arange(16, dtype=np.float64).reshape(4, 4) / 10.0
    bg = np.array([1.0, 1.0, 1.0, 1.0])
    Wc = np.arange(8, dtype=np.float64).reshape(4, 2) / 5.0 - 0.5
    bc = np.array([0.1, -0.2])
    got = oracle.sparse_step(h, adj, nin, W, dict(Wg=Wg, bg=bg, Wc=Wc, bc=bc), use_edge_msg_avg_aggregation=True)
    sig = lambda z: 1 / (1 + np.exp(-z))
    want = np.zeros_like(h)
    msgs = {1: h[0] @ W[0], 0: h[1] @ W[0]}
    for v in range(3):
        inc = np.zeros(2)
        if v in msgs:
            inc = msgs[v] / (1.0 + 1e-7)
        xh = np.concatenate([inc, h[v]])
        g = [sig(sum(xh[k] * Wg[k, n] for k in range(4)) + bg[n]) for n in range(4)]
        r, u = np.array(g[:2]), np.array(g[2:])
        xrh = np.concatenate([inc, r * h[v]])
        c = np.tanh(np.array([sum(xrh[k] * Wc[k, n] for k in range(4)) + bc[n] for n in range(2)]))
        want[v] = u * h[v] + (1 - u) * c
    assert np.abs(got - want).max() < 1e-14
    # isolated node: incoming is exactly 0 (0 / 1e-7), so it only sees the GRU of (0, h)
    iso, _, _, _ = oracle.gru_cell(np.zeros((1, 2)), h[2:3], Wg, bg, Wc, bc)
    assert np.abs(got[2] - iso[0]).max() < 1e-15


def test_segment_sum_semantics(oracle):
    data = np.array([[1.0], [2.0], [4.0], [8.0]])
    out = oracle.unsorted_segment_sum(data, np.array([2, 0, 2, 2]), 4)
    assert np.array_equal(out[:, 0], [2.0, 0.0, 13.0, 0.0])
    with pytest.raises(IndexError):
        oracle.unsorted_segment_sum(data, np.array([0, 1, 2, 4]), 4)
    with pytest.raises(IndexError):
        oracle.embedding_lookup(np.zeros((3, 2)), np.array([3]))


def test_properties_edge_order_empty_type_linearity(oracle):
    rng = np.random.default_rng(0)
    V, M, T, D = 30, 120, 4, 8
    h, adj, nin = random_graph_batch(rng, V, M, T, D)
    adj[2] = np.zeros((0, 2), np.int32)                       # an empty edge type (sparse.py:346-347)
    nin[:, 2] = 0
    p = small_params(oracle, hidden_size=D, layer_timesteps=[2], residual_connections={})
    layers = oracle.make_sparse_layers(rng, p, T, random_bias=True)
    base = oracle.sparse_propagate(h, adj, nin, layers, p)
    shuffled = [a[rng.permutation(len(a))] for a in adj]
    perm = oracle.sparse_propagate(h, shuffled, nin, layers, p)
    assert np.abs(base - perm).max() < 1e-13                   # permutation invariance (fp64 reassociation only)
    a = rng.normal(size=(M, D)); b = rng.normal(size=(M, D)); ids = rng.integers(0, V, M)
    lhs = oracle.unsorted_segment_sum(2 * a + 3 * b, ids, V)
    rhs = 2 * oracle.unsorted_segment_sum(a, ids, V) + 3 * oracle.unsorted_segment_sum(b, ids, V)
    assert np.abs(lhs - rhs).max() < 1e-12


def test_readout_and_loss(oracle, oracle_torch):
    rng = np.random.default_rng(2)
    V, D, G = 50, 6, 7
    last, h0 = rng.normal(size=(V, D)), rng.normal(size=(V, D))
    gnl = np.sort(rng.integers(0, G, V)).astype(np.int32)
    gW, gb, tW, tb = rng.normal(size=(2 * D, 1)), rng.normal(size=1), rng.normal(size=(D, 1)), rng.normal(size=1)
    pred = oracle.gated_regression(last, h0, gnl, G, gW, gb, tW, tb)
    tt = lambda a: torch.from_numpy(np.asarray(a))
    pred_t = oracle_torch.gated_regression(tt(last), tt(h0), tt(gnl), G, tt(gW), tt(gb), tt(tW), tt(tb))
    assert np.abs(pred - pred_t.numpy()).max() < 1e-12
    tv, tm = rng.normal(size=G), (rng.random(G) > 0.3).astype(np.float64)
    loss, mae = oracle.task_loss(pred, tv, tm)
    d = (pred - tv) * tm
    assert abs(loss - (0.5 * d ** 2).sum() / (tm.sum() + 1e-7)) < 1e-15
    lt, mt = oracle_torch.task_loss(pred_t, tt(tv), tt(tm))
    assert abs(float(lt) - loss) < 1e-12 and abs(float(mt) - mae) < 1e-12


def test_default_params_match_reference_table(oracle, pkg):
    """chem_tensorflow.py:18-37 + chem_tensorflow_sparse.py:40-61."""
    ours = pkg.SparseGGNNChemModel.default_params()
    assert ours == oracle.default_sparse_params()
    assert sum(ours["layer_timesteps"]) == 8 and ours["hidden_size"] == 100 and ours["batch_size"] == 100000


def test_golden_fixture(oracle):
    g = np.load(os.path.join(ROOT, "tests", "golden", "sparse_small.npz"), allow_pickle=True)
    p = oracle.default_sparse_params()
    p.update(g["params"].item())
    layers = g["layers"].tolist()
    adj = [g["adj_%d" % t] for t in range(int(g["T"]))]
    states = oracle.sparse_propagate(g["h0"], adj, g["nin"], layers, p, return_all_layers=True)
    for i, s in enumerate(states):
        assert np.abs(s - g["state_%d" % i]).max() < 1e-12


def test_split_matrix_path_arithmetic(oracle):
    """csrc/ggnn_split.hpp in numpy: the 3-way bf16 split of an f32 value is exact, and six of the nine partial products reproduce
    the f32 product to f32 accuracy (the three dropped ones are below 2^-23 of it)."""
    rng = np.random.default_rng(5)
    x = np.concatenate([rng.standard_normal(4000).astype(np.float32) * s for s in (1e-20, 1e-3, 1.0, 1e4, 1e30)] +
                       [np.array([0.0, -0.0, 1.0, -1.0, np.float32(2 ** -126), np.float32(3.4e38), 1 + 2 ** -23], np.float32)])
    hi, mid, lo = oracle.bf16_split3(x)
    for p in (hi, mid, lo):
        assert not np.any(p.view(np.uint32) & np.uint32(0xFFFF))           # each piece is a bf16 value
    assert np.array_equal(hi.astype(np.float64) + mid.astype(np.float64) + lo.astype(np.float64), x.astype(np.float64))
    # a GRU-sized product: K = 300, operands of the model's size
    A = rng.uniform(-1, 1, (512, 300)).astype(np.float32)
    W = rng.uniform(-0.2, 0.2, (300, 100)).astype(np.float32)
    ref = A.astype(np.float64) @ W.astype(np.float64)
    scale = np.abs(A.astype(np.float64)) @ np.abs(W.astype(np.float64))
    chain = np.zeros((512, 100), np.float32)                                # the f32 MFMA's arithmetic: one rounding per k
    for k in range(300):
        chain = (chain.astype(np.float64) + A[:, k:k + 1].astype(np.float64) * W[k:k + 1].astype(np.float64)).astype(np.float32)
    e_split = np.abs(oracle.split6_matmul(A, W) - ref)
    e_chain = np.abs(chain - ref)
    assert (e_split / scale).max() < 2e-7 and (e_chain / scale).max() < 4e-7
    assert np.sqrt((e_split ** 2).mean()) <= np.sqrt((e_chain ** 2).mean())      # fewer roundings: at least as accurate
    # dropping the second-order terms as well (three products) would NOT be f32-accurate -- the reason there are six
    a, w = oracle.bf16_split3(A), oracle.bf16_split3(W)
    three = sum(a[i].astype(np.float64) @ w[j].astype(np.float64) for i, j in ((0, 0), (0, 1), (1, 0)))
    assert (np.abs(three - ref) / scale).max() > 2e-6


def test_two_piece_f16_split_arithmetic(oracle):
    """csrc/ggnn_split.hpp under GGNN_SPLIT2 (round-4 experiment) in numpy: an f32 operand as TWO f16 pieces of a * 2^8 and three
    of the four partial products.  Not exact (22 of 24 significand bits per operand) -- but on GRU-shaped products its error against
    f64 is of the six-product bf16 form's size and below the f32 FMA chain's, which rounds its running sum K times; without the
    power-of-two scaling the lo pieces of operands below 0.125 fall into f16's subnormal range and it is no better than the chain."""
    rng = np.random.default_rng(11)
    x = np.concatenate([rng.standard_normal(4000).astype(np.float32) * s for s in (1e-3, 0.1, 1.0, 30.0)])
    hi, lo = oracle.f16_split2(x)
    for p in (hi, lo):
        assert np.array_equal(p.astype(np.float16).astype(np.float32), p)                      # each piece is an f16 value
    t = x.astype(np.float64) * 256.0
    big = np.abs(t) >= 2.0 ** -3                                                               # lo piece normal: 22 bits kept
    assert (np.abs(hi.astype(np.float64) + lo.astype(np.float64) - t)[big] <= 2.0 ** -22 * np.abs(t)[big]).all()
    err = np.abs(hi.astype(np.float64) + lo.astype(np.float64) - t)
    assert (err <= np.maximum(2.0 ** -22 * np.abs(t), 2.0 ** -25)).all()                       # (subnormal lo: half its spacing)
    for K in (100, 300):
        lim = np.sqrt(6.0 / (K + 100))
        A = np.tanh(rng.standard_normal((512, K))).astype(np.float32)
        W = rng.uniform(-lim, lim, (K, 100)).astype(np.float32)
        ref = A.astype(np.float64) @ W.astype(np.float64)
        scale = np.abs(A.astype(np.float64)) @ np.abs(W.astype(np.float64))
        chain = np.zeros((512, 100), np.float32)
        for k in range(K):
            chain = (chain.astype(np.float64) + A[:, k:k + 1].astype(np.float64) * W[k:k + 1].astype(np.float64)).astype(np.float32)
        rms = lambda y: float(np.sqrt(((y.astype(np.float64) - ref) ** 2).mean()))
        e2, e6, ec, e2u = (rms(oracle.split3_f16_matmul(A, W)), rms(oracle.split6_matmul(A, W)), rms(chain),
                           rms(oracle.split3_f16_matmul(A, W, scale=1.0)))
        assert e2 <= 1.1 * e6 and e2 <= 0.7 * ec, (K, e2, e6, ec)
        assert e2u > 1.5 * e2, (K, e2u, e2)                                                     # the scaling is what buys it
        assert (np.abs(oracle.split3_f16_matmul(A, W) - ref) / scale).max() < 2e-7


def test_philox_known_answers_and_counter_dropout(oracle):
    """The counter-based dropout mask (ggnn_dropout_f32) is Philox4x32-10: Random123's published known-answer vectors
    (kat_vectors: zero, all-ones, pi digits) pin the restatement; the mask keeps a fraction keep_prob, scales by 1/keep_prob, and
    depends on (seed, row key, column) only."""
    kat = lambda c, k: [int(v) for v in oracle.philox4x32_10(np.array(c, dtype=np.uint32), np.array(k, dtype=np.uint32))]
    assert kat([0, 0, 0, 0], [0, 0]) == [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]
    assert kat([0xffffffff] * 4, [0xffffffff] * 2) == [0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd]
    assert kat([0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344], [0xa4093822, 0x299f31d0]) == [0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1]
    x = np.random.default_rng(0).uniform(-1, 1, (3000, 100)).astype(np.float32)
    y = oracle.counter_dropout(x, 0.8, 99)
    kept = y != 0
    assert abs(kept.mean() - 0.8) < 0.01
    assert np.array_equal(y[kept], (x / np.float32(0.8))[kept])
    # row keys, not row positions, select the mask: a permuted batch carries its masks along
    keys = np.arange(3000, dtype=np.int64) * 7 + 5
    perm = np.random.default_rng(1).permutation(3000)
    a = oracle.counter_dropout(x, 0.8, 99, row_key=keys)
    b = oracle.counter_dropout(x[perm], 0.8, 99, row_key=keys[perm])
    assert np.array_equal(a[perm], b)
    assert not np.array_equal(oracle.counter_dropout(x, 0.8, 100), y)
    assert np.array_equal(oracle.counter_dropout(x, 0.8, 99, row_key_base=0), y)

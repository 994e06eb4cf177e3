"""GPU parity of the dense-adjacency path (chem_tensorflow_dense.py:93-129) against the oracle, and the
sparse == dense cross-formulation identity evaluated on the HIP kernels."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def dev(a, cuda):
    return torch.from_numpy(np.ascontiguousarray(a)).to(cuda)


@pytest.mark.parametrize("b,v,E,D,bias", [(3, 4, 4, 100, True), (16, 29, 4, 100, True), (5, 10, 2, 64, False),
                                          (2, 29, 8, 32, True)])
def test_dense_aggregate(pkg, cuda, b, v, E, D, bias):
    rng = np.random.default_rng(b * v)
    A = (rng.random((b, E, v, v)) < 0.15).astype(np.float32)
    Hm = rng.uniform(-1, 1, (b * v, E * D)).astype(np.float32)
    bb = rng.uniform(-1, 1, (E, D)).astype(np.float32) if bias else None
    got = pkg.ops.dense_aggregate(dev(A, cuda), dev(Hm, cuda), None if bb is None else dev(bb, cuda)).cpu().numpy()
    m = Hm.astype(np.float64).reshape(b, v, E, D)
    if bias:
        m = m + bb.astype(np.float64)[None, None]
    want = np.einsum("geij,gjed->gid", A.astype(np.float64), m).reshape(b * v, D)
    np.testing.assert_allclose(got, want, atol=2e-5, rtol=1e-5)


@pytest.mark.parametrize("b,v,E,D,bias,steps", [(7, 29, 4, 100, True, 4), (3, 32, 4, 100, False, 2), (5, 17, 8, 64, True, 3),
                                                 (4, 5, 2, 32, True, 4), (2, 16, 6, 100, True, 1), (256, 29, 4, 100, True, 4)])
@pytest.mark.parametrize("fmt", [3, 2])
def test_graph_resident_dense_forward(pkg, oracle, cuda, b, v, E, D, bias, steps, fmt):
    """ggnn_dense_propagate_f32 -- all timesteps of a graph in one workgroup -- against the fp64 oracle of
    chem_tensorflow_dense.py:93-117 and against the three-launches-per-timestep path on the same inputs (both run fp32 MFMA
    products in different summation orders: tolerance, not bit equality); repeatable bit for bit.  fmt: the operand format of the
    kernel's products, a per-launch argument (3 exact bf16x3; 2 two f16 pieces -- these operands are inside its range)."""
    if fmt == 2 and not pkg._lib.load().ggnn_dense_propagate_is_split(v, E, D):
        pytest.skip("the launch does not run the split-form kernel")
    rng = np.random.default_rng(b * v + E)
    A = (rng.random((b, E, v, v)) < 2.0 / v).astype(np.float32)
    h0 = rng.uniform(-1, 1, (b, v, D)).astype(np.float32)
    W = oracle.glorot_init(rng, [E, D, D])
    eb = rng.normal(0, 0.1, [E, 1, D]).astype(np.float32) if bias else None
    gru = {"Wg": oracle.glorot_init(rng, [2 * D, 2 * D]), "bg": (1 + rng.normal(0, 0.1, 2 * D)).astype(np.float32),
           "Wc": oracle.glorot_init(rng, [2 * D, D]), "bc": rng.normal(0, 0.1, D).astype(np.float32)}
    assert pkg.ops.dense_propagate_supported(v, E, D)
    P = pkg.ops.PackedWeights()
    dW, dWg, dWc = dev(W, cuda), dev(gru["Wg"], cuda), dev(gru["Wc"], cuda)
    dbias = None if eb is None else dev(eb.reshape(E, D), cuda)
    run = lambda: pkg.ops.dense_propagate(dev(h0, cuda), dev(A, cuda), P.dense_edge(dW), P.dense_gru(dWg, dWc, D), dbias,
                                          dev(gru["bg"], cuda), dev(gru["bc"], cuda), steps, fmt=fmt)
    got = run()
    want = oracle.dense_propagate(h0, A, W, eb, gru, steps)
    np.testing.assert_allclose(got.cpu().numpy(), want, atol=1e-5, rtol=1e-4)
    assert torch.equal(got, run())
    h = dev(h0, cuda).reshape(b * v, D)
    for _ in range(steps):
        acts = pkg.ops.dense_aggregate(dev(A, cuda), pkg.ops.msg_transform(h, dW), dbias)
        h = pkg.ops.gru([acts], h, dWg, dev(gru["bg"], cuda), dWc, dev(gru["bc"], cuda), "tanh")
    np.testing.assert_allclose(got.cpu().numpy().reshape(b * v, D), h.cpu().numpy(), atol=5e-6, rtol=1e-5)
    assert not pkg.ops.dense_propagate_supported(33, 4, 100) and not pkg.ops.dense_propagate_supported(29, 4, 128)


def _dense_model(pkg, oracle, ms, config=None):
    cfg = {"batch_size": 16}
    cfg.update(config or {})
    model = pkg.DenseGGNNChemModel({"--quiet": True, "--device": "cuda:0", "train_data": ms, "valid_data": ms, "--config": cfg})
    rng = np.random.default_rng(0)
    D, T = model.params["hidden_size"], model.num_edge_types
    W = oracle.glorot_init(rng, [T, D, D])
    b = rng.normal(0, 0.1, [T, 1, D]).astype(np.float32)
    gru = {"Wg": oracle.glorot_init(rng, [2 * D, 2 * D]), "bg": (1 + rng.normal(0, 0.1, 2 * D)).astype(np.float32),
           "Wc": oracle.glorot_init(rng, [2 * D, D]), "bc": rng.normal(0, 0.1, D).astype(np.float32)}
    model.set_graph_weights(W, b, gru)
    return model, W, b, gru


def test_dense_model_matches_oracle(pkg, oracle, cuda):
    ms = pkg.synthetic_qm9(200, mean_nodes=12, seed=5)
    model, W, b, gru = _dense_model(pkg, oracle, ms)
    feeds = list(model.make_minibatch_iterator(model.valid_data, is_training=False))
    assert len(feeds) >= 2 and all(f["num_graphs"] == 16 for f in feeds)       # only full batches (:160-162)
    for feed in feeds[:3]:
        with torch.no_grad():
            loss = model.forward_batch(feed)
            got = model.ops['final_node_representations'].cpu().numpy()
        want = oracle.dense_propagate(feed["initial_node_representation"].cpu().numpy(), feed["adjacency_matrix"].cpu().numpy(),
                                      W, b, gru, model.params["num_timesteps"])
        np.testing.assert_allclose(got, want, atol=1e-5, rtol=1e-4)
        if pkg.formats.split_path() and pkg._lib.load().ggnn_dense_propagate_is_split(int(feed["num_vertices"]), model.num_edge_types, 100):
            assert model.last_format == pkg.formats.F16X2 and model.last_format_bounds["proven"]     # one-hot states, glorot weights
        g = model.weights['regression_gate_task0']; t = model.weights['regression_transform_task0']
        f = lambda x: x.cpu().numpy().astype(np.float64)
        pred = oracle.dense_gated_regression(want, f(feed["initial_node_representation"]), f(feed["node_mask"]),
                                             f(g.params["weights"][0]), f(g.params["biases"][0]),
                                             f(t.params["weights"][0]), f(t.params["biases"][0]))
        wl, _ = oracle.task_loss(pred, f(feed["target_values"])[0], f(feed["target_mask"])[0])
        assert abs(float(loss) - wl) < 1e-5 * max(1.0, abs(wl))


def test_sparse_equals_dense_on_gpu(pkg, oracle, cuda):
    """SURVEY 4.3 on the HIP kernels: sparse model (edge bias, no mean, one 4-step layer, no residuals) ==
    dense model on the real (unpadded) nodes."""
    ms = pkg.synthetic_qm9(64, mean_nodes=8, seed=9)
    dmodel, W, b, gru = _dense_model(pkg, oracle, ms, {"batch_size": 4})
    cfg = {"use_edge_bias": True, "use_edge_msg_avg_aggregation": False, "layer_timesteps": [4], "residual_connections": {}}
    smodel = pkg.SparseGGNNChemModel({"--quiet": True, "--device": "cuda:0", "train_data": ms, "valid_data": ms, "--config": cfg})
    smodel.set_graph_weights([dict(edge_weights=W, edge_biases=b.reshape(b.shape[0], -1), **gru)])
    dfeed = next(iter(dmodel.make_minibatch_iterator(dmodel.valid_data, is_training=False)))
    with torch.no_grad():
        dmodel.feed(dfeed)
        dense = dmodel.compute_final_node_representations().cpu().numpy()
    mask = dfeed["node_mask"].cpu().numpy().astype(bool)
    # the same 4 graphs through the sparse model: recover their ids from the bucket bookkeeping
    data = dmodel.valid_data
    bucket = data["bucket_at_step"][0]
    ids = np.asarray(data["bucketed"][bucket][:4])
    sb = pkg.data.pack_batch(ms, ids, smodel.num_edge_types, smodel.params["hidden_size"])
    sfeed = smodel.to_device_batch(sb)
    with torch.no_grad():
        smodel.feed(sfeed)
        sparse = smodel.compute_final_node_representations().cpu().numpy()
    np.testing.assert_allclose(dense[mask], sparse, atol=2e-6, rtol=1e-5)


def test_dense_model_runs_exact_outside_the_f16x2_range(pkg, oracle, cuda):
    """The dense model's format policy (DenseGGNNChemModel.propagate_format): an edge weight beyond the x 2^8 packing's range, or edge
    weights large enough that v E (D max|W| S + max|b|) leaves f16, select the exact format -- and the result matches the oracle."""
    f = pkg.formats
    if not f.split_path():
        pytest.skip("f32 matrix path")
    ms = pkg.synthetic_qm9(120, mean_nodes=10, seed=6)
    for case in ("big-weight", "unbounded-acts"):
        model, W, b, gru = _dense_model(pkg, oracle, ms)
        W = W.copy()
        if case == "big-weight":
            W[1, 2, 3] = 400.0
        else:
            W *= 150.0                                          # v E D max|W| beyond 65504 for every bucket size (v >= 4: 4 * 4 * 100 * 26)
        model.set_graph_weights(W, b, gru)
        feed = next(iter(model.make_minibatch_iterator(model.valid_data, is_training=False)))
        with torch.no_grad(), f.forced("auto"):
            model.feed(feed)
            got = model.compute_final_node_representations().cpu().numpy()
        if pkg._lib.load().ggnn_dense_propagate_is_split(int(feed["num_vertices"]), model.num_edge_types, 100):
            assert model.last_format == f.BF16X3, (case, model.last_format_bounds)
        want = oracle.dense_propagate(feed["initial_node_representation"].cpu().numpy(), feed["adjacency_matrix"].cpu().numpy(),
                                      W, b, gru, model.params["num_timesteps"])
        if case == "big-weight":
            np.testing.assert_allclose(got, want, atol=2e-5, rtol=1e-4)
        else:       # pre-activations of O(1000) that cancel: ill conditioned in ANY f32 evaluation -- the measure is the oracle evaluated in f32
            want32 = oracle.dense_propagate(feed["initial_node_representation"].cpu().numpy(), feed["adjacency_matrix"].cpu().numpy(),
                                            W, b, gru, model.params["num_timesteps"], dtype=np.float32).astype(np.float64)
            rms = lambda x: float(np.sqrt(np.mean(x * x)))
            assert np.isfinite(got).all() and rms(got - want) <= 2.0 * rms(want32 - want) + 1e-6, (rms(got - want), rms(want32 - want))


def test_dense_model_weighted_adjacency_leaves_the_f16x2_proof(pkg, oracle, cuda):
    """Advisor (round 5): the bound |acts| <= v E (D max|W| S + max|b|) assumes |A| <= 1 (the reference's 0 / 1 matrices,
    chem_tensorflow_dense.py:30-36).  A foreign feed with weighted edges scales the sum: max|A| is measured (cached per tensor and
    version) and multiplies the bound; beyond the format's range the exact format runs and the result matches the oracle (finite,
    where the unchecked fast format returns Inf / NaN)."""
    f = pkg.formats
    if not f.split_path():
        pytest.skip("f32 matrix path")
    ms = pkg.synthetic_qm9(120, mean_nodes=10, seed=7)
    model, W, b, gru = _dense_model(pkg, oracle, ms)
    feed = next(iter(model.make_minibatch_iterator(model.valid_data, is_training=False)))
    v = int(feed["num_vertices"])
    split = bool(pkg._lib.load().ggnn_dense_propagate_is_split(v, model.num_edge_types, 100))
    with torch.no_grad(), f.forced("auto"):
        model.feed(feed)
        model.compute_final_node_representations()
        if split:
            assert model.last_format == f.F16X2 and model.last_format_bounds["adjacency_absmax"] == 1.0
        feed["adjacency_matrix"] = feed["adjacency_matrix"] * 4000.0          # weighted edges: v E D max|W| * 4000 > 65504
        model.feed(feed)
        got = model.compute_final_node_representations().cpu().numpy()
    if split:
        assert model.last_format == f.BF16X3 and model.last_format_bounds["adjacency_absmax"] == 4000.0, model.last_format_bounds
    want = oracle.dense_propagate(feed["initial_node_representation"].cpu().numpy(), feed["adjacency_matrix"].cpu().numpy(),
                                  W, b, gru, model.params["num_timesteps"])
    want32 = oracle.dense_propagate(feed["initial_node_representation"].cpu().numpy(), feed["adjacency_matrix"].cpu().numpy(),
                                    W, b, gru, model.params["num_timesteps"], dtype=np.float32).astype(np.float64)
    rms = lambda x: float(np.sqrt(np.mean(x * x)))
    assert np.isfinite(got).all() and rms(got - want) <= 2.0 * rms(want32 - want) + 1e-6, (rms(got - want), rms(want32 - want))

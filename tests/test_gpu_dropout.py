"""Counter-based dropout (ggnn_dropout_f32; chem_tensorflow_sparse.py:91,113-114, chem_tensorflow_dense.py:104, utils.py:68) on the
MI355X against the NumPy oracle -- bit for bit, it is integer arithmetic up to one correctly rounded division -- and its plumbing
through the training step: weight masks applied once per layer on the accumulated gradient, masks that are functions of
(random_seed, step, site) and of a node's identity."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("rows,cols", [(1, 4), (37, 100), (400, 100), (5, 10), (3, 1), (1000, 256), (0, 100), (129, 7)])
@pytest.mark.parametrize("keep", [0.8, 0.5, 1.0])
def test_dropout_kernel_equals_oracle_bit_for_bit(pkg, oracle, cuda, rows, cols, keep):
    rng = np.random.default_rng(rows * 131 + cols)
    x = rng.uniform(-2, 2, (rows, cols)).astype(np.float32)
    seed = int(rng.integers(0, 2 ** 63)) * 2 + 1                      # exercises the upper key word
    xd = torch.from_numpy(x).to(cuda)
    got = pkg.ops.dropout(xd, keep, seed).cpu().numpy()
    assert np.array_equal(got, oracle.counter_dropout(x, keep, seed))
    keys = rng.integers(-2 ** 40, 2 ** 40, rows).astype(np.int64)
    got = pkg.ops.dropout(xd, keep, seed, row_key=torch.from_numpy(keys).to(cuda)).cpu().numpy()
    assert np.array_equal(got, oracle.counter_dropout(x, keep, seed, row_key=keys))
    got = pkg.ops.dropout(xd, keep, seed, row_key_base=12345678901).cpu().numpy()
    assert np.array_equal(got, oracle.counter_dropout(x, keep, seed, row_key_base=12345678901))
    if rows:                                                          # in place
        pkg.ops.dropout(xd, keep, seed, out=xd)
        assert np.array_equal(xd.cpu().numpy(), oracle.counter_dropout(x, keep, seed))


def test_dropout_backward_rederives_the_mask(pkg, oracle, cuda):
    x = torch.rand(64, 100, device=cuda, requires_grad=True)
    y = pkg.utils.tf_dropout(x, 0.7, seed=5)
    g = torch.rand_like(y)
    y.backward(g)
    mask = oracle.counter_dropout(np.ones((64, 100), np.float32), 0.7, 5)
    assert np.array_equal(x.grad.cpu().numpy(), (g.cpu().numpy() / np.float32(0.7) * (mask != 0)).astype(np.float32))
    assert pkg.utils.tf_dropout(x, 1.0) is x                          # exact identity at keep 1
    with pytest.raises(ValueError):
        pkg.utils.tf_dropout(x, 0.5)                                  # no seed: no silent global generator


def _model(pkg, oracle, cfg, n=80, seed=2):
    ms = pkg.synthetic_qm9(n, mean_nodes=10, seed=seed)
    model = pkg.SparseGGNNChemModel({"--quiet": True, "--device": "cuda:0", "train_data": ms, "valid_data": ms, "--config": dict(cfg)})
    layers = oracle.make_sparse_layers(np.random.default_rng(seed), model.params, model.num_edge_types, random_bias=True)
    model.set_graph_weights(layers)
    feed = dict(next(iter(model.make_minibatch_iterator(model.valid_data, is_training=False))))
    return model, layers, feed


def test_edge_weight_dropout_gradients_match_oracle(pkg, oracle, oracle_torch, cuda):
    """Training forward + hand-written backward under the reference's default edge_weight_dropout_keep_prob = 0.8: the oracle runs on
    the masked weights (mask from the NumPy Philox restatement, same seeds) and its autograd gradient, sent back through the mask,
    must be the package's gradient of the VARIABLE."""
    from test_gpu_train import _oracle_loss_and_grads
    model, layers, feed = _model(pkg, oracle, {"layer_timesteps": [2, 1], "residual_connections": {"1": [0]}})
    feed["edge_weight_dropout_keep_prob"] = 0.8
    feed["out_layer_dropout_keep_prob"] = 1.0
    model.dropout_step = 5
    T, D = model.num_edge_types, model.params["hidden_size"]
    masked, masks = [], []
    for l, L in enumerate(layers):
        W = np.asarray(L["edge_weights"], np.float32).reshape(T * D, D)
        Wm = oracle.counter_dropout(W, 0.8, model.dropout_seed("edge_weights", l))
        masks.append(oracle.counter_dropout(np.ones_like(W), 0.8, model.dropout_seed("edge_weights", l)))
        masked.append(dict(L, edge_weights=Wm.reshape(T, D, D)))
    want_loss, want = _oracle_loss_and_grads(oracle_torch, model, masked, feed)
    variables = model.trainable_variables
    for v in variables.values():
        v.requires_grad_(True); v.grad = None
    model.training = True
    loss = model.forward_batch(feed)
    loss.backward()
    model.training = False
    assert abs(float(loss) - want_loss) < 1e-5 * max(1.0, abs(want_loss))
    for l in range(len(layers)):
        name = "graph_model/gnn_layer_%i/gnn_edge_weights_%i:0" % (l, l)
        w = want[name].numpy() * masks[l]
        got = variables[name].grad.cpu().numpy()
        assert np.abs(got - w).max() <= 2e-4 * np.abs(w).max() + 1e-7
        assert np.array_equal(got == 0, (masks[l] == 0) | (got == 0)) and (got[masks[l] == 0] == 0).all()
    for v in variables.values():
        v.requires_grad_(False); v.grad = None


@pytest.mark.parametrize("keep", [1.0, 0.8])
@pytest.mark.parametrize("cfg", [{}, {"use_edge_bias": True, "layer_timesteps": [2, 1, 2], "residual_connections": {"1": [0], "2": [0, 1]}}])
def test_edge_weight_gradients_go_through_the_sink_and_are_masked_once(pkg, oracle, cuda, cfg, keep, monkeypatch):
    """Advisor finding (round 2): the edge-weight products never reached the side-stream sink (the step saw a [T,D,D] view, the sink
    a [T*D,D] variable).  They do now.  Without weight dropout three steps with and without the sink end in the same weights, bit
    for bit.  Under weight dropout the sink accumulates the RAW products of a layer's timesteps and masks the sum once
    (mask * (dW_1 + dW_2) / keep), autograd sums the masked products (dW_1 / keep + dW_2 / keep on the kept entries): the same
    gradient up to one rounding per addend, and exactly zero on the dropped entries either way."""
    results = []
    monkeypatch.setattr(pkg.backward, "USE_NATIVE_STEP", False)          # (the autograd path is the one with the sink)
    for side in (True, False):
        monkeypatch.setattr(pkg.backward, "USE_WGRAD_STREAM", side)
        model, layers, feed = _model(pkg, oracle, cfg, n=300, seed=4)
        feed["out_layer_dropout_keep_prob"] = 1.0
        feed["edge_weight_dropout_keep_prob"] = keep
        for _ in range(3):
            model.train_batch(feed)
        torch.cuda.synchronize()
        results.append({k: v.detach().clone() for k, v in model.trainable_variables.items()})
        if side:
            used = pkg.backward._SINK.used
            for l in range(len(model.params["layer_timesteps"])):
                assert model._edge_weight_vars[l].data_ptr() in used, "edge weights of layer %d bypassed the sink" % l
            assert not pkg.backward._SINK.masks
    for k in results[0]:
        if keep >= 1.0:
            assert torch.equal(results[0][k], results[1][k]), k
        else:
            # three Adam steps of size ~lr = 1e-3 each: 1 % of one step
            assert float((results[0][k] - results[1][k]).abs().max()) < 1e-5, k


def test_masks_follow_step_and_node_identity(pkg, oracle, cuda):
    """State dropout (chem_tensorflow_sparse.py:113-114): a node's mask is keyed by (dataset graph id, node within graph), so the
    same graphs packed in another order produce the same per-node result; the next optimisation step draws other masks."""
    ms = pkg.synthetic_qm9(60, mean_nodes=9, seed=7)
    cfg = {"layer_timesteps": [2], "residual_connections": {}}
    model = pkg.SparseGGNNChemModel({"--quiet": True, "--device": "cuda:0", "train_data": ms, "valid_data": ms, "--config": cfg})
    msd = model.valid_data["molecules"]
    ids = np.arange(msd.num_graphs)
    outs = []
    for order in (ids, ids[::-1].copy()):
        sb = pkg.data.pack_batch(msd, order, model.num_edge_types, model.params["hidden_size"])
        feed = model.to_device_batch(sb)
        feed["graph_state_keep_prob"] = 0.9; feed["edge_weight_dropout_keep_prob"] = 0.8
        with torch.no_grad():
            model.feed(feed)
            h = model.compute_final_node_representations().cpu().numpy()
        n = np.diff(msd.node_ptr)[order]
        starts = np.concatenate([[0], np.cumsum(n)])[:-1]
        outs.append({int(g): h[s:s + k] for g, s, k in zip(order, starts, n)})
    for g in ids:
        np.testing.assert_allclose(outs[0][int(g)], outs[1][int(g)], rtol=0, atol=0)
    assert any((v == 0).any() for v in outs[0].values())              # the mask did drop state entries
    model.dropout_step += 1
    with torch.no_grad():
        model.feed(feed)
        h2 = model.compute_final_node_representations().cpu().numpy()
    assert not np.array_equal(h2 == 0, h == 0)


def test_dense_model_weight_dropout_masks_each_timestep(pkg, oracle, cuda, monkeypatch):
    """Advisor finding (round 3): the dense model trains ONE shared edge-weight variable under a different mask per timestep
    (chem_tensorflow_dense.py:104).  The gradient sink used to keep a single pending mask per variable and apply the last one
    registered to the SUM of all timesteps' raw products -- mask_0 * sum_i dW_i instead of sum_i mask_i * dW_i.  Now a second,
    different mask settles the pending one and the variable's later contributions are masked one by one: the sink's gradient
    must equal plain autograd's (every product masked where it is formed) to rounding, and the trained weights with it."""
    ms = pkg.synthetic_qm9(120, mean_nodes=9, seed=3)
    grads, weights = [], []
    for side in (True, False):
        monkeypatch.setattr(pkg.backward, "USE_WGRAD_STREAM", side)
        model = pkg.DenseGGNNChemModel({"--quiet": True, "--device": "cuda:0", "train_data": ms, "valid_data": ms,
                                        "--config": {"batch_size": 8, "graph_state_dropout_keep_prob": 0.8}})
        feed = dict(next(iter(model.make_minibatch_iterator(model.train_data, is_training=True))))
        assert feed["edge_weight_dropout_keep_prob"] == 0.8            # (:222-223: fed from graph_state_dropout_keep_prob)
        feed["graph_state_keep_prob"] = 1.0                            # isolate the weight masks
        feed["out_layer_dropout_keep_prob"] = 1.0
        seeds = {model.dropout_seed("edge_weights", i) for i in range(model.params["num_timesteps"])}
        assert len(seeds) == model.params["num_timesteps"]             # one mask per timestep
        W0 = model.weights["edge_weights"].detach().clone()
        model.train_batch(feed)
        torch.cuda.synchronize()
        # Adam's first step is lr * sign-like in g / (|g| + eps): compare the recorded gradient itself where the model keeps it,
        # else the weight update
        weights.append((model.weights["edge_weights"].detach() - W0).cpu().numpy())
        if side:
            assert not any(m is not None for m in pkg.backward._SINK.masks.values())
    upd_sink, upd_auto = weights
    # first Adam step: update = -lr * g / (|g| + 1e-8) -> +-lr where g != 0, 0 where the entry was dropped at EVERY timestep.
    # The old behaviour zeroes every entry the LAST mask drops (20 % of them), this one only those all four masks drop (0.2^4).
    assert np.array_equal(upd_sink == 0, upd_auto == 0)
    T = upd_auto.size // (upd_auto.shape[-1] ** 2)
    blocks = upd_auto.reshape(T, -1)
    present = [t for t in range(T) if (blocks[t] != 0).any()]           # (an edge type without a bond in this batch: zero gradient)
    assert len(present) >= 2 and all((blocks[t] == 0).mean() < 0.02 for t in present)
    np.testing.assert_allclose(upd_sink, upd_auto, atol=2e-6, rtol=0)

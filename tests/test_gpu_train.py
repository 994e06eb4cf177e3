"""GPU tests of the training path: gradients of the HIP forward + hand-written backward against torch
autograd of the CPU oracle (float64), one optimisation step against a restated TF-1.3 Adam with
per-variable clipping, and loss descent."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _setup(pkg, oracle, config, n=60, seed=0):
    ms = pkg.synthetic_qm9(n, mean_nodes=10, seed=seed)
    cfg = {"edge_weight_dropout_keep_prob": 1.0}
    cfg.update(config)
    model = pkg.SparseGGNNChemModel({"--quiet": True, "--device": "cuda:0", "train_data": ms, "valid_data": ms, "--config": cfg})
    layers = oracle.make_sparse_layers(np.random.default_rng(seed), model.params, model.num_edge_types, random_bias=True)
    model.set_graph_weights(layers)
    feed = next(iter(model.make_minibatch_iterator(model.valid_data, is_training=False)))
    return model, layers, feed


def _oracle_loss_and_grads(oracle_torch, model, layers, feed):
    """float64 torch-CPU autograd of the oracle: loss and d loss / d (every variable)."""
    p = model.params
    dd = lambda t: t.detach().cpu().double()
    tl = []
    for L in layers:
        tl.append({k: torch.from_numpy(np.asarray(v)).double().requires_grad_(True) for k, v in L.items()})
    g = model.weights['regression_gate_task0']; t = model.weights['regression_transform_task0']
    ro = [dd(g.params["weights"][0]).requires_grad_(True), dd(g.params["biases"][0]).requires_grad_(True),
          dd(t.params["weights"][0]).requires_grad_(True), dd(t.params["biases"][0]).requires_grad_(True)]
    h0 = dd(feed["initial_node_representation"])
    adj = [a.cpu() for a in feed["adjacency_lists"]]
    last = oracle_torch.sparse_propagate(h0, adj, dd(feed["num_incoming_edges_per_type"]), tl, p)
    pred = oracle_torch.gated_regression(last, h0, feed["graph_nodes_list"].cpu(), feed["num_graphs"], *ro)
    loss, _ = oracle_torch.task_loss(pred, dd(feed["target_values"])[0], dd(feed["target_mask"])[0])
    loss.backward()
    grads = {}
    for l, L in enumerate(tl):
        scope = "graph_model/gnn_layer_%i" % l
        grads["%s/gnn_edge_weights_%i:0" % (scope, l)] = L["edge_weights"].grad.reshape(-1, L["edge_weights"].shape[-1])
        if "edge_biases" in L and p["use_edge_bias"]:
            grads["%s/gnn_edge_biases_%i:0" % (scope, l)] = L["edge_biases"].grad
        base = "%s/timestep_0/gru_cell" % scope
        grads[base + "/gates/kernel:0"] = L["Wg"].grad; grads[base + "/gates/bias:0"] = L["bg"].grad
        grads[base + "/candidate/kernel:0"] = L["Wc"].grad; grads[base + "/candidate/bias:0"] = L["bc"].grad
    grads["out_layer_task0/regression_gate/MLP_W_layer0:0"] = ro[0].grad
    grads["out_layer_task0/regression_gate/MLP_b_layer0:0"] = ro[1].grad
    grads["out_layer_task0/regression/MLP_W_layer0:0"] = ro[2].grad
    grads["out_layer_task0/regression/MLP_b_layer0:0"] = ro[3].grad
    return float(loss), grads


@pytest.mark.parametrize("config", [{}, {"use_edge_bias": True, "graph_rnn_activation": "relu"},
                                    {"use_edge_msg_avg_aggregation": False, "hidden_size": 64,
                                     "layer_timesteps": [2, 1], "residual_connections": {"1": [0]}}])
@pytest.mark.parametrize("compact", [False, True, "unfused-gru-backward"])
@pytest.mark.parametrize("policy", ["auto", "exact"])
def test_gradients_match_oracle_autograd(pkg, oracle, oracle_torch, cuda, config, compact, policy, monkeypatch):
    """compact: the message transform (forward AND backward) on the active (node,type) pairs only vs the dense
    [V, T*D] form; "unfused-gru-backward": the compacted form with the GRU backward as separate launches instead of the
    single fused kernel -- all must reproduce the oracle's autograd gradients."""
    monkeypatch.setattr(pkg.backward, "USE_COMPACT_TRANSFORM", bool(compact))
    monkeypatch.setattr(pkg.formats._local, "policy", policy, raising=False)     # (both operand-format policies of the GRU forward)
    if compact == "unfused-gru-backward":
        monkeypatch.setattr(pkg.ops, "gru_bwd_is_fused", lambda D: False)
    model, layers, feed = _setup(pkg, oracle, config)
    want_loss, want = _oracle_loss_and_grads(oracle_torch, model, layers, feed)
    variables = model.trainable_variables
    for v in variables.values():
        v.requires_grad_(True); v.grad = None
    model.training = True
    loss = model.forward_batch(feed)
    loss.backward()
    model.training = False
    assert abs(float(loss) - want_loss) < 1e-5 * max(1.0, abs(want_loss))
    if pkg.formats.split_path():
        p_ = model.params
        provable = policy == "auto" and p_["graph_rnn_activation"].lower() == "tanh" and p_["use_edge_msg_avg_aggregation"]
        assert model.last_gru_formats == [pkg.formats.F16X2 if provable else pkg.formats.BF16X3] * len(p_["layer_timesteps"])
    assert set(want) == set(variables)
    for name, v in variables.items():
        got = v.grad.detach().cpu().double().reshape(want[name].shape)
        scale = float(want[name].abs().max()) + 1e-12
        err = float((got - want[name]).abs().max())
        assert err <= 2e-4 * scale + 1e-7, (name, err, scale)
        v.requires_grad_(False); v.grad = None


@pytest.mark.parametrize("config", [
    {"hidden_size": 52, "layer_timesteps": [1, 1, 1, 2], "residual_connections": {"2": [0, 1], "3": [0, 1, 2]}},
    {"hidden_size": 100, "layer_timesteps": [1, 1, 1, 1, 1], "residual_connections": {"3": [0, 1, 2], "4": [0, 1, 2, 3]}, "use_edge_bias": True},
    {"hidden_size": 84, "layer_timesteps": [2, 1], "residual_connections": {"1": [0]}, "use_edge_msg_avg_aggregation": False},
    {"hidden_size": 30, "layer_timesteps": [2], "residual_connections": {}},
])
def test_gradients_any_hidden_size_and_residual_fan_in(pkg, oracle, oracle_torch, cuda, config):
    """Training on what the reference accepts beyond the kernels' native shapes: hidden sizes that run zero-padded (the gradient
    of a padded weight block is sliced back to the variable) and layers with 3 / 4 residual inputs (unfused GRU backward, weight
    gradient products over more than 4 column segments) -- against the oracle's float64 autograd."""
    model, layers, feed = _setup(pkg, oracle, config)
    want_loss, want = _oracle_loss_and_grads(oracle_torch, model, layers, dict(feed, initial_node_representation=feed["initial_node_representation"][:, :config["hidden_size"]]))
    variables = model.trainable_variables
    for v in variables.values():
        v.requires_grad_(True); v.grad = None
    model.training = True
    loss = model.forward_batch(feed)
    loss.backward()
    model.training = False
    assert abs(float(loss) - want_loss) < 1e-5 * max(1.0, abs(want_loss))
    assert set(want) == set(variables)
    for name, v in variables.items():
        assert v.grad is not None and tuple(v.grad.shape) == tuple(v.shape), name
        got = v.grad.detach().cpu().double().reshape(want[name].shape)
        scale = float(want[name].abs().max()) + 1e-12
        err = float((got - want[name]).abs().max())
        assert err <= 2e-4 * scale + 1e-7, (name, err, scale)
        v.requires_grad_(False); v.grad = None
    # ... and whole optimisation steps run on it (flat gradient buffer, per-variable clip + Adam): every variable moves by about
    # the learning rate, in the reference's shape
    before = {k: v.detach().clone() for k, v in variables.items()}
    feed = dict(feed); feed["out_layer_dropout_keep_prob"] = 1.0
    losses = [float(model.train_batch(feed)) for _ in range(3)]
    assert np.isfinite(losses).all()
    for k, v in variables.items():
        step = float((v - before[k]).abs().max())
        assert 0 < step < 5e-3 and v.shape == before[k].shape, (k, step)


def test_train_step_matches_restated_tf_adam(pkg, oracle, oracle_torch, cuda):
    """One train step == oracle grads -> per-variable clip_by_norm (chem_tensorflow.py:186-190) -> TF-1.3 Adam."""
    model, layers, feed = _setup(pkg, oracle, {})
    _, grads = _oracle_loss_and_grads(oracle_torch, model, layers, feed)
    before = {k: v.detach().cpu().double().clone() for k, v in model.trainable_variables.items()}
    feed = dict(feed); feed["out_layer_dropout_keep_prob"] = 1.0
    model.train_batch(feed)
    lr, b1, b2, eps, clip = 1e-3, 0.9, 0.999, 1e-8, model.params["clamp_gradient_norm"]
    for name, v in model.trainable_variables.items():
        g = grads[name].reshape(before[name].shape)
        g = g * clip / max(float(g.norm()), clip)
        m = (1 - b1) * g; vv = (1 - b2) * g * g
        lr_t = lr * np.sqrt(1 - b2) / (1 - b1)
        want = before[name] - lr_t * m / (vv.sqrt() + eps)
        got = v.detach().cpu().double()
        # first Adam step = lr * g / (|g| + eps/sqrt(1-b2)): for gradient entries near 3e-7 an fp32-level gradient
        # difference (1e-9) moves the update by ~1e-6; 5e-6 = 0.5 % of the step size lr
        assert float((got - want).abs().max()) < 5e-6, name


@pytest.mark.parametrize("config", [{"use_edge_bias": True}, {"layer_timesteps": [2, 1, 2], "residual_connections": {"1": [0], "2": [0, 1]}}])
def test_side_stream_weight_gradients_are_bit_identical(pkg, oracle, cuda, config, monkeypatch):
    """The fused training step adds the weight gradients of the propagation steps into the optimizer's flat gradient buffer
    on a second stream (backward.weight_gradient_sink) instead of returning them to autograd: three steps from the same
    weights give the same weights, bit for bit, with the sink switched off."""
    results = []
    monkeypatch.setattr(pkg.backward, "USE_NATIVE_STEP", False)          # (the autograd path is the one with the sink)
    for side in (True, False):
        monkeypatch.setattr(pkg.backward, "USE_WGRAD_STREAM", side)
        model, layers, feed = _setup(pkg, oracle, config, n=300, seed=4)
        feed = dict(feed); feed["out_layer_dropout_keep_prob"] = 1.0
        for _ in range(3):
            model.train_batch(feed)
        torch.cuda.synchronize()
        results.append({k: v.detach().clone() for k, v in model.trainable_variables.items()})
        if side:
            assert pkg.backward._SINK.stream is not None and pkg.backward._SINK.targets is None
    for k in results[0]:
        assert torch.equal(results[0][k], results[1][k]), k


def test_threaded_batches_give_identical_epochs(pkg, oracle, cuda):
    """run_epoch packs the training batches on a producer thread and a side stream (utils.ThreadedIterator, the reference's
    chem_tensorflow.py:219): losses and weights after three epochs are bit-identical to packing them inline."""
    ms = pkg.synthetic_qm9(500, mean_nodes=9, seed=5)
    out = []
    for threaded in (True, False):
        np.random.seed(3); torch.manual_seed(3); torch.cuda.manual_seed(3)
        model = pkg.SparseGGNNChemModel({"--quiet": True, "--device": "cuda:0", "train_data": ms, "valid_data": ms,
                                         "--config": {"batch_size": 900, "layer_timesteps": [2, 1], "residual_connections": {"1": [0]},
                                                      "threaded_batches": threaded, "random_seed": 7}})
        losses = [model.run_epoch("train", model.train_data, True)[0] for _ in range(3)]
        torch.cuda.synchronize()
        out.append((losses, {k: v.detach().clone() for k, v in model.trainable_variables.items()}))
    assert out[0][0] == out[1][0]
    for k in out[0][1]:
        assert torch.equal(out[0][1][k], out[1][1][k]), k


def test_training_reduces_loss(pkg, oracle, cuda):
    ms = pkg.synthetic_qm9(400, mean_nodes=9, seed=3)
    model = pkg.SparseGGNNChemModel({"--quiet": True, "--device": "cuda:0", "train_data": ms, "valid_data": ms,
                                     "--config": {"batch_size": 2000, "layer_timesteps": [2, 1],
                                                  "residual_connections": {"1": [0]}, "learning_rate": 3e-3}})
    l0 = model.run_epoch("valid", model.valid_data, False)[0]
    for _ in range(6):
        model.run_epoch("train", model.train_data, True)
    l1 = model.run_epoch("valid", model.valid_data, False)[0]
    assert np.isfinite(l1) and l1 < l0


@pytest.mark.parametrize("config", [
    {"use_propagation_attention": True},
    {"use_propagation_attention": True, "use_edge_bias": True, "use_edge_msg_avg_aggregation": False, "hidden_size": 64},
    {"graph_rnn_cell": "RNN", "graph_rnn_activation": "ReLU"},
    {"graph_rnn_cell": "RNN", "use_edge_bias": True, "layer_timesteps": [2, 1], "residual_connections": {"1": [0]}},
    {"graph_rnn_cell": "CudnnCompatibleGRUCell"},
    {"graph_rnn_cell": "CudnnCompatibleGRUCell", "use_propagation_attention": True, "layer_timesteps": [1, 2],
     "residual_connections": {"1": [0]}},
    {"use_propagation_attention": True, "hidden_size": 128, "layer_timesteps": [2, 1], "residual_connections": {"1": [0]}},
    {"graph_rnn_cell": "RNN", "hidden_size": 192, "layer_timesteps": [1, 1], "residual_connections": {"1": [0]}},
    {"graph_rnn_cell": "CudnnCompatibleGRUCell", "hidden_size": 256, "use_edge_bias": True, "layer_timesteps": [2]},
    # widths that keep their size (multiples of 32 / 100) but have NO compacted transform kernel (these raised
    # NotImplementedError in training): the transform backward takes one generic GEMM per edge type there
    {"use_propagation_attention": True, "hidden_size": 96, "layer_timesteps": [2, 1], "residual_connections": {"1": [0]}},
    {"graph_rnn_cell": "RNN", "hidden_size": 160, "use_edge_bias": True, "layer_timesteps": [2]},
    {"graph_rnn_cell": "CudnnCompatibleGRUCell", "hidden_size": 200, "layer_timesteps": [1, 1], "residual_connections": {"1": [0]}},
    {"graph_rnn_cell": "RNN", "graph_rnn_activation": "ReLU", "hidden_size": 300, "layer_timesteps": [1]},
], ids=["attention", "attention-bias-sum-h64", "rnn-relu", "rnn-bias-residual", "cudnn-gru", "cudnn-gru-attention-residual",
        "attention-h128", "rnn-h192", "cudnn-gru-h256", "attention-h96", "rnn-bias-h160", "cudnn-gru-h200", "rnn-relu-h300"])
def test_variant_hip_backward_equals_autograd_of_torch_restatement(pkg, oracle, cuda, config, monkeypatch):
    """The non-default switches (attention, BasicRNNCell, CudnnCompatibleGRUCell): the hand-written HIP backward against torch
    autograd of the timestep restated in differentiable torch ops (tests/variant_oracle.py)."""
    from importlib import import_module
    import variant_oracle
    variants = import_module(pkg.__name__ + ".variants")
    grads = {}
    for mode in (False, True):
        monkeypatch.setattr(variants, "BACKWARD_ORACLE", variant_oracle.autograd_backward if mode else None)
        model, layers, feed = _setup(pkg, oracle, config, n=80, seed=4)
        variables = model.trainable_variables
        for v in variables.values():
            v.requires_grad_(True); v.grad = None
        model.training = True
        loss = model.forward_batch(feed)
        loss.backward()
        model.training = False
        grads[mode] = {k: v.grad.detach().double().cpu() for k, v in variables.items()}
        grads[(mode, "loss")] = float(loss)
    assert abs(grads[(False, "loss")] - grads[(True, "loss")]) <= 1e-6 * max(1.0, abs(grads[(True, "loss")]))
    for name, want in grads[True].items():
        got = grads[False][name]
        scale = float(want.abs().max()) + 1e-12
        err = float((got - want).abs().max())
        assert err <= 3e-4 * scale + 1e-7, (name, err, scale)


def test_stream_prefetcher_matches_sequential_packing(pkg, oracle, cuda):
    """utils.StreamPrefetcher: batch i+1 assembled on a side stream under batch i's forward, forwards alternating over two compute
    streams -- every batch's final states are bit-identical to packing and running the batches one after the other on one
    stream (three passes, so that the packing stream's allocator pool is recycled under load)."""
    ms = pkg.synthetic_qm9(700, mean_nodes=11, seed=12)
    model = pkg.SparseGGNNChemModel({"--quiet": True, "--device": "cuda:0", "train_data": None, "valid_data": ms,
                                     "--config": {"batch_size": 1100}})
    model.set_graph_weights(oracle.make_sparse_layers(np.random.default_rng(3), model.params, model.num_edge_types, random_bias=True))
    dd = pkg.data_device
    model.prepare_resident_data(model.valid_data, False)
    dms = model.valid_data["molecules_dev"]
    want = []
    with torch.no_grad():
        for fb in dd.pack_batches_device(dms, model.params, model.num_edge_types, None):
            model.feed(fb)
            want.append(model.compute_final_node_representations().cpu())
        assert len(want) >= 6
        streams = [torch.cuda.Stream(), torch.cuda.Stream()]
        for _ in range(3):
            got = []
            gen = dd.pack_batches_device(dms, model.params, model.num_edge_types, None)
            for fb, st in pkg.utils.StreamPrefetcher(gen, cuda, consumer_streams=streams):
                with torch.cuda.stream(st):
                    model.feed(fb)
                    got.append(model.compute_final_node_representations())
            torch.cuda.synchronize()
            assert len(got) == len(want)
            for a, b in zip(got, want):
                assert torch.equal(a.cpu(), b)
        # the product entry point of the same pipeline
        got = [states for _, states, _ in model.forward_dataset(model.valid_data, num_streams=2)]
        torch.cuda.synchronize()
        assert len(got) == len(want) and all(torch.equal(a.cpu(), b) for a, b in zip(got, want))


@pytest.mark.parametrize("config,keeps", [
    ({}, (1.0, 1.0)),                                                                  # the reference's default model
    ({}, (0.8, 0.9)),                                                                  # its training recipe: weight dropout, + readout dropout
    ({"hidden_size": 64, "layer_timesteps": [2, 1, 2], "residual_connections": {"1": [0], "2": [0, 1]},
      "use_edge_msg_avg_aggregation": False, "graph_rnn_activation": "ReLU"}, (0.8, 1.0)),
    ({"layer_timesteps": [1, 2], "residual_connections": {"1": [1, 0]}, "task_ids": [0, 1], "task_sample_ratios": {"1": 0.5}}, (1.0, 1.0)),
])
def test_native_training_step_equals_autograd_path(pkg, oracle, cuda, config, keeps, monkeypatch):
    """train_native (two C calls per step: csrc/ggnn_train.hip) against the torch.autograd path it replaces for the default
    model: the same losses and, after three optimisation steps from the same weights, the same weights -- the two paths launch
    the same kernels on the same operands; only the order in which gradient contributions of shared layer inputs are added
    differs (fp32 rounding)."""
    results = []
    for native in (True, False):
        monkeypatch.setattr(pkg.backward, "USE_NATIVE_STEP", native)
        ms = pkg.synthetic_qm9(300, mean_nodes=10, seed=4, num_tasks=max(config.get("task_ids", [0])) + 1)
        cfg = dict(config)
        model = pkg.SparseGGNNChemModel({"--quiet": True, "--device": "cuda:0", "train_data": ms, "valid_data": ms, "--config": cfg})
        model.set_graph_weights(oracle.make_sparse_layers(np.random.default_rng(4), model.params, model.num_edge_types, random_bias=True))
        feed = dict(next(iter(model.make_minibatch_iterator(model.train_data, is_training=False))))
        feed["edge_weight_dropout_keep_prob"], feed["out_layer_dropout_keep_prob"] = keeps
        assert pkg.train_native.eligible(model, feed) == native
        losses = [float(model.train_batch(feed)) for _ in range(3)]
        torch.cuda.synchronize()
        results.append((losses, {k: v.detach().clone() for k, v in model.trainable_variables.items()},
                        float(model.ops["accuracy_task0"]), model.ops["final_node_representations"].detach().clone()))
    (l1, w1, a1, f1), (l0, w0, a0, f0) = results
    np.testing.assert_allclose(l1, l0, rtol=2e-6)
    assert abs(a1 - a0) <= 2e-6 * max(1.0, abs(a0))
    assert float((f1 - f0).abs().max()) < 2e-5
    for k in w0:
        assert float((w1[k] - w0[k]).abs().max()) < 2e-5, k       # three Adam steps of ~1e-3: <1 % of one step


def test_native_training_step_is_not_taken_by_other_variants(pkg, cuda):
    ms = pkg.synthetic_qm9(40, mean_nodes=8, seed=1)
    for cfg in ({"use_edge_bias": True}, {"graph_rnn_cell": "RNN"}, {"use_propagation_attention": True}, {"hidden_size": 84},
                {"layer_timesteps": [1, 1, 1, 1], "residual_connections": {"3": [0, 1, 2]}}):
        model = pkg.SparseGGNNChemModel({"--quiet": True, "--device": "cuda:0", "train_data": ms, "valid_data": ms, "--config": cfg})
        feed = dict(next(iter(model.make_minibatch_iterator(model.train_data, is_training=False))))
        assert not pkg.train_native.eligible(model, feed), cfg
        assert np.isfinite(float(model.train_batch(feed)))
    model = pkg.SparseGGNNChemModel({"--quiet": True, "--device": "cuda:0", "train_data": ms, "valid_data": ms})
    feed = dict(next(iter(model.make_minibatch_iterator(model.train_data, is_training=False))))
    assert pkg.train_native.eligible(model, feed)
    assert not pkg.train_native.eligible(model, dict(feed, graph_state_keep_prob=0.9))


@pytest.mark.parametrize("keep", [1.0, 0.8])
def test_fused_weight_preparation_equals_per_layer_packing(pkg, oracle, cuda, keep):
    """ggnn_sparse_train_prepare_f32 (all of a step's stage images in one launch, weight-dropout mask on the fly) against the
    per-layer path: ggnn_dropout_f32 on the variable, a transpose copy, ggnn_edge_weights_pack_f32 twice, ggnn_gru_pack_weights_f32,
    the backward's pack -- bit for bit."""
    import ctypes
    ms = pkg.synthetic_qm9(30, mean_nodes=8, seed=1)
    model = pkg.SparseGGNNChemModel({"--quiet": True, "--device": "cuda:0", "train_data": ms, "valid_data": ms})
    model.set_graph_weights(oracle.make_sparse_layers(np.random.default_rng(1), model.params, model.num_edge_types, random_bias=True))
    lib = pkg._lib.load()
    T, D, L = model.num_edge_types, model.params["hidden_size"], len(model.params["layer_timesteps"])
    cells = model.gnn_weights.rnn_cells
    nxs = [len(model.params["residual_connections"].get(str(l)) or []) + 1 for l in range(L)]
    f32 = lambda nbytes: torch.full((nbytes // 4,), float("nan"), device=cuda)
    eb = lib.ggnn_msg_transform_compact_workspace_bytes(D, T)
    imgs = ([f32(eb) for _ in range(L)], [f32(eb) for _ in range(L)], [f32(lib.ggnn_gru_packed_bytes(D, nxs[l])) for l in range(L)],
            [f32(lib.ggnn_gru_bwd_packed_bytes(D, nxs[l])) for l in range(L)])
    ptrs = lambda ts: (ctypes.c_void_p * len(ts))(*[t.data_ptr() for t in ts])
    seeds = [model.dropout_seed("edge_weights", l) for l in range(L)]
    fmts = [(2, 3)[l % 2] for l in range(L)]                   # the GRU forward's operand format is a per-layer argument (ABI 3)
    pkg._lib.check(lib.ggnn_sparse_train_prepare_f32(
        L, T, D, (ctypes.c_int32 * L)(*nxs), ptrs(model._edge_weight_vars), keep, (ctypes.c_uint64 * L)(*seeds),
        ptrs([c.gates_kernel for c in cells]), ptrs([c.candidate_kernel for c in cells]), (ctypes.c_int32 * L)(*fmts), ptrs(imgs[0]),
        ptrs(imgs[1]), ptrs(imgs[2]), ptrs(imgs[3]), torch.cuda.current_stream().cuda_stream))
    packed = pkg.ops.PackedWeights()
    for l in range(L):
        W = model._edge_weight_vars[l].view(T, D, D)
        if keep < 1.0:
            W = pkg.ops.dropout(W.contiguous(), keep, seeds[l])
        want = [packed.edge(W), packed.edge(W.transpose(1, 2).contiguous()),
                packed.gru(cells[l].gates_kernel, cells[l].candidate_kernel, nxs[l], D, fmts[l]),
                packed.gru_bwd(cells[l].gates_kernel, cells[l].candidate_kernel, nxs[l], D)]
        for k in range(4):
            n = want[k].numel() if k >= 2 else (eb - 256) // 4          # (the edge buffers end in 256 bytes of alignment slack)
            if k == 2 and fmts[l] == 2 and pkg.formats.split_path():    # (sized for the larger bf16x3 images: the f16x2 images fill 48 KiB each at D = 100)
                assert D == 100
                n = 3 * (nxs[l] + 1) * 48 * 1024 // 4
            assert torch.equal(imgs[k][l][:n], want[k][:n]), (l, k)


@pytest.mark.gpu
@pytest.mark.parametrize("V,D,nx,T", [(1000, 100, 1, 4), (777, 100, 2, 3), (530, 64, 1, 4), (400, 32, 3, 2), (1, 100, 1, 4)])
def test_gru_backward_with_gathered_gradient_equals_sum_then_backward(pkg, cuda, V, D, nx, T):
    """ggnn_gru_bwd_fused_gather_f32 (the per-node sum dh[v] += sum_t Z[row(v,t)] of the previous timestep's transform backward taken
    while g is loaded) == ggnn_gather_segment_sum_heads_f32(accumulate) into g, then ggnn_gru_bwd_fused_f32 -- bit for bit (the same
    adds in the same order), nodes with 0 .. 4 rows."""
    ops = pkg.ops
    gen = torch.Generator(device="cpu").manual_seed(V + D)
    r = lambda *s: (torch.rand(*s, generator=gen) * 2 - 1).to(cuda)
    g, h, c = r(V, D), r(V, D), r(V, D)
    rr, u = ((r(V, D) + 1) / 2), ((r(V, D) + 1) / 2)
    Wg, Wc = r((nx + 1) * D, 2 * D) * 0.2, r((nx + 1) * D, D) * 0.2
    nin = torch.randint(0, 3, (V, T), generator=gen).float().to(cuda)
    # a node owns between 0 and min(T, 4) rows of Z, rows of one node are not adjacent
    counts = torch.randint(0, min(T, 4) + 1, (V,), generator=gen)
    R = int(counts.sum())
    perm = torch.randperm(max(R, 1), generator=gen)[:R].to(torch.int32)
    row_ptr = torch.zeros(V + 1, dtype=torch.int32); row_ptr[1:] = torch.cumsum(counts, 0).to(torch.int32)
    Z = r(max(R, 1), D)
    heads = torch.full((V, 4), -1, dtype=torch.int32)
    for v in range(V):
        for k in range(int(counts[v])):
            heads[v, k] = perm[int(row_ptr[v]) + k]
    heads = heads.to(cuda)
    packed = ops.PackedWeights().gru_bwd(Wg, Wc, nx, D)
    got = ops.gru_bwd_fused(g, h, rr, u, c, packed, nin, True, nx, "tanh", gather=(Z, heads))
    g2 = g.clone()
    lib = pkg._lib.load()
    # (the device copies must outlive the launch: a temporary's block goes back to the caching allocator as soon as data_ptr() has
    #  returned, and the next temporary can be handed the same block -- nodes with four rows then walk a slot range read from it)
    row_ptr_d, perm_d = row_ptr.to(cuda), perm.to(cuda)
    pkg._lib.check(lib.ggnn_gather_segment_sum_heads_f32(Z.data_ptr(), row_ptr_d.data_ptr(), perm_d.data_ptr(), heads.data_ptr(),
                                                         None, None, 0, g2.data_ptr(), V, D, 1, 1, torch.cuda.current_stream().cuda_stream))
    want = ops.gru_bwd_fused(g2, h, rr, u, c, packed, nin, True, nx, "tanh")
    torch.cuda.synchronize()
    flat = lambda o: [x for x in o[:4]] + list(o[4])
    for a, b in zip(flat(got), flat(want)):
        assert torch.equal(a, b)


@pytest.mark.gpu
def test_native_step_with_stand_alone_node_sums(cuda):
    """The native backward defers dh[v] += sum_t Z[row(v,t)] to the next GRU backward launch (fuse_node_sum, csrc/ggnn_train.hip);
    GGNN_TRAIN_FUSE_NODE_SUM=0 runs the stand-alone sums instead.  The switch is read once per process, so the native-vs-autograd
    tests (default model incl. its residual connections, weight dropout) run again in a process with the other setting: both orders
    of accumulation must reproduce the autograd path's gradients."""
    import subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, GGNN_TRAIN_FUSE_NODE_SUM="0")
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(root, "tests", "test_gpu_train.py"), "-x", "-q", "-m", "gpu", "-k",
                        "test_native_training_step_equals_autograd_path", "-p", "no:cacheprovider"],
                       env=env, capture_output=True, text=True, timeout=900, cwd=root)
    assert r.returncode == 0 and " passed" in r.stdout, (r.stdout[-3000:], r.stderr[-1500:])

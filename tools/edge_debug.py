"""Debug aid for tests/test_gpu_parity.py::test_default_path_is_f32_outside_the_f16x2_operand_range: per case, the error of the GPU
path and of an f32 NumPy evaluation of the oracle against the f64 oracle, and where the largest deviations sit."""
import importlib, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tests"))
pkg = importlib.import_module("gated-graph-neural-network-samples_amd")
import ggnn_oracle as oracle
import test_gpu_parity as T

f = pkg.formats
cases = sys.argv[1:] or ["h0-above-65504-small-weights", "relu-sum-aggregation-hub", "tanh-sum-hub", "huge-edge-weights"]
for case in cases:
    config = {"layer_timesteps": [2, 2, 1], "residual_connections": {"2": [0]}}
    hub = "hub" in case
    if case == "relu-sum-aggregation-hub":
        config.update({"graph_rnn_activation": "ReLU", "use_edge_msg_avg_aggregation": False})
    if case == "tanh-sum-hub":
        config.update({"use_edge_msg_avg_aggregation": False})
    ms = T._hub_molecules(pkg, 3000) if hub else pkg.synthetic_qm9(150, mean_nodes=12, seed=3)
    model, layers, feeds = T._model_and_feed(pkg, oracle, ms, config, seed=11)
    feed = feeds[0]
    L = len(model.params["layer_timesteps"])
    rng = np.random.default_rng(4)
    if case == "h0-above-65504-small-weights":
        h0 = feed["initial_node_representation"].clone()
        V = h0.shape[0]
        h0[torch.arange(0, V, 3), 7] = torch.from_numpy(rng.uniform(7e4, 5e5, len(range(0, V, 3))).astype(np.float32)).to(h0.device)
        feed = dict(feed, initial_node_representation=h0)
        for l in range(L):
            for key in ("edge_weights", "Wg", "Wc"):
                layers[l][key] *= 2e-5
    elif case == "huge-edge-weights":
        for l in range(L):
            layers[l]["edge_weights"] *= 1e4
    model.set_graph_weights(layers)
    print("==", case, "dtype of layers:", layers[0]["Wg"].dtype, layers[0]["edge_weights"].dtype)
    want_all = T._oracle_states.__wrapped__(oracle, feed, layers, model.params) if hasattr(T._oracle_states, "__wrapped__") else None
    adj = [a.cpu().numpy() for a in feed["adjacency_lists"]]
    h0n = feed["initial_node_representation"].cpu().numpy()[:, :model.params["hidden_size"]]
    nin = feed["num_incoming_edges_per_type"].cpu().numpy()
    w64 = oracle.sparse_propagate(h0n, adj, nin, layers, model.params, dtype=np.float64, return_all_layers=True)
    w32 = oracle.sparse_propagate(h0n, adj, nin, layers, model.params, dtype=np.float32, return_all_layers=True)
    for pol in ("auto", "exact"):
        with torch.no_grad(), f.forced(pol):
            model.feed(feed)
            got = model.compute_final_node_representations().cpu().numpy()
        want = w64[-1]
        tol = 1e-5 + 1e-4 * np.abs(want)
        e = np.abs(got - want); e32 = np.abs(w32[-1].astype(np.float64) - want)
        print(" policy", pol, "formats", model.last_gru_formats, "| gpu: viol %d max_abs %.3g | f32-numpy: viol %d max_abs %.3g | max|want| %.3g" % (
            int((e > tol).sum()), e.max(), int((e32 > tol).sum()), e32.max(), np.abs(want).max()))
        idx = np.argsort((e / tol).ravel())[::-1][:6]
        deg = nin.sum(1)
        for i in idx:
            r, c = divmod(int(i), want.shape[1])
            print("    row %d col %d deg %g: gpu %.9g want %.9g f32np %.9g | h0[row,7]=%g" % (r, c, deg[r], got[r, c], want[r, c], w32[-1][r, c], h0n[r, 7]))
    # per-layer: python-loop path states vs oracle (exact policy)
    with torch.no_grad(), f.forced("exact"):
        model.feed(feed)
        ph = model.placeholders
        import types
        states = []
        orig = pkg.ops.sparse_propagate
        def spy(*a, **k):
            outs = orig(*a, **k); states.extend(outs); return outs
        pkg.ops.sparse_propagate = spy
        try:
            model.compute_final_node_representations()
        finally:
            pkg.ops.sparse_propagate = orig
    for l, s in enumerate(states):
        want = w64[l + 1]; e = np.abs(s.cpu().numpy() - want); tol = 1e-5 + 1e-4 * np.abs(want)
        e32 = np.abs(w32[l + 1].astype(np.float64) - want)
        print("   layer %d: gpu viol %d max %.3g | f32np viol %d max %.3g | max|want| %.3g" % (l, int((e > tol).sum()), e.max(), int((e32 > tol).sum()), e32.max(), np.abs(want).max()))

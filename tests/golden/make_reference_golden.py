"""Generates tests/golden/reference_*.npz by RUNNING THE REFERENCE'S OWN PYTHON SOURCE
(/root/reference/chem_tensorflow{,_sparse,_dense}.py, utils.py -- imported from where they lie, unmodified)
on top of the TF-1.3 op shim in oracle/tf13_shim (TensorFlow itself cannot be installed here).

Executed reference code: ChemModel.__init__ (parameter merge, load_data, np.random-seeded initialisation),
process_raw_graphs / make_minibatch_iterator (the batch packer), prepare_specific_graph_model,
compute_final_node_representations (the hot path), gated_regression, the loss, make_train_step
(Adam + per-variable clip_by_norm).  Restated (in the shim, from TF-1.3's published semantics): the TF ops.

    python tests/golden/make_reference_golden.py          # only works where /root/reference exists

Weights are not stored: after construction every variable is overwritten (through the reference's own
``variable.assign``, the mechanism restore_progress uses) by `golden_weights(name, shape, seed)` below, which
tests re-evaluate; the reference's np.random-drawn initial values are kept as per-variable checksums
(`init_stats`) so the package's seeded initialisation can be compared too.
"""
import json
import os
import sys
import tempfile
import zlib

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REFERENCE = "/root/reference"


def golden_weights(name: str, shape, seed: int) -> np.ndarray:
    """Deterministic fp32 value of variable `name`: glorot-range uniform for matrices, U(-0.5, 0.5) for vectors
    (so biases that the reference initialises to 0 / 1 take part in the comparison).  Keyed by the variable name,
    independent of creation order."""
    rng = np.random.default_rng([seed, zlib.crc32(name.encode())])
    shape = tuple(int(s) for s in shape)
    if len(shape) >= 2:
        limit = np.sqrt(6.0 / (shape[-2] + shape[-1]))
        return rng.uniform(-limit, limit, size=shape).astype(np.float32)
    return rng.uniform(-0.5, 0.5, size=shape).astype(np.float32)


def stats(a: np.ndarray):
    a = np.asarray(a, dtype=np.float64)
    return np.array([a.sum(), np.abs(a).sum(), (a * a).sum()])


CASES = {
    # name: (model, params, number of Adam steps recorded)
    "sparse_small": ("sparse", {"layer_timesteps": [2, 1], "residual_connections": {"1": [0]}, "use_edge_bias": True,
                                "batch_size": 120, "random_seed": 3}, 3),
    "sparse_default": ("sparse", {"batch_size": 250, "random_seed": 0}, 2),
    "sparse_sum_agg": ("sparse", {"layer_timesteps": [3], "residual_connections": {}, "use_edge_bias": True,
                                  "use_edge_msg_avg_aggregation": False, "batch_size": 250, "random_seed": 1}, 0),
    "sparse_relu_rnn": ("sparse", {"layer_timesteps": [2, 2], "residual_connections": {"1": [0]},
                                   "graph_rnn_cell": "RNN", "graph_rnn_activation": "ReLU", "batch_size": 250}, 2),
    "sparse_cudnn_gru": ("sparse", {"layer_timesteps": [2, 1], "residual_connections": {"1": [0]},
                                    "graph_rnn_cell": "CudnnCompatibleGRUCell", "batch_size": 250}, 2),
    "sparse_attention": ("sparse", {"layer_timesteps": [2, 1], "residual_connections": {"1": [0]},
                                    "use_propagation_attention": True, "batch_size": 250}, 2),
    # two regression tasks, the second with labels for the first 50 % of the (shuffled) training graphs only
    # (sparse:245-250; target_mask, chem_tensorflow.py:161-169)
    "sparse_multitask": ("sparse", {"layer_timesteps": [2, 1], "residual_connections": {"1": [0]}, "task_ids": [0, 1],
                                    "task_sample_ratios": {"1": 0.5}, "batch_size": 150, "random_seed": 2}, 2),
    # (sparse with tie_fwd_bkwd=False is not a case: the reference raises IndexError in its own packer, sparse:268-272
    #  offsets backward types by the already doubled num_edge_types -- verified by running it under this harness)
    # BASELINE.json configs[0]: chem_tensorflow_sparse.py on 1000 QM9-sized molecules, 1 task (mu), h=100, every parameter at the
    # reference's default (5 layers / 8 steps / residuals, batch_size 100000: the 1000 molecules are ONE batch) -- the reference's
    # CPU plumbing end to end on the JSON schema of get_data.py:82-86; two Adam steps of its own train op on that batch
    "sparse_config0": ("sparse", {"random_seed": 0}, 2, {"train": (1000, 9, 31), "valid": (100, 9, 32)}),
    "dense_default": ("dense", {"batch_size": 4, "random_seed": 5}, 3),     # dense drops incomplete batches (dense:160)
    "dense_untied": ("dense", {"tie_fwd_bkwd": False, "num_timesteps": 2, "batch_size": 5}, 0),
}
WEIGHT_SEED = 20


def run_case(name, kind, params, train_steps, pkg, tf, models, data=None):
    tmp = tempfile.mkdtemp(prefix="ggnn_ref_")
    num_tasks = max(params.get("task_ids", [0])) + 1
    data = data or {"train": (40, 9, 11), "valid": (24, 9, 12)}           # (molecules, mean atoms, generator seed)
    train_ms = pkg.synthetic_qm9(data["train"][0], mean_nodes=data["train"][1], seed=data["train"][2], num_tasks=num_tasks)
    valid_ms = pkg.synthetic_qm9(data["valid"][0], mean_nodes=data["valid"][1], seed=data["valid"][2], num_tasks=num_tasks)
    for fn, ms in (("molecules_train.json", train_ms), ("molecules_valid.json", valid_ms)):
        with open(os.path.join(tmp, fn), "w") as f:
            json.dump(ms.to_json(), f)
    args = {"--data_dir": tmp, "--log_dir": tmp, "--config": json.dumps(params)}
    model = models[kind](args)                                       # the reference's constructor, end to end
    g = model.sess.graph
    trainable = g.get_collection(tf.GraphKeys.TRAINABLE_VARIABLES)
    out = {"params": np.array(json.dumps(model.params)), "kind": np.array(kind), "weight_seed": WEIGHT_SEED,
           "num_edge_types": model.num_edge_types, "annotation_size": model.annotation_size,
           "train_molecules": np.array(json.dumps(train_ms.to_json())),
           "valid_molecules": np.array(json.dumps(valid_ms.to_json())),
           "trainable_names": np.array([v.name for v in trainable]),
           "trainable_shapes": np.array([json.dumps(list(v.value.shape)) for v in trainable]),
           "global_names": np.array([v.name for v in g.get_collection(tf.GraphKeys.GLOBAL_VARIABLES)])}
    # the reference's own initial values (np.random-seeded glorot for edge weights / MLPs; shim RNG for GRU kernels)
    out["init_stats"] = np.stack([stats(model.sess.run(v)) for v in trainable])
    out["init_head"] = np.stack([np.resize(model.sess.run(v).ravel()[:8], 8) for v in trainable])
    # the order in which the reference's data pipeline left the training graphs (np.random.shuffle in process_raw_graphs)
    model.sess.run([v.assign(golden_weights(v.name, v.value.shape, WEIGHT_SEED)) for v in trainable])

    ph = model.placeholders
    fetch = [model.ops["final_node_representations"], model.output, model.ops["loss"],
             model.ops["accuracy_task0"]]

    def record(prefix, feed):
        feed[ph["out_layer_dropout_keep_prob"]] = 1.0
        for key, p in ph.items():
            if key == "adjacency_lists":
                for t, pt in enumerate(p):
                    out["%s_feed_adjacency_%d" % (prefix, t)] = np.asarray(feed[pt])
            elif p in feed:
                out["%s_feed_%s" % (prefix, key)] = np.asarray(feed[p])
        return feed

    nb = 0
    for nb, feed in enumerate(model.make_minibatch_iterator(model.valid_data, False)):
        record("valid%d" % nb, feed)
        h, per_graph, loss, mae = model.sess.run(fetch, feed_dict=feed)
        out["valid%d_final_node_representations" % nb] = h
        out["valid%d_output" % nb] = np.atleast_1d(per_graph)
        out["valid%d_loss" % nb] = loss
        out["valid%d_accuracy" % nb] = mae
    out["num_valid_batches"] = nb + 1

    if train_steps:
        # training batches as the reference's iterator yields them WITHOUT the per-epoch shuffle (is_training=False keeps
        # its order and sets the dropout keep-probabilities to 1); the train op is the reference's own.
        batches = list(model.make_minibatch_iterator(model.train_data, False))
        losses = []
        for s in range(train_steps):
            feed = record("train%d" % s, batches[s % len(batches)])
            loss, _, _ = model.sess.run([model.ops["loss"], model.ops["accuracy_task0"], model.ops["train_step"]],
                                        feed_dict=feed)
            losses.append(loss)
        out["train_losses"] = np.array(losses)
        out["num_train_batches"] = len(batches)
        final = [model.sess.run(v) for v in trainable]
        out["trained_stats"] = np.stack([stats(a) for a in final])
        out["trained_head"] = np.stack([np.resize(a.ravel()[:8], 8) for a in final])
        for v, a in zip(trainable, final):                             # small variables in full (biases, readout)
            if a.size <= 400:
                out["trained/" + v.name] = a
    np.savez_compressed(os.path.join(HERE, "reference_%s.npz" % name), **out)
    print("%-18s %d valid batches, %d variables, loss %.6f" % (name, nb + 1, len(trainable), out["valid0_loss"]))


LOOP_CASES = {
    # the reference's whole train() loop (chem_tensorflow.py:255-307): per-epoch in-place shuffles, training and
    # validation epochs, best-model checkpoint.  Edge-weight dropout off (its mask comes from TF's generator).
    "loop_sparse": ("sparse", {"layer_timesteps": [2, 1], "residual_connections": {"1": [0]}, "batch_size": 100,
                               "num_epochs": 3, "edge_weight_dropout_keep_prob": 1.0, "random_seed": 4}),
    "loop_dense": ("dense", {"batch_size": 4, "num_epochs": 3, "random_seed": 6}),
}


def run_loop_case(name, kind, params, pkg, tf, models):
    import pickle
    tmp = tempfile.mkdtemp(prefix="ggnn_ref_")
    train_ms = pkg.synthetic_qm9(60, mean_nodes=9, seed=21)
    valid_ms = pkg.synthetic_qm9(24, mean_nodes=9, seed=22)
    for fn, ms in (("molecules_train.json", train_ms), ("molecules_valid.json", valid_ms)):
        with open(os.path.join(tmp, fn), "w") as f:
            json.dump(ms.to_json(), f)
    model = models[kind]({"--data_dir": tmp, "--log_dir": tmp, "--config": json.dumps(params)})
    # numpy >= 2 (NEP 50) keeps `np.float32 * int` in float32 where the reference's numpy 1.13 promoted to float64, and
    # json cannot serialise np.float32: give the reference module a json whose dump converts such scalars (its source
    # is untouched; the epoch sums are then accumulated in fp32 instead of fp64, far inside the test tolerance)
    import types
    import chem_tensorflow
    chem_tensorflow.json = types.SimpleNamespace(
        dump=lambda o, f, **kw: json.dump(o, f, default=float, **kw), dumps=json.dumps, load=json.load, loads=json.loads)
    model.train()                                                    # the reference's loop, start to end
    with open(model.log_file) as f:
        log = json.load(f)
    with open(model.best_model_file, "rb") as f:
        best = pickle.load(f)
    names = sorted(best["weights"])
    out = {"params": np.array(json.dumps(model.params)), "kind": np.array(kind),
           "train_molecules": np.array(json.dumps(train_ms.to_json())),
           "valid_molecules": np.array(json.dumps(valid_ms.to_json())),
           "train_loss": np.array([e["train_results"][0] for e in log]),
           "train_accuracy": np.array([e["train_results"][1] for e in log]),
           "train_error_ratio": np.array([e["train_results"][2] for e in log]),
           "valid_loss": np.array([e["valid_results"][0] for e in log]),
           "valid_accuracy": np.array([e["valid_results"][1] for e in log]),
           "best_train_step": best["train_step"], "best_valid_step": best["valid_step"],
           "best_names": np.array(names),
           "best_stats": np.stack([stats(best["weights"][n]) for n in names])}
    np.savez_compressed(os.path.join(HERE, "reference_%s.npz" % name), **out)
    print("%-18s %d epochs, train loss %s, valid loss %s" % (name, len(log), out["train_loss"], out["valid_loss"]))


def main():
    import importlib
    sys.path.insert(0, ROOT)
    pkg = importlib.import_module("gated-graph-neural-network-samples_amd")     # only its synthetic molecule generator
    sys.path[:0] = [os.path.join(ROOT, "oracle", "tf13_shim"), REFERENCE]
    import tensorflow as tf
    assert tf.__version__.endswith("shim")
    cwd = os.getcwd()
    os.chdir(tempfile.mkdtemp(prefix="ggnn_ref_cwd_"))                          # the reference writes logs to cwd-relative paths
    try:
        from chem_tensorflow_dense import DenseGGNNChemModel
        from chem_tensorflow_sparse import SparseGGNNChemModel
        models = {"sparse": SparseGGNNChemModel, "dense": DenseGGNNChemModel}
        only = sys.argv[1:]
        for name, spec in CASES.items():
            if not only or name in only:
                run_case(name, spec[0], spec[1], spec[2], pkg, tf, models, spec[3] if len(spec) > 3 else None)
        for name, (kind, params) in LOOP_CASES.items():
            if not only or name in only:
                run_loop_case(name, kind, params, pkg, tf, models)
    finally:
        os.chdir(cwd)


if __name__ == "__main__":
    main()

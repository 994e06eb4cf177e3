#!/usr/bin/env python
"""Secondary workloads of BASELINE.json (not the headline bench line; bench.py keeps that contract):

    python tools/bench_extra.py dense   # configs[2]: dense-adjacency GGNN, padded batch 256, v = 29, 4 timesteps
    python tools/bench_extra.py large   # configs[4]: one graph, 100k nodes / 1M edges / 4 types, h = 256, 8 steps
    python tools/bench_extra.py train   # configs[1] shapes, full training step (fwd + bwd + clip + Adam)

Each prints one JSON line with wall-clock throughput and per-kernel HIP-event timings.
"""
from __future__ import annotations

import importlib
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("gated-graph-neural-network-samples_amd")
DEV = "cuda:0"


def timed(fn, warmup, steps):
    import gc
    for _ in range(warmup):
        fn()
    gc.collect(); gc.freeze(); gc.disable()        # no 45 ms cyclic-GC pause inside the timed region (see bench.py)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    gc.enable()
    return dt


def kernel_table(fn, reps=6):
    with pkg.ops.kernel_timing() as kt:
        for _ in range(reps):
            fn()
    res = kt.results()
    return {k: {"avg_us": float(np.mean(v)) * 1e3, "launches_per_step": len(v) / reps} for k, v in res.items()}


def dense():
    ms = pkg.synthetic_qm9(4000, mean_nodes=27, seed=0)
    model = pkg.DenseGGNNChemModel({"--quiet": True, "--device": DEV, "train_data": None, "valid_data": ms})
    feeds = [f for f in model.make_minibatch_iterator(model.valid_data, False) if f["num_vertices"] == 29][:4]
    assert feeds, "no full v=29 batch"
    for f in feeds:
        f["initial_node_representation"] = (torch.rand_like(f["initial_node_representation"]) * 2 - 1)
    i = [0]

    def step():
        model.feed(feeds[i[0] % len(feeds)]); i[0] += 1
        with torch.no_grad():
            model.compute_final_node_representations()
    dt = timed(step, 5, 50)
    b, v = feeds[0]["initial_node_representation"].shape[:2]
    print(json.dumps({"workload": "dense GGNN forward, batch %d x v=%d, h=100, 4 edge types, 4 timesteps" % (b, v),
                      "ms_per_step": dt * 1e3, "node_state_updates_per_sec": b * v * model.params["num_timesteps"] / dt,
                      "graphs_per_sec": b / dt, "kernels": kernel_table(step)}))


def large():
    V, M, T, D = 100000, 1000000, 4, 256
    rng = np.random.default_rng(0)
    raw = [{"targets": [[0.0]], "graph": [[0, t + 1, 1] for t in range(T)], "node_features": [[1, 0, 0, 0, 0]] * 2}]
    cfg = {"hidden_size": D, "layer_timesteps": [8], "residual_connections": {}, "tie_fwd_bkwd": True}
    model = pkg.SparseGGNNChemModel({"--quiet": True, "--device": DEV, "train_data": None, "valid_data": raw, "--config": cfg})
    types = rng.integers(0, T, M)
    src = rng.integers(0, V, M).astype(np.int32); dst = rng.integers(0, V, M).astype(np.int32)
    adj = []
    nin = np.zeros((V, T), np.float32)
    for t in range(T):
        a = np.stack([src[types == t], dst[types == t]], 1).astype(np.int32)
        a = a[np.lexsort((a[:, 1], a[:, 0]))]
        np.add.at(nin[:, t], a[:, 1], 1.0)
        adj.append(torch.from_numpy(a).to(DEV))
    feed = {"initial_node_representation": (torch.rand(V, D, device=DEV) * 2 - 1), "adjacency_lists": adj,
            "num_incoming_edges_per_type": torch.from_numpy(nin).to(DEV), "message_index": None}

    def step():
        model.feed(feed)
        with torch.no_grad():
            model.compute_final_node_representations()
        feed["message_index"] = model.placeholders["message_index"]
    dt = timed(step, 3, 20)
    kt = kernel_table(step, 3)
    k2 = kt.get("gather_segment_sum")
    if k2:
        k2["algorithmic_GBps"] = (M * D * 4 + M * 8 + V * D * 4) / (k2["avg_us"] * 1e-6) / 1e9
    print(json.dumps({"workload": "sparse GGNN forward, ONE graph: %d nodes / %d edges / %d types, h=%d, 8 steps" % (V, M, T, D),
                      "ms_per_step": dt * 1e3, "node_state_updates_per_sec": V * 8 / dt, "kernels": kt}))


def train():
    ms = pkg.synthetic_qm9(5700 * 3, mean_nodes=18, seed=0)
    model = pkg.SparseGGNNChemModel({"--quiet": True, "--device": DEV, "train_data": None, "valid_data": ms})
    feeds = list(model.make_minibatch_iterator(model.valid_data, False))[:3]
    for f in feeds:
        f["edge_weight_dropout_keep_prob"] = model.params["edge_weight_dropout_keep_prob"]
        f["out_layer_dropout_keep_prob"] = 1.0
    i = [0]

    def step():
        model.train_batch(feeds[i[0] % len(feeds)]); i[0] += 1
    dt = timed(step, 3, 12)
    V = int(np.mean([f["initial_node_representation"].shape[0] for f in feeds]))
    G = int(np.mean([f["num_graphs"] for f in feeds]))
    print(json.dumps({"workload": "sparse GGNN TRAIN step (fwd+bwd+clip+Adam), %d nodes / %d graphs per batch, h=100" % (V, G),
                      "ms_per_step": dt * 1e3, "node_state_updates_per_sec": V * 8 / dt, "graphs_per_sec": G / dt}))


def pack():
    """The step before the path: one epoch of ~100k-node batches packed by the NumPy packer (+ upload + index build)
    and by the device packer (data_device.py), shuffled graph order as in training."""
    ms = pkg.synthetic_qm9(5700 * 8, mean_nodes=18, seed=0)
    model = pkg.SparseGGNNChemModel({"--quiet": True, "--device": DEV, "train_data": None, "valid_data": ms})
    order = np.random.default_rng(0).permutation(ms.num_graphs)
    lm = model.valid_data["label_mask"]
    T = model.num_edge_types

    def host():
        return [model.to_device_batch(b) for b in pkg.data.pack_batches(ms, model.params, T, order, lm)]
    dms = pkg.data_device.DeviceMoleculeSet(ms, DEV, lm)

    def device():
        return list(pkg.data_device.pack_batches_device(dms, model.params, T, order))
    nb = len(device())
    th = timed(host, 1, 3) / nb
    td = timed(device, 1, 5) / nb
    V = int(np.diff(ms.node_ptr).sum() / nb)
    print(json.dumps({"workload": "batch packing, %d batches of ~%d nodes per epoch" % (nb, V),
                      "host_numpy_pack_upload_index_ms_per_batch": th * 1e3, "device_pack_index_ms_per_batch": td * 1e3,
                      "speedup": th / td}))


if __name__ == "__main__":
    {"dense": dense, "large": large, "train": train, "pack": pack}[sys.argv[1]]()

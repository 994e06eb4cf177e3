#!/usr/bin/env python
"""Secondary workloads of BASELINE.json on their own (bench.py prints the same objects under `secondary` / `train`;
this entry point exists so that rocprofv3 can be pointed at ONE of them -- tools/profile_round.sh does):

    python tools/bench_extra.py dense   # configs[2]: dense-adjacency GGNN, padded batch 256, v = 29, 4 timesteps
    python tools/bench_extra.py large   # configs[4]: one graph, 100k nodes / 1M edges / 4 types, h = 256, 8 steps
    python tools/bench_extra.py train   # configs[1] shapes, full training step (fwd + bwd + clip + Adam)
    python tools/bench_extra.py pack    # the step before the path: batch packing + message-index build
    python tools/bench_extra.py epoch   # training epochs as run_epoch runs them: every batch packed fresh after the shuffle

Each prints one JSON line with wall-clock throughput and per-kernel HIP-event timings.
"""
from __future__ import annotations

import importlib
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench                                                        # noqa: E402  (the workload definitions live there)

pkg = importlib.import_module("gated-graph-neural-network-samples_amd")
DEV = torch.device("cuda:0")
bench.SPLIT_ACTIVE = bool(pkg._lib.load().ggnn_matrix_path_is_split())     # (bench.main() sets it; these legs bypass main)


def dense():
    print(json.dumps(bench.secondary_dense(pkg, DEV)))


def large():
    print(json.dumps(bench.secondary_large(pkg, DEV)))


def train():
    ms = pkg.synthetic_qm9(5700 * 3, mean_nodes=18, seed=0)
    model = pkg.SparseGGNNChemModel({"--quiet": True, "--device": str(DEV), "train_data": None, "valid_data": ms})
    feeds = list(model.make_minibatch_iterator(model.valid_data, False))[:3]
    for f in feeds:
        f["edge_weight_dropout_keep_prob"] = model.params["edge_weight_dropout_keep_prob"]
        f["out_layer_dropout_keep_prob"] = 1.0

    def step(i):
        model.train_batch(feeds[i % len(feeds)])
    dt, n = bench.timed_loop(step, 3, 12, 0.3)
    V = int(np.mean([f["initial_node_representation"].shape[0] for f in feeds]))
    G = int(np.mean([f["num_graphs"] for f in feeds]))
    print(json.dumps({"workload": "sparse GGNN TRAIN step (fwd+bwd+clip+Adam), %d nodes / %d graphs per batch, h=100" % (V, G),
                      "ms_per_step": dt * 1e3, "steps_timed": n, "node_state_updates_per_sec": V * 8 / dt, "graphs_per_sec": G / dt}))


def pack():
    """The step before the path: one epoch of ~100k-node batches packed by the NumPy packer (+ upload + index build)
    and by the device packer (data_device.py), shuffled graph order as in training."""
    ms = pkg.synthetic_qm9(5700 * 8, mean_nodes=18, seed=0)
    model = pkg.SparseGGNNChemModel({"--quiet": True, "--device": str(DEV), "train_data": None, "valid_data": ms})
    order = np.random.default_rng(0).permutation(ms.num_graphs)
    lm = model.valid_data["label_mask"]
    T = model.num_edge_types

    def host(i):
        return [model.to_device_batch(b) for b in pkg.data.pack_batches(ms, model.params, T, order, lm)]
    dms = pkg.data_device.DeviceMoleculeSet(ms, DEV, lm)

    def device(i):
        return list(pkg.data_device.pack_batches_device(dms, model.params, T, order))
    nb = len(device(0))
    th = bench.timed_loop(host, 1, 3)[0] / nb
    td = bench.timed_loop(device, 1, 5)[0] / nb
    V = int(np.diff(ms.node_ptr).sum() / nb)
    print(json.dumps({"workload": "batch packing, %d batches of ~%d nodes per epoch" % (nb, V),
                      "host_numpy_pack_upload_index_ms_per_batch": th * 1e3, "device_pack_index_ms_per_batch": td * 1e3,
                      "speedup": th / td}))


def epoch():
    """Training epochs exactly as ChemModel.run_epoch runs them (chem_tensorflow.py:214-253): shuffle, pack every ~100k-node
    batch fresh on the GPU, train on it -- with the batches packed inline, and by the producer thread on its side stream
    (utils.ThreadedIterator, the reference's chem_tensorflow.py:219)."""
    import time
    NB = 32
    ms = pkg.synthetic_qm9(5700 * NB, mean_nodes=18, seed=0)
    res = {}
    for threaded in (False, True):
        np.random.seed(0); torch.manual_seed(0)
        model = pkg.SparseGGNNChemModel({"--quiet": True, "--device": str(DEV), "train_data": ms, "valid_data": ms,
                                         "--config": {"threaded_batches": threaded}})
        model.run_epoch("warm", model.train_data, True)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        steps = 0
        for _ in range(2):
            steps += model.run_epoch("train", model.train_data, True)[4]
        torch.cuda.synchronize()
        res["threaded" if threaded else "inline"] = (time.perf_counter() - t0) / steps * 1e3
        del model
        torch.cuda.empty_cache()
    V = int(np.diff(ms.node_ptr).sum() / NB)
    print(json.dumps({"workload": "training epochs with fresh batches (shuffle, pack on the GPU, train), ~%d nodes per batch, h=100" % V,
                      "ms_per_step_inline_packing": res["inline"], "ms_per_step_threaded_packing": res["threaded"],
                      "node_state_updates_per_sec_threaded": V * 8 / (res["threaded"] * 1e-3)}))


if __name__ == "__main__":
    {"dense": dense, "large": large, "train": train, "pack": pack, "epoch": epoch}[sys.argv[1]]()

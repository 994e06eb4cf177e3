#!/usr/bin/env python
"""bench.py -- node-state updates/sec of the sparse GGNN propagation hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json configs[1]): sparse GGNN, full-QM9-sized synthetic data (QM9-shaped molecules,
mean 18 atoms incl. hydrogens, 4 edge types), packed as the reference does into super-graph batches of
< 100,000 nodes (chem_tensorflow_sparse.py:44,297), hidden 100, layer_timesteps [2,2,1,2,1] = 8
propagation steps with the default residual connections, mean aggregation on, keep-probs 1.
A "step" = compute_final_node_representations() over ONE such batch (inputs resident in HBM).
metric value = nodes * 8 / time, whole job (all ranks).  Multi-GPU: graphs are independent, so ranks get
disjoint batches and the forward path has no collective ("scaling": "weak", per-GPU batch fixed).

Also on the JSON line:
  roofline      -- the dominant kernel (by measured time): algorithmic flops (or bytes) per launch /
                   its average launch duration, measured live with HIP events on the launch stream.
  kernels       -- the same for every kernel of the path.
  cpu_baseline  -- the torch-CPU fp32 port of the reference op order (oracle/ggnn_oracle_torch.py),
                   timed on this host on rank 0 at N=1, on a bounded sample (one batch, few reps).
"""
from __future__ import annotations

import argparse
import gc
import importlib
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
PKG = "gated-graph-neural-network-samples_amd"

FP32_MFMA_PEAK_TFLOPS = 157.3      # MI355X_MICROARCH.md: v_mfma_f32_16x16x4_f32, dense
HBM_PEAK_GBPS = 8000.0             # MI355X_MICROARCH.md: HBM3E spec (6.29 TB/s measured copy)


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=96)
    ap.add_argument("--warmup", type=int, default=12)
    ap.add_argument("--mean-nodes", type=float, default=18.0, help="mean atoms per molecule (18 = QM9 with H)")
    ap.add_argument("--batches", type=int, default=6, help="distinct resident batches to cycle through")
    ap.add_argument("--streams", type=int, default=2, help="HIP streams the independent batches are issued on")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--cpu-reps", type=int, default=3)
    ap.add_argument("--batch-size", type=int, default=0, help="override the reference's batch_size (100000 nodes); experiments only")
    return ap.parse_args()


def kernel_model(name, V, M, D, T, R=None):
    """Algorithmic flops / bytes per launch -> (bound, work).  R = active (node,type) pairs."""
    if name == "msg_transform":
        return "mfma", 2.0 * V * D * T * D
    if name == "msg_transform_compact":
        return "mfma", 2.0 * R * D * D
    if name == "gather_segment_sum":
        return "hbm", float(M * D * 4 + M * 8 + V * D * 4)
    if name.startswith("gru_fused"):
        nx = int(name.split("nx=")[1].rstrip("]"))
        return "mfma", 6.0 * V * (nx + 1) * D * D
    if name.startswith("gru_gates"):
        nx = int(name.split("nx=")[1].rstrip("]"))
        return "mfma", 2.0 * V * (nx + 1) * D * 2 * D
    if name.startswith("gru_candidate"):
        nx = int(name.split("nx=")[1].rstrip("]"))
        return "mfma", 2.0 * V * (nx + 1) * D * D
    raise KeyError(name)


def kernel_bytes(name, V, M, D, T, R=None):
    """Algorithmic (compulsory) HBM bytes per launch: every operand read once, every result written once."""
    if name == "msg_transform":
        return float(V * D * 4 + V * T * D * 4)
    if name == "msg_transform_compact":
        return float(V * D * 4 + R * 4 + R * D * 4)          # states (each read once) + pair list + compact rows
    if name == "gather_segment_sum":
        return float(M * D * 4 + M * 8 + V * D * 4)
    nx = int(name.split("nx=")[1].rstrip("]"))
    if name.startswith("gru_fused_gather"):                  # residual segments + h + h_out + gathered rows + slots
        return float((nx - 1 + 2) * V * D * 4 + M * D * 4 + M * 4 + V * 4 + V * T * 4)
    if name.startswith("gru_fused"):
        return float((nx + 2) * V * D * 4)
    if name.startswith("gru_gates"):
        return float((nx + 1 + 2) * V * D * 4)
    if name.startswith("gru_candidate"):
        return float((nx + 3 + 1) * V * D * 4)
    raise KeyError(name)


def main():
    args = parse_args()
    pkg = importlib.import_module(PKG)
    dist_ctx = pkg.parallel.DataParallelContext.from_env()
    rank, world = dist_ctx.rank, dist_ctx.world_size
    assert torch.cuda.is_available(), "bench.py needs a GPU (the hot path has no CPU implementation)"
    assert world == args.gpus, "launch with torchrun --nproc-per-node == --gpus"
    dev = dist_ctx.device

    # ---- data: enough QM9-shaped molecules for `batches` distinct ~100k-node batches per rank ------
    mols_per_batch = int(100000 / args.mean_nodes * 1.02) + 8
    ms = pkg.synthetic_qm9(mols_per_batch * args.batches, mean_nodes=args.mean_nodes, seed=1000 + rank)
    model = pkg.SparseGGNNChemModel({"--quiet": True, "--device": str(dev), "train_data": None, "valid_data": ms})
    params = model.params
    if args.batch_size:
        params["batch_size"] = args.batch_size
    D, T = params["hidden_size"], model.num_edge_types
    n_prop = sum(params["layer_timesteps"])
    feeds = list(model.make_minibatch_iterator(model.valid_data, is_training=False))[:args.batches]
    rng = torch.Generator(device="cpu").manual_seed(1234 + rank)
    for f in feeds:   # random dense states: one-hot inputs are sparse and inflate clocks (DVFS)
        f["initial_node_representation"] = (torch.rand(f["initial_node_representation"].shape, generator=rng) * 2 - 1).to(dev)
    nodes = [int(f["initial_node_representation"].shape[0]) for f in feeds]
    msgs = [f["message_index"].num_messages for f in feeds]
    graphs = [int(f["num_graphs"]) for f in feeds]

    # Batches are independent, so consecutive steps may be issued on different HIP streams (--streams N):
    # the tail of one batch's kernel (a partially filled last wave of workgroups) is then back-filled by the
    # next batch's kernels.  Every step still runs the full 8-step forward of one batch; K steps are timed.
    streams = [torch.cuda.Stream(device=dev) for _ in range(args.streams)] if args.streams > 1 else None

    def step(i, multi=True):
        f = feeds[i % len(feeds)]
        if streams is None or not multi:
            model.feed(f)
            return model.compute_final_node_representations()
        with torch.cuda.stream(streams[i % len(streams)]):
            model.feed(f)
            return model.compute_final_node_representations()

    with torch.no_grad():
        if streams is not None:
            for s in streams:
                s.wait_stream(torch.cuda.current_stream())
        for i in range(args.warmup):
            step(i)
        # One generation-2 pass of Python's cyclic garbage collector over the interpreter's long-lived objects
        # (torch, numpy, the model) blocks the host for ~45 ms; the launch queue runs dry and a 70-140 ms timed
        # region reads 20-40 % slow, depending on where the allocation counter happens to trip
        # (tools/stream_jitter.py shows the single gap).  Nothing in the timed region creates reference cycles.
        gc.collect()
        gc.freeze()
        gc.disable()
        torch.cuda.synchronize()
        dist_ctx.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(args.steps):
            step(i)
        torch.cuda.synchronize()
        dist_ctx.barrier()
        torch.cuda.synchronize()
        elapsed = time.perf_counter() - t0
        gc.enable()
    el = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    dist_ctx.all_reduce_max_(el)
    elapsed = float(el.item())
    my_nodes = sum(nodes[i % len(feeds)] for i in range(args.steps))
    my_graphs = sum(graphs[i % len(feeds)] for i in range(args.steps))
    tot = torch.tensor([my_nodes, my_graphs], dtype=torch.float64, device=dev)
    dist_ctx.all_reduce_sum_(tot)
    total_nodes, total_graphs = float(tot[0].item()), float(tot[1].item())
    value = total_nodes * n_prop / elapsed

    out = {
        "metric": "node-state updates/sec on QM9-shaped graphs, h=100, 4 edge types",
        "value": value, "unit": "node-state updates/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": "sparse GGNN forward propagation, full-QM9-sized synthetic batches (configs[1])",
                   "hidden_size": D, "num_edge_types": T, "propagation_steps": n_prop,
                   "layer_timesteps": params["layer_timesteps"], "residual_connections": params["residual_connections"],
                   "nodes_per_batch": int(np.mean(nodes)), "messages_per_batch": int(np.mean(msgs)),
                   "graphs_per_batch": int(np.mean(graphs)), "mean_nodes_per_graph": args.mean_nodes,
                   "active_source_type_pairs_per_batch": None, "hip_streams": max(args.streams, 1),
                   "batch_size_param": params["batch_size"], "parallelism": "dp%d (independent graph batches)" % world},
        "graphs_per_sec": total_graphs / elapsed,
    }

    # ---- roofline leg: per-launch HIP-event timing of every kernel (rank 0) -----------------------------
    if rank == 0 and not args.no_roofline:
        reps = max(4, min(args.steps, 12))
        with torch.no_grad(), pkg.ops.kernel_timing():
            step(0, multi=False)              # untimed: the per-step entry points' first launches (one-off set-up)
        torch.cuda.synchronize()
        with torch.no_grad(), pkg.ops.kernel_timing() as kt:
            for i in range(reps):
                step(i, multi=False)          # single stream: launches must not overlap while they are timed
        res = kt.results()
        Vb, Mb = float(np.mean([nodes[i % len(feeds)] for i in range(reps)])), float(np.mean([msgs[i % len(feeds)] for i in range(reps)]))
        comps = [getattr(f["message_index"], "_compact", None) for f in feeds]
        Rb = float(np.mean([c.num_rows for c in comps])) if all(c is not None for c in comps) else None
        kernels = {}
        for name, times in res.items():
            bound, work = kernel_model(name, Vb, Mb, D, T, Rb)
            avg_ms = float(np.mean(times))
            if bound == "mfma":
                ach, peak, unit = work / (avg_ms * 1e-3) / 1e12, FP32_MFMA_PEAK_TFLOPS, "TFLOP/s"
            else:
                ach, peak, unit = work / (avg_ms * 1e-3) / 1e9, HBM_PEAK_GBPS, "GB/s"
            kernels[name] = {"bound": bound, "achieved": ach, "peak": peak, "unit": unit, "frac": ach / peak,
                             "avg_us": avg_ms * 1e3, "median_us": float(np.median(times)) * 1e3,
                             "max_us": float(np.max(times)) * 1e3, "launches_per_step": len(times) / reps,
                             "time_share": None, "traffic": None,
                             "algorithmic_bytes": kernel_bytes(name, Vb, Mb, D, T, Rb),
                             "hbm_frac": kernel_bytes(name, Vb, Mb, D, T, Rb) / (avg_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS}
        tot_ms = sum(float(np.sum(t)) for t in res.values())
        for name, times in res.items():
            kernels[name]["time_share"] = float(np.sum(times)) / tot_ms
        # HBM traffic per launch cannot be read from inside this process: it comes from the committed rocprofv3
        # PMC passes (separate --pmc FETCH_SIZE / WRITE_SIZE runs; FETCH_SIZE doubled as MI355X_MICROARCH.md
        # prescribes for gfx950), same workload, same kernels -- profiles/*_pmc_summary.json
        pmc_file = os.path.join(ROOT, "profiles", "r01_final_pmc_summary.json")
        if os.path.exists(pmc_file):
            pmc = json.load(open(pmc_file))
            for name in kernels:
                key = {"msg_transform_compact": "msg_transform_compact"}.get(name)
                if name == "gather_segment_sum":
                    key = next((k for k in pmc if k.startswith("gather_segment_sum") and "attn" not in k), None)
                if key is None and name.startswith("gru_fused"):      # template args <D, NX, NW, SAVE, GATHER>
                    nx = name.split("nx=")[1].rstrip("]")
                    tail = "true>" if name.startswith("gru_fused_gather") else "false>"
                    key = next((k for k in pmc if k.startswith("gru_fused<%d, %s," % (D, nx)) and k.endswith(tail)
                                and k.count(",") == 4), None)
                if key in pmc:
                    kernels[name]["traffic"] = pmc[key]["hbm_bytes_fetch_x2_plus_write"]
                    kernels[name]["traffic_source"] = "profiles/r01_final_pmc_summary.json"
        # The edge-indexed scatter-add (chem_tensorflow_sparse.py:198-209) runs INSIDE the GRU launch on the timed path
        # (ggnn_gru_packed_gather_f32).  Its stand-alone kernel -- the one the training path, edge-bias layers and
        # non-fused hidden sizes use -- is timed here over all 8 timesteps of the same batches so that its HBM rate is
        # reported on its own.
        if pkg.ops.FUSE_GATHER:
            saved, pkg.ops.FUSE_GATHER = pkg.ops.FUSE_GATHER, 0
            try:
                with torch.no_grad(), pkg.ops.kernel_timing() as kt2:
                    for i in range(min(reps, 4)):
                        step(i, multi=False)
            finally:
                pkg.ops.FUSE_GATHER = saved
            t2 = kt2.results().get("gather_segment_sum")
            if t2:
                avg_ms = float(np.mean(t2))
                by = kernel_bytes("gather_segment_sum", Vb, Mb, D, T, Rb)
                out["scatter_add"] = {"kernel": "gather_segment_sum", "bound": "hbm", "achieved": by / (avg_ms * 1e-3) / 1e9,
                                      "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": by / (avg_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS,
                                      "avg_us": avg_ms * 1e3, "algorithmic_bytes": by, "in_timed_path": False,
                                      "traffic": None}
                if os.path.exists(pmc_file):       # (kernel variants: gather_segment_sum / gather_segment_sum_flat)
                    key = next((k for k in pmc if k.startswith("gather_segment_sum") and "attn" not in k), None)
                    if key:
                        out["scatter_add"]["traffic"] = pmc[key]["hbm_bytes_fetch_x2_plus_write"]
                        out["scatter_add"]["traffic_source"] = "profiles/r01_final_pmc_summary.json"
        dom = max(kernels, key=lambda k: kernels[k]["time_share"])
        out["roofline"] = dict(kernels[dom], kernel=dom)
        out["config"]["active_source_type_pairs_per_batch"] = None if Rb is None else int(Rb)
        out["kernels"] = kernels
        out["kernel_time_ms_per_step"] = tot_ms / reps

    # ---- CPU baseline leg: torch-CPU port of the reference op order, bounded sample (rank 0, N=1) ---------
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        sys.path.insert(0, os.path.join(ROOT, "oracle"))
        import ggnn_oracle_torch as OT       # the reported baseline; never the thing measured above
        f = feeds[0]
        h0 = f["initial_node_representation"].cpu()
        adj = [a.cpu() for a in f["adjacency_lists"]]
        nin = f["num_incoming_edges_per_type"].cpu()
        layers = []
        for l in range(len(params["layer_timesteps"])):
            c = model.gnn_weights.rnn_cells[l]
            layers.append({"edge_weights": model.gnn_weights.edge_weights[l].cpu(), "Wg": c.gates_kernel.cpu(),
                           "bg": c.gates_bias.cpu(), "Wc": c.candidate_kernel.cpu(), "bc": c.candidate_bias.cpu()})
        with torch.no_grad():
            OT.sparse_propagate(h0, adj, nin, layers, params)          # warm-up
            t0 = time.perf_counter()
            for _ in range(args.cpu_reps):
                ref = OT.sparse_propagate(h0, adj, nin, layers, params)
            cpu_t = (time.perf_counter() - t0) / args.cpu_reps
            model.feed(f)
            got = model.compute_final_node_representations().cpu()
        out["cpu_baseline"] = {"value": nodes[0] * n_prop / cpu_t, "unit": "node-state updates/s",
                               "cores": torch.get_num_threads(), "kind": "port",
                               "sample": "1 batch (%d nodes, %d messages) x %d reps of the 8-step forward, torch-CPU fp32 "
                                         "port of chem_tensorflow_sparse.py:117-218 in reference op order" % (nodes[0], msgs[0], args.cpu_reps),
                               "host_cpus": os.cpu_count(), "graphs_per_sec": graphs[0] / cpu_t,
                               "max_abs_diff_gpu_vs_cpu": float((got - ref).abs().max())}
        out["speedup_vs_cpu_baseline"] = value / out["cpu_baseline"]["value"]

    if rank == 0:
        print(json.dumps(out))
    if world > 1:
        dist_ctx.barrier()          # rank 0 was still busy with the roofline leg: leave together
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()

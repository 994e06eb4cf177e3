#!/usr/bin/env python
"""bench.py -- node-state updates/sec of the sparse GGNN propagation hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--mode forward|train] [--n1-value V1]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Started as plain `python bench.py --gpus N` with N > 1 (no WORLD_SIZE in the environment) it launches its own N ranks through
torch.distributed.run on 127.0.0.1 and exits with their status; `ranks_seen` on the JSON line is the size of the process
group the ranks actually formed and `allreduce_us` the flat gradient all-reduce timed on its own (N > 1).

Workload (BASELINE.json configs[1]): sparse GGNN, full-QM9-sized synthetic data (QM9-shaped molecules,
mean 18 atoms incl. hydrogens, 4 edge types), packed as the reference does into super-graph batches of
< 100,000 nodes (chem_tensorflow_sparse.py:44,297), hidden 100, layer_timesteps [2,2,1,2,1] = 8
propagation steps with the default residual connections, mean aggregation on, keep-probs 1.
A "step" = compute_final_node_representations() over ONE such batch (inputs resident in HBM).
metric value = nodes * 8 / time, whole job (all ranks).  Multi-GPU: graphs are independent, so ranks get
disjoint batches and the forward path has no collective ("scaling": "weak", per-GPU batch fixed).

Timing: every resident batch is run once before anything is timed (whatever --warmup says), then W warm-up steps, then
the K-step loop is timed `timed_repeats` times back to back inside ONE barrier/synchronize bracket -- as often as
it takes to reach --min-time seconds (a 20-step loop is 25 ms, too short to be stable across boxes) -- and
ms_per_step / value are taken over all `steps_timed` = K * timed_repeats steps.

Also on the JSON line:
  roofline      -- the dominant kernel (by measured time): algorithmic flops (or bytes) per launch /
                   its average launch duration, measured live with HIP events on the launch stream; `traffic` from
                   the newest profiles/*_pmc_summary.json, only if it was recorded for the kernel sources in this tree.
  kernels       -- the same for every kernel of the path.
  index_build_ms_per_batch, pack_ms_per_batch, end_to_end_fresh_batch
                -- the work of chem_tensorflow_sparse.py:120-129 / 278-350 that the metric leaves outside the timed
                   region (it is paid once per batch, at pack time) and the rate with it inside.
  train         -- the full optimisation step (forward, backward, gradient all-reduce over RCCL when N > 1, per-variable
                   clip, Adam) on the same batches: ms per step and the all-reduce's share (--mode train makes this
                   the headline instead).
  secondary     -- BASELINE.json configs[2] (dense, padded batch 256 x 29 vertices) and configs[4] (one graph,
                   100k nodes / 1M edges, h = 256): throughput and per-kernel roofline fractions (rank 0, N = 1).
  cpu_baseline  -- the torch-CPU fp32 port of the reference op order (oracle/ggnn_oracle_torch.py),
                   timed on this host on rank 0 at N=1, on a bounded sample (one batch, few reps).
"""
from __future__ import annotations

import argparse
import gc
import socket
import subprocess
import glob
import hashlib
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
BF16_MFMA_PEAK_TFLOPS = 2500.0     # MI355X_MICROARCH.md: bf16 dense (2495 measured)
# Kernels that multiply in 3-way split form (csrc/ggnn_split.hpp: every f32 product = 6 bf16 MFMA products, f32 accumulation)
# when the library runs its default matrix path: their ceiling on the bf16 pipe is its peak / 6 f32-equivalent flops.
SPLIT_PRODUCTS = 6
SPLIT_KERNELS = ("msg_transform_compact", "gru_fused")
SPLIT_ACTIVE = False               # set from ggnn_matrix_path_is_split() in main()
# The fused GRU forward at D = 32 / 64 / 100 multiplies in the TWO-piece f16 form since round 4 (ggnn_gru_forward_format() == 2): three
# f16 MFMA products per f32 product -- its ceiling is the f16 pipe's peak (= the bf16 pipe's) / 3.  So does the column-panel GRU of the
# wider hidden sizes (ggnn_panel.hip); the transforms stay on the six-product bf16 form.
GRU_FWD_FORMAT = 3                 # set from ggnn_gru_forward_format() in main()
EDGE_FORMAT = 3                    # operand format of the compacted message transform of the timed steps (model.last_edge_formats)
F16X2_PRODUCTS = 3
DENSE_SPLIT = False                # set from ggnn_dense_propagate_is_split() for the configs[2] shape in secondary_dense()
DENSE_FORMAT = 3                   # operand format the dense model's policy chose for the configs[2] launches (model.last_format)
HBM_PEAK_GBPS = 8000.0             # MI355X_MICROARCH.md: HBM3E spec (6.29 TB/s measured copy)
HBM_COPY_GBPS = 6290.0


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=96)
    ap.add_argument("--warmup", type=int, default=12)
    ap.add_argument("--mode", choices=("forward", "train"), default="forward",
                    help="forward: the propagation (headline metric); train: the full optimisation step incl. the RCCL all-reduce")
    ap.add_argument("--min-time", type=float, default=0.5, help="minimum duration of the timed region in seconds")
    ap.add_argument("--mean-nodes", type=float, default=18.0, help="mean atoms per molecule (18 = QM9 with H)")
    ap.add_argument("--batches", type=int, default=6, help="distinct resident batches to cycle through")
    ap.add_argument("--streams", type=int, default=2, help="HIP streams the independent batches are issued on")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true", help="skip the configs[2] / configs[4] / train legs")
    ap.add_argument("--cpu-reps", type=int, default=2)
    ap.add_argument("--batch-size", type=int, default=0, help="override the reference's batch_size (100000 nodes); experiments only")
    ap.add_argument("--n1-value", type=float, default=0.0,
                    help="N > 1 runs: the `value` a --gpus 1 run of this script printed on the same node; the line then carries "
                         "weak_scaling_efficiency = value / (N * n1_value) (per-GPU work is fixed: scaling 'weak')")
    ap.add_argument("--dry-run", action="store_true",
                    help="no kernels: rendezvous, per-rank data sharding, the flat gradient all-reduce and the timing bracket only "
                         "(runs on CPU over gloo; exercises the N-rank launch path where there is no GPU)")
    return ap.parse_args()


def respawn_as_ranks(args):
    """`python bench.py --gpus N` without a launcher: start N ranks of this script through torch.distributed.run (one process
    per GPU, rendezvous on 127.0.0.1) and leave with their exit status.  Under torchrun (WORLD_SIZE set) this is a no-op."""
    if args.gpus <= 1 or "WORLD_SIZE" in os.environ:
        return
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")            # dmabuf IPC: RCCL's only working mode on this driver
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or 8) // args.gpus)))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=%d" % args.gpus,
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    print("[bench] no launcher detected: starting %d ranks: %s" % (args.gpus, " ".join(cmd)), file=sys.stderr, flush=True)
    sys.exit(subprocess.call(cmd, env=env))


# ---- algorithmic work per launch -----------------------------------------------------------------------
def kernel_model(name, V, M, D, T, R=None):
    """Algorithmic flops / bytes per launch -> (bound, work).  R = active (node,type) pairs."""
    if name == "msg_transform":
        return "mfma", 2.0 * V * D * T * D
    if name == "msg_transform_compact":
        return "mfma", 2.0 * R * D * D
    if name in ("gather_segment_sum", "dense_aggregate"):
        return "hbm", kernel_bytes(name, V, M, D, T, R)
    if name.startswith("dense_propagate"):                  # the whole dense forward: per timestep T transforms + the GRU (+ aggregation)
        steps = int(name.split("steps=")[1].rstrip("]"))
        return "mfma", steps * (2.0 * V * D * T * D + 12.0 * V * D * D)
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
    if name == "dense_aggregate":                            # adjacency [b,T,v,v] (= M floats here) + transformed rows + output
        return float(M * 4 + V * T * D * 4 + V * D * 4)
    if name.startswith("dense_propagate"):                   # states in and out once, the adjacency tensor once per timestep
        steps = int(name.split("steps=")[1].rstrip("]"))
        return float(2 * V * D * 4 + steps * M * 4)
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


def kernel_table(res, reps, V, M, D, T, R=None):
    """HIP-event timings {name: [ms..]} -> per-kernel roofline records."""
    kernels = {}
    for name, times in res.items():
        try:
            bound, work = kernel_model(name, V, M, D, T, R)
            by = kernel_bytes(name, V, M, D, T, R)
        except (KeyError, IndexError, ValueError):
            continue
        avg_ms = float(np.mean(times))
        if bound == "mfma":
            ach, peak, unit = work / (avg_ms * 1e-3) / 1e12, FP32_MFMA_PEAK_TFLOPS, "TFLOP/s"
        else:
            ach, peak, unit = work / (avg_ms * 1e-3) / 1e9, HBM_PEAK_GBPS, "GB/s"
        kernels[name] = {"bound": bound, "achieved": ach, "peak": peak, "unit": unit, "frac": ach / peak,
                         "avg_us": avg_ms * 1e3, "median_us": float(np.median(times)) * 1e3,
                         "max_us": float(np.max(times)) * 1e3, "launches_per_step": len(times) / reps,
                         "time_share": None, "traffic": None, "algorithmic_bytes": by,
                         "hbm_frac": by / (avg_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS}
        if bound == "mfma" and SPLIT_ACTIVE and (name.startswith(SPLIT_KERNELS) or (name.startswith("dense_propagate") and DENSE_SPLIT)):
            # A split-form kernel (whole-block kernels at D = 32 / 64 / 100, column-panel kernels at 128 / 192 / 256) issues
            # v_mfma_f32_*_bf16: its ceiling is the bf16 pipe's dense peak / 6 products per f32 product, in f32-equivalent flops.
            # `peak` / `frac` are that pipe's; the ratio to the f32-MFMA peak (which such a kernel can exceed) is kept beside them.
            pipe = BF16_MFMA_PEAK_TFLOPS / SPLIT_PRODUCTS
            kernels[name].update({"peak": pipe, "frac": ach / pipe, "pipe": "bf16 MFMA, 6 products per f32 product (2500 / 6 TF f32-equivalent)",
                                  "f32_mfma_peak": FP32_MFMA_PEAK_TFLOPS, "frac_of_f32_mfma_peak": ach / FP32_MFMA_PEAK_TFLOPS,
                                  "matrix_path": "bf16x3 split: f32 operands as 3 bf16 pieces each, 6 bf16 MFMA products per f32 product, "
                                                 "f32 accumulation (error bound of an f32 FMA chain)"})
            if (name.startswith("gru_fused") and GRU_FWD_FORMAT == 2) or (name == "msg_transform_compact" and EDGE_FORMAT == 2) or \
                    (name.startswith("dense_propagate") and DENSE_FORMAT == 2):
                pipe = BF16_MFMA_PEAK_TFLOPS / F16X2_PRODUCTS
                kernels[name].update({"peak": pipe, "frac": ach / pipe, "pipe": "f16 MFMA, 3 products per f32 product (2500 / 3 TF f32-equivalent)",
                                      "matrix_path": "f16x2 split: f32 operands as 2 f16 pieces each (22 of 24 significand bits, round to nearest; "
                                                     "weights packed x 2^8), 3 f16 MFMA products per f32 product, f32 accumulation; error against "
                                                     "f64 below the bf16x3 form's and the f32 MFMA's (tests/test_gpu_split_precision.py)"})
        elif bound == "mfma":
            kernels[name]["pipe"] = "f32 MFMA"
        if bound == "mfma":
            # Which roof is the kernel under?  Its arithmetic intensity (algorithmic flops per algorithmic byte) against the ridge point
            # of the pipe it runs on (pipe peak / HBM peak): below the ridge the HBM roof is the nearer one -- `bound`, `achieved`,
            # `peak`, `unit`, `frac` are then the HBM figures (= hbm_frac) and the matrix pipe's stay beside them as mfma_*.  (The fused
            # GRU at h = 100 has 74-99 flop/B: above the six-product bf16 form's ridge of 52, below the three-product f16 form's 104;
            # the compacted transform's 27 flop/B is below both and above the f32 MFMA's 20.)
            rec = kernels[name]
            rec.update({"arithmetic_intensity": work / by, "ridge_point": rec["peak"] * 1e12 / (HBM_PEAK_GBPS * 1e9),
                        "mfma_achieved": rec["achieved"], "mfma_peak": rec["peak"], "mfma_frac": rec["frac"]})
            if rec["arithmetic_intensity"] < rec["ridge_point"]:
                rec.update({"bound": "hbm", "achieved": by / (avg_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": rec["hbm_frac"]})
    tot_ms = sum(float(np.sum(res[n])) for n in kernels)
    for name in kernels:
        kernels[name]["time_share"] = float(np.sum(res[name])) / tot_ms if tot_ms else None
    return kernels, tot_ms


# ---- HBM traffic: the committed PMC summary of THIS source tree, or nothing ---------------------------------------
def csrc_sha1():
    """Hash of the kernel sources: a PMC summary is only attached to kernels built from the same sources."""
    h = hashlib.sha1()
    for path in sorted(glob.glob(os.path.join(ROOT, PKG, "csrc", "*"))):
        if path.endswith((".hip", ".hpp", ".h")):
            h.update(os.path.basename(path).encode())
            h.update(open(path, "rb").read())
    return h.hexdigest()


def load_pmc(workload):
    """Newest profiles/r*_pmc_summary*.json for `workload` ('bench' | 'large' | 'dense'), with the reason when unusable."""
    suffix = {"bench": "_pmc_summary.json", "large": "_config5_pmc_summary.json", "dense": "_config3_pmc_summary.json"}[workload]
    files = sorted(f for f in glob.glob(os.path.join(ROOT, "profiles", "r*" + suffix))
                   if workload != "bench" or "_config" not in os.path.basename(f))
    if not files:
        return None, None, "no profiles/*%s" % suffix
    path = files[-1]
    pmc = json.load(open(path))
    meta = pmc.get("_meta") or {}
    rel = os.path.relpath(path, ROOT)
    if meta.get("csrc_sha1") != csrc_sha1():
        msg = ("%s was recorded for other kernel sources (csrc sha1 %s, this tree %s): traffic NOT reported -- rerun "
               "tools/profile_round.sh" % (rel, str(meta.get("csrc_sha1"))[:10], csrc_sha1()[:10]))
        print("[bench] WARNING: " + msg, file=sys.stderr)
        return None, rel, msg
    return pmc, rel, None


def pmc_key(pmc, name, D):
    """bench kernel label -> key of tools/pmc_summary.py (short kernel name, template arguments for the fused GRUs)."""
    if name in ("msg_transform_compact", "dense_aggregate"):
        return next((k for k in pmc if k.startswith(name)), None) or \
            (next((k for k in pmc if k.startswith("msg_transform_ring") or k.startswith("msg_transform_panel")), None)
             if name == "msg_transform_compact" else None)
    if name == "msg_transform" or name.startswith("gru_gates") or name.startswith("gru_candidate"):
        # all three are instances of ggnn_gemm_kernel: separable by name only when the workload launched a single one
        # (the dense configs[2]: the message transform is its only GEMM-kernel launch)
        gemms = [k for k in pmc if k.startswith("gemm<")]
        return gemms[0] if name == "msg_transform" and len(gemms) == 1 else None
    if name == "gather_segment_sum":
        return next((k for k in pmc if k.startswith("gather_segment_sum") and "attn" not in k), None)
    if name.startswith("dense_propagate"):
        return next((k for k in pmc if k.startswith("dense_graph")), None)
    if name.startswith("gru_fused"):                     # template args <D, NX, NW, SAVE, GATHER, SPLIT>
        nx = name.split("nx=")[1].rstrip("]")
        gather = "true" if name.startswith("gru_fused_gather") else "false"
        hits = []
        for k in pmc:
            if k.startswith("gru_fused<") and k.endswith(">"):
                t = [x.strip() for x in k[len("gru_fused<"):-1].split(",")]
                if len(t) >= 5 and t[0] == str(D) and t[1] == nx and t[4] == gather:
                    hits.append((len(t) > 5 and t[5] == ("true" if SPLIT_ACTIVE else "false"), pmc[k].get("launches", 0), k))
        if hits:
            return max(hits)[2]                          # the instantiation of this process's matrix path, the most-launched one
        return next((k for k in pmc if k.startswith("gru_panel<%d," % D)), None)
    return None


def attach_traffic(kernels, workload, D):
    pmc, rel, err = load_pmc(workload)
    missing = []
    for name, rec in kernels.items():
        key = pmc_key(pmc, name, D) if pmc else None
        if key and key in pmc:
            rec["traffic"] = pmc[key]["hbm_bytes_fetch_x2_plus_write"]
            rec["traffic_source"] = rel
            rec["mfma_util_pmc"] = pmc[key].get("mfma_util")
        elif pmc is not None and pmc_key({}, name, D) is None and name not in ("msg_transform",) and not name.startswith(("gru_gates", "gru_candidate")):
            missing.append(name)
    if pmc is not None and missing:
        print("[bench] WARNING: %s has no counters for %s (kernel set changed?)" % (rel, missing), file=sys.stderr)
    return err


def timed_loop(fn, warmup, steps, min_time=0.0, sync=torch.cuda.synchronize):
    import gc as _gc
    for i in range(warmup):
        fn(i)
    _gc.collect(); _gc.freeze(); _gc.disable()       # no 45 ms cyclic-GC pause inside the timed region
    sync()
    t0 = time.perf_counter()
    n = 0
    while True:
        for i in range(steps):
            fn(n + i)
        n += steps
        sync()
        dt = time.perf_counter() - t0
        if dt >= min_time:
            break
    _gc.enable()
    return dt / n, n


# ---- secondary workloads (rank 0, N = 1) ---------------------------------------------------------------------------
def secondary_large(pkg, dev):
    """BASELINE.json configs[4]: one graph, 100k nodes / 1M edges / 4 edge types, h = 256, one weight set, 8 steps."""
    V, M, T, D = 100000, 1000000, 4, 256
    adj_np, nin_np = pkg.synthetic_large_graph(V, M, T, seed=0)
    raw = [{"targets": [[0.0]], "graph": [[0, t + 1, 1] for t in range(T)], "node_features": [[1, 0, 0, 0, 0]] * 2}]
    cfg = {"hidden_size": D, "layer_timesteps": [8], "residual_connections": {}, "tie_fwd_bkwd": True}
    model = pkg.SparseGGNNChemModel({"--quiet": True, "--device": str(dev), "train_data": None, "valid_data": raw, "--config": cfg})
    adj = [torch.from_numpy(a).to(dev) for a in adj_np]
    index = pkg.ops.prepare_message_index(pkg.ops.build_message_index(adj, V), D)
    feed = {"initial_node_representation": (torch.rand(V, D, device=dev) * 2 - 1), "adjacency_lists": adj,
            "num_incoming_edges_per_type": torch.from_numpy(nin_np).to(dev), "message_index": index}

    def step(i):
        model.feed(feed)
        with torch.no_grad():
            model.compute_final_node_representations()
    dt, n = timed_loop(step, 3, 10, 0.3)
    with pkg.ops.kernel_timing() as kt:
        for i in range(3):
            step(i)
    comp = getattr(index, "_compact", None)
    R = comp.num_rows if comp is not None else None
    global GRU_FWD_FORMAT, EDGE_FORMAT             # (the formats THIS model's policy chose: the kernels' pipe ceilings follow them)
    gf, ef = list(getattr(model, "last_gru_formats", None) or [3]), list(getattr(model, "last_edge_formats", None) or [3])
    GRU_FWD_FORMAT = gf[0] if SPLIT_ACTIVE and all(x == gf[0] for x in gf) else (3 if SPLIT_ACTIVE else 0)
    EDGE_FORMAT = ef[0] if SPLIT_ACTIVE and all(x == ef[0] for x in ef) else (3 if SPLIT_ACTIVE else 0)
    kernels, tot_ms = kernel_table(kt.results(), 3, V, M, D, T, R)
    err = attach_traffic(kernels, "large", D)
    out = {"workload": "configs[4]: sparse GGNN forward, ONE graph: %d nodes / %d edges / %d edge types, h=%d, 8 steps" % (V, M, T, D),
           "ms_per_step": dt * 1e3, "steps_timed": n, "node_state_updates_per_sec": V * 8 / dt,
           "active_source_type_pairs": R, "kernels": kernels, "kernel_time_ms_per_step": tot_ms / 3}
    k2 = kernels.get("gather_segment_sum")
    if k2:
        out["scatter_add"] = {"achieved_GBps": k2["achieved"], "frac_of_8TBps_spec": k2["frac"],
                              "frac_of_6.29TBps_copy": k2["achieved"] / HBM_COPY_GBPS, "avg_us": k2["avg_us"],
                              "algorithmic_bytes": k2["algorithmic_bytes"], "traffic": k2["traffic"]}
    if err:
        out["traffic_error"] = err
    return out


def secondary_dense(pkg, dev):
    """BASELINE.json configs[2]: dense-adjacency GGNN, padded batch 256 x 29 vertices, h = 100, 4 timesteps."""
    ms = pkg.synthetic_qm9(4000, mean_nodes=27, seed=0)
    model = pkg.DenseGGNNChemModel({"--quiet": True, "--device": str(dev), "train_data": None, "valid_data": ms})
    feeds = [f for f in model.make_minibatch_iterator(model.valid_data, False) if f["num_vertices"] == 29][:4]
    if not feeds:
        return {"error": "no full v=29 batch"}
    for f in feeds:
        f["initial_node_representation"] = (torch.rand_like(f["initial_node_representation"]) * 2 - 1)

    def step(i):
        model.feed(feeds[i % len(feeds)])
        with torch.no_grad():
            model.compute_final_node_representations()
    dt, n = timed_loop(step, 5, 50, 0.3)
    with pkg.ops.kernel_timing() as kt:
        for i in range(6):
            step(i)
    b, v = feeds[0]["initial_node_representation"].shape[:2]
    D, T = model.params["hidden_size"], model.num_edge_types
    global DENSE_SPLIT, DENSE_FORMAT
    DENSE_FORMAT = int(getattr(model, "last_format", 3))
    DENSE_SPLIT = bool(pkg._lib.load().ggnn_dense_propagate_is_split(int(v), int(T), int(D)))
    kernels, tot_ms = kernel_table(kt.results(), 6, b * v, b * T * v * v, D, T)
    err = attach_traffic(kernels, "dense", D)
    out = {"workload": "configs[2]: dense GGNN forward, padded batch %d x v=%d, h=%d, %d edge types, %d timesteps" % (
               b, v, D, T, model.params["num_timesteps"]),
           "ms_per_step": dt * 1e3, "steps_timed": n, "graphs_per_sec": b / dt,
           "node_state_updates_per_sec": b * v * model.params["num_timesteps"] / dt,
           "kernels": kernels, "kernel_time_ms_per_step": tot_ms / 6}
    if err:
        out["traffic_error"] = err
    return out


def measure_sustained_mfma(pkg, dev, launches=60):
    """Dense bf16 MFMA TFLOP/s and shader clock the chip sustains (ggnn_probe_mfma_rate): all-zero operands, random operands, and the
    operand pattern of the 3-way split product.  ~60 launches of ~16 ms each per pattern, the last one reported."""
    import ctypes
    lib = pkg._lib.load()
    nbytes = int(lib.ggnn_probe_mfma_workspace_bytes())
    ws = torch.empty(nbytes // 4 + 64, dtype=torch.float32, device=dev)
    st = torch.cuda.current_stream(dev).cuda_stream
    res = {}
    for name, mode in (("zero_operands", 0), ("random_operands", 1), ("split_pattern", 2)):
        tf, mhz = ctypes.c_double(0.0), ctypes.c_double(0.0)
        pkg._lib.check(lib.ggnn_probe_mfma_rate(mode, launches, ws.data_ptr(), ws.numel() * 4, ctypes.byref(tf), ctypes.byref(mhz), st))
        res[name] = {"tflops_bf16": tf.value, "shader_mhz": mhz.value, "frac_of_2500": tf.value / BF16_MFMA_PEAK_TFLOPS}
    res["what"] = ("whole-chip v_mfma_f32_16x16x32_bf16 stream on register operands, 8 waves per CU, ~1 s per pattern (csrc/ggnn_probe.hip); "
                   "f32-equivalent ceiling of a split-form kernel on this box = split_pattern.tflops_bf16 / %d" % SPLIT_PRODUCTS)
    return res


def child_line(args, env_extra, argv_extra, run=subprocess.run, timeout=180):
    """The compact line of this script run again in a child process (another operand-format policy, another molecule size): the two
    legs then share nothing but the box."""
    env = dict(os.environ, GGNN_BENCH_CHILD="1", **env_extra)
    cmd = [sys.executable, os.path.abspath(__file__), "--gpus", "1", "--steps", str(args.steps), "--warmup", str(args.warmup),
           "--min-time", "0.3", "--streams", str(args.streams), "--batches", str(args.batches), "--no-secondary", "--no-cpu-baseline"] + argv_extra
    r = run(cmd, env=env, capture_output=True, text=True, timeout=timeout)
    return json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])


def exact_format_reference(args, value, run=subprocess.run):
    """The same timed forward with EVERY product in the exact three-piece bf16 format (the host policy forced to `exact` by
    GGNN_GRU_FMT=3 in a child process, so that the two legs share nothing but the box), so that the line carries both numbers of one
    box and one run -- `value` is under the default policy (two-piece f16 operands where their range is proven, formats.py).  This
    is the f32 number of record; the child also times its kernels, so the exact leg has a `roofline` of its own (its dominant
    launch against the six-product pipe).  A reference leg must never take the headline down: any failure comes back as {"error": ...}."""
    try:
        c = child_line(args, {"GGNN_GRU_FMT": "3"}, ["--mean-nodes", str(args.mean_nodes)], run=run)
        return {"what": "the same timed forward in a child process with GGNN_GRU_FMT=3 (policy `exact`): the fused GRU and the message transform in "
                        "the exact three-piece bf16 format, six products per f32 product (the format of every backward kernel)",
                "value": c["value"], "unit": c["unit"], "ms_per_step": c["ms_per_step"], "ms_per_step_one_stream": c.get("ms_per_step_one_stream"),
                "gru_forward_format": c.get("gru_forward_format"), "value_ratio_default_over_exact": value / c["value"],
                "roofline": c.get("roofline")}
    except Exception as exc:
        return {"error": "%s: %s" % (type(exc).__name__, str(exc)[:300])}


def small_molecule_reference(args, run=subprocess.run):
    """BASELINE.json words its workload "~9 nodes" (heavy atoms); the reference's data has hydrogens (mean 18) and the
    headline follows the data.  This leg is the same timed forward on batches of mean-9 molecules (same node cap: twice the
    graphs per batch), in a child process."""
    try:
        c = child_line(args, {}, ["--mean-nodes", "9", "--no-roofline"], run=run)
        cfg = c.get("config") or {}
        return {"value": c["value"], "unit": c["unit"], "ms_per_step": c["ms_per_step"], "mean_nodes_per_graph": 9.0,
                "nodes_per_batch": cfg.get("nodes_per_batch"), "messages_per_batch": cfg.get("messages_per_batch"),
                "graphs_per_batch": cfg.get("graphs_per_batch"), "graphs_per_sec": c.get("graphs_per_sec")}
    except Exception as exc:
        return {"error": "%s: %s" % (type(exc).__name__, str(exc)[:300])}


def compact_line(out, detail_file):
    """The ONE JSON line of the contract, short enough to survive a captured tail: the contract's fields, `roofline` and `cpu_baseline`
    trimmed to their numbers, and the headline figures of the other legs.  Everything else (per-kernel tables, secondary configs,
    sweeps, prose) is in `detail_file` and on stderr."""
    def pick(d, keys):
        return None if not isinstance(d, dict) else {k: d[k] for k in keys if k in d}
    line = {k: out.get(k) for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                                    "vs_baseline", "dtype", "data")}
    cfg = out.get("config") or {}
    line["config"] = pick(cfg, ("workload", "mode", "hidden_size", "num_edge_types", "propagation_steps", "nodes_per_batch",
                                "messages_per_batch", "graphs_per_batch", "mean_nodes_per_graph", "batch_size_param", "hip_streams", "parallelism"))
    of = out.get("operand_format") or {}
    line["operand_format"] = pick(of, ("gru_forward", "gru_forward_per_layer", "message_transform", "policy", "every_other_kernel"))
    if isinstance(of.get("bounds"), dict):
        line["operand_format"]["proven"] = of["bounds"].get("proven")
    ex = out.get("exact_bf16x3_gru_reference")
    line["exact_format_value"] = None if not isinstance(ex, dict) else ex.get("value", ex.get("error"))
    if isinstance(ex, dict) and isinstance(ex.get("roofline"), dict):
        line["roofline_exact"] = pick(ex["roofline"], ("kernel", "bound", "achieved", "peak", "unit", "frac", "avg_us", "mfma_achieved", "mfma_peak",
                                                         "mfma_frac", "arithmetic_intensity", "ridge_point"))
    rf = out.get("roofline")
    if isinstance(rf, dict):
        line["roofline"] = pick(rf, ("kernel", "bound", "achieved", "peak", "unit", "frac", "traffic", "algorithmic_bytes", "avg_us",
                                     "launches_per_step", "time_share", "mfma_achieved", "mfma_peak", "mfma_frac", "mfma_util_pmc",
                                     "arithmetic_intensity", "ridge_point", "traffic_source"))
        line["roofline"]["peak_is"] = "HBM3E spec 8 TB/s (measured copy: 6.29 TB/s)" if rf.get("bound") == "hbm" else "dense MFMA peak of the pipe"
    cb = out.get("cpu_baseline")
    if isinstance(cb, dict):
        line["cpu_baseline"] = pick(cb, ("value", "unit", "cores", "kind", "host_cpus", "max_abs_diff_gpu_vs_cpu"))
        line["cpu_baseline"]["sample"] = str(cb.get("sample", ""))[:160]
    line["speedup_vs_cpu_baseline"] = out.get("speedup_vs_cpu_baseline")
    line["ms_per_step_one_stream"] = out.get("ms_per_step_one_stream")
    tr = out.get("train")
    line["train"] = pick(tr, ("ms_per_step", "value", "n_gpus", "allreduce_ms", "allreduce_share", "error"))
    e2e = out.get("end_to_end_fresh_batch")
    line["end_to_end_fresh_batch"] = pick(e2e, ("value", "one_stream_value", "value_measured_h0", "one_stream_value_measured_h0"))
    sec = out.get("secondary") or {}
    line["secondary"] = {k: pick(v, ("ms_per_step", "node_state_updates_per_sec", "graphs_per_sec", "value", "mean_nodes_per_graph",
                                     "nodes_per_batch", "messages_per_batch", "graphs_per_batch", "error")) for k, v in sec.items()}
    if out.get("weak_scaling_efficiency") is not None:
        line["weak_scaling_efficiency"] = out["weak_scaling_efficiency"]
    line["allreduce_us"] = out.get("allreduce_us")
    line["ranks_seen"] = out.get("ranks_seen")
    line["graphs_per_sec"] = out.get("graphs_per_sec")
    line["detail_file"] = detail_file
    return line


def emit(out):
    """Full record -> gpurun_out/bench_detail.json (best effort) and stderr; the compact line -> stdout, LAST."""
    detail_file = None
    try:
        d = os.path.join(ROOT, "gpurun_out")
        os.makedirs(d, exist_ok=True)
        detail_file = os.path.join("gpurun_out", "bench_detail.json")
        with open(os.path.join(ROOT, detail_file), "w") as fh:
            json.dump(out, fh)
    except Exception:
        detail_file = None
    print("[bench detail] " + json.dumps(out), file=sys.stderr, flush=True)
    print(json.dumps(compact_line(out, detail_file)), flush=True)


def ranks_seen(dist_ctx):
    """Size of the process group the ranks actually formed (1 without one)."""
    return torch.distributed.get_world_size() if torch.distributed.is_initialized() else 1


def dry_run(args, pkg, dist_ctx):
    """--dry-run: everything around the kernels.  Each rank builds ITS shard of a small synthetic dataset with the host packer
    (the rank/world assignment of data.pack_batches), the ranks all-reduce a gradient-sized flat fp32 buffer and run the timing
    bracket (barrier | K no-op steps | barrier, MAX over ranks); rank 0 prints a JSON line with value null."""
    rank, world = dist_ctx.rank, dist_ctx.world_size
    ms = pkg.synthetic_qm9(400, mean_nodes=args.mean_nodes, seed=1000)
    model = pkg.SparseGGNNChemModel({"--quiet": True, "--device": "cpu", "train_data": None, "valid_data": ms, "dist": dist_ctx,
                                     "--config": {"batch_size": args.batch_size or 1200}})
    if world > 1:
        dist_ctx.broadcast_(list(model.named_variables().values()))
    batches = pkg.data.pack_batches(ms, model.params, model.num_edge_types, None, None, rank, world)
    tot = torch.tensor([sum(b.num_nodes for b in batches), sum(b.num_graphs for b in batches), len(batches)], dtype=torch.float64)
    per_rank = torch.zeros(world, dtype=torch.float64); per_rank[rank] = tot[0]
    dist_ctx.all_reduce_sum_(tot)
    dist_ctx.all_reduce_sum_(per_rank)
    variables = list(model.trainable_variables.values())
    grads = [torch.full_like(v, float(rank + 1)) for v in variables]
    ar = []
    for _ in range(5):
        g = list(grads)
        t0 = time.perf_counter(); dist_ctx.reduce_gradients(variables, g); ar.append((time.perf_counter() - t0) * 1e6)
    want = float(sum(r + 1 for r in range(world)))
    assert all(bool((x == want).all()) for x in g), "flat gradient all-reduce returned a wrong sum"
    dist_ctx.barrier(); t0 = time.perf_counter()
    for _ in range(args.steps):
        pass
    dist_ctx.barrier()
    el = torch.tensor([time.perf_counter() - t0], dtype=torch.float64); dist_ctx.all_reduce_max_(el)
    if rank == 0:
        print(json.dumps({"metric": "node-state updates/sec on QM9-shaped graphs, h=100, 4 edge types", "value": None, "dry_run": True,
                          "unit": "node-state updates/s", "n_gpus": world, "ranks_seen": ranks_seen(dist_ctx), "steps": args.steps,
                          "warmup": args.warmup, "scaling": "weak", "data": "synthetic",
                          # (a real run divides its value by world * n1_value: weak_scaling_efficiency; nothing is measured here)
                          "n1_value": args.n1_value, "weak_scaling_efficiency": None,
                          "backend": torch.distributed.get_backend() if torch.distributed.is_initialized() else None,
                          "allreduce_us": float(np.min(ar)) if world > 1 else None,
                          "allreduce_bytes": int(sum(v.numel() for v in variables) * 4),
                          "sharded_nodes_total": float(tot[0]), "sharded_graphs_total": float(tot[1]), "dataset_graphs": ms.num_graphs,
                          "batches_total_incl_padding": float(tot[2]), "bracket_seconds": float(el.item()),
                          "rank_nodes_max_over_min": float(per_rank.max() / per_rank.min().clamp(min=1.0)),
                          "cpu_affinity_rank0": sorted(os.sched_getaffinity(0))[:4] + ["...", len(os.sched_getaffinity(0))]}))
    if world > 1:
        dist_ctx.barrier()
        torch.distributed.destroy_process_group()


def pin_rank_cpus():
    """One process per GPU: each rank keeps to its own slice of the host's CPUs (LOCAL_RANK-th of LOCAL_WORLD_SIZE equal slices of
    the CPUs this process may run on), so that eight ranks' launch threads, packer threads and OpenMP pools do not migrate over
    each other.  GGNN_PIN_CPUS=0 leaves the affinity alone."""
    if os.environ.get("GGNN_PIN_CPUS", "1") == "0" or "LOCAL_RANK" not in os.environ or not hasattr(os, "sched_setaffinity"):
        return
    local, lworld = int(os.environ["LOCAL_RANK"]), int(os.environ.get("LOCAL_WORLD_SIZE", os.environ.get("WORLD_SIZE", "1")))
    cpus = sorted(os.sched_getaffinity(0))
    per = len(cpus) // max(lworld, 1)
    if lworld > 1 and per >= 1:
        mine = cpus[local * per:(local + 1) * per]
        os.sched_setaffinity(0, mine)
        torch.set_num_threads(max(1, min(torch.get_num_threads(), len(mine))))


def main():
    args = parse_args()
    respawn_as_ranks(args)
    pin_rank_cpus()
    pkg = importlib.import_module(PKG)
    dist_ctx = pkg.parallel.DataParallelContext.from_env()
    rank, world = dist_ctx.rank, dist_ctx.world_size
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but the launcher started %d rank(s) (WORLD_SIZE)" % (args.gpus, world))
    if args.dry_run:
        return dry_run(args, pkg, dist_ctx)
    assert torch.cuda.is_available(), "bench.py needs a GPU (the hot path has no CPU implementation)"
    dev = dist_ctx.device
    global SPLIT_ACTIVE, GRU_FWD_FORMAT, EDGE_FORMAT
    SPLIT_ACTIVE = bool(pkg._lib.load().ggnn_matrix_path_is_split())
    GRU_FWD_FORMAT = int(pkg._lib.load().ggnn_gru_forward_format())      # (the policy's default; replaced below by what the timed steps ran)

    # ---- data: enough QM9-shaped molecules for `batches` distinct ~100k-node batches per rank ------
    mols_per_batch = int(100000 / args.mean_nodes * 1.02) + 8
    ms = pkg.synthetic_qm9(mols_per_batch * args.batches, mean_nodes=args.mean_nodes, seed=1000 + rank)
    model = pkg.SparseGGNNChemModel({"--quiet": True, "--device": str(dev), "train_data": None, "valid_data": ms, "dist": dist_ctx})
    if world > 1:
        dist_ctx.broadcast_(list(model.named_variables().values()))     # every rank starts from rank 0's weights
    params = model.params
    if args.batch_size:
        params["batch_size"] = args.batch_size
    D, T = params["hidden_size"], model.num_edge_types
    n_prop = sum(params["layer_timesteps"])
    model.dist = None                       # (the iterator shards by rank; here every rank packs its OWN dataset)
    feeds = list(model.make_minibatch_iterator(model.valid_data, is_training=False))[:args.batches]
    model.dist = dist_ctx
    rng = torch.Generator(device="cpu").manual_seed(1234 + rank)
    for f in feeds:   # random dense states: one-hot inputs are sparse and inflate clocks (DVFS)
        f["initial_node_representation"] = (torch.rand(f["initial_node_representation"].shape, generator=rng) * 2 - 1).to(dev)
        # (uniform in (-1, 1) by construction: the producer's statement of max |h0|, which the operand-format policy of the fused GRU
        #  -- formats.py -- would otherwise measure once per tensor; it is what makes the two-piece f16 format PROVABLY applicable)
        pkg.formats.declare_h0_absmax(f, 1.0)
    nodes = [int(f["initial_node_representation"].shape[0]) for f in feeds]
    msgs = [f["message_index"].num_messages for f in feeds]
    graphs = [int(f["num_graphs"]) for f in feeds]

    # Batches are independent, so consecutive steps may be issued on different HIP streams (--streams N):
    # the tail of one batch's kernel (a partially filled last wave of workgroups) is then back-filled by the
    # next batch's kernels.  Every step still runs the full 8-step forward of one batch; K steps are timed.
    streams = [torch.cuda.Stream(device=dev) for _ in range(args.streams)] if args.streams > 1 else None
    train_feeds = []
    for f in feeds:
        tf = dict(f)
        tf["edge_weight_dropout_keep_prob"] = params["edge_weight_dropout_keep_prob"]      # the reference's training default (0.8)
        tf["out_layer_dropout_keep_prob"] = 1.0
        train_feeds.append(tf)

    def fwd_step(i, multi=True):
        f = feeds[i % len(feeds)]
        if streams is None or not multi:
            model.feed(f)
            return model.compute_final_node_representations()
        with torch.cuda.stream(streams[i % len(streams)]):
            model.feed(f)
            return model.compute_final_node_representations()

    dbg_steps = os.environ.get("GGNN_BENCH_DEBUG", "0") != "0"

    def train_step(i, multi=True):
        if dbg_steps:
            torch.cuda.synchronize(); t0 = time.perf_counter()
            r = model.train_batch(train_feeds[i % len(train_feeds)])
            torch.cuda.synchronize()
            print("[rank %d] train step %d: %.2f ms" % (rank, i, (time.perf_counter() - t0) * 1e3), file=sys.stderr)
            return r
        return model.train_batch(train_feeds[i % len(train_feeds)])

    def timed_region(step, steps, warmup, min_time, no_grad):
        """barrier + synchronize | `repeats` x K steps | synchronize + barrier; MAX over ranks.  The repeat count is
        agreed between the ranks from a calibration pass (one untimed K-step loop)."""
        ctx = torch.no_grad() if no_grad else torch.enable_grad()
        with ctx:
            if streams is not None:
                for s in streams:
                    s.wait_stream(torch.cuda.current_stream())
            for i in range(len(feeds) * (len(streams) if streams else 1)):     # every resident batch, on every stream
                step(i)
            for i in range(warmup):
                step(i)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for i in range(steps):                                              # calibration (also warm-up)
                step(i)
            torch.cuda.synchronize()
            cal = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=dev)
            dist_ctx.all_reduce_max_(cal)
            repeats = max(1, int(np.ceil(min_time / max(float(cal.item()), 1e-6))))
            # One generation-2 pass of Python's cyclic garbage collector over the interpreter's long-lived objects
            # (torch, numpy, the model) blocks the host for ~45 ms; the launch queue runs dry and a short timed
            # region reads 20-40 % slow (tools/stream_jitter.py).  Nothing in the timed region creates reference cycles.
            gc.collect()
            gc.freeze()
            gc.disable()
            torch.cuda.synchronize()
            dist_ctx.barrier()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(repeats):
                for i in range(steps):
                    step(i)
            torch.cuda.synchronize()
            dist_ctx.barrier()
            torch.cuda.synchronize()
            elapsed = time.perf_counter() - t0
            gc.enable()
        el = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist_ctx.all_reduce_max_(el)
        return float(el.item()), repeats

    def totals(steps_timed):
        my_nodes = sum(nodes[i % len(feeds)] for i in range(args.steps)) * (steps_timed // args.steps)
        my_graphs = sum(graphs[i % len(feeds)] for i in range(args.steps)) * (steps_timed // args.steps)
        tot = torch.tensor([my_nodes, my_graphs], dtype=torch.float64, device=dev)
        dist_ctx.all_reduce_sum_(tot)
        return float(tot[0].item()), float(tot[1].item())

    headline_train = args.mode == "train"
    elapsed, repeats = timed_region(train_step if headline_train else fwd_step, args.steps, args.warmup, args.min_time,
                                    no_grad=not headline_train)
    steps_timed = args.steps * repeats
    total_nodes, total_graphs = totals(steps_timed)
    value = total_nodes * n_prop / elapsed
    # the operand format the model's policy chose for the fused GRU forward of the timed steps (formats.py: per layer, per launch)
    fmts = list(getattr(model, "last_gru_formats", None) or [])
    GRU_FWD_FORMAT = (fmts[0] if fmts and all(x == fmts[0] for x in fmts) else 3) if SPLIT_ACTIVE else 0
    efmts = list(getattr(model, "last_edge_formats", None) or [])
    EDGE_FORMAT = (efmts[0] if efmts and all(x == efmts[0] for x in efmts) else 3) if SPLIT_ACTIVE else 0
    operand_format = {
        "gru_forward": ("f32 MFMA (GGNN_MATRIX=f32)" if not SPLIT_ACTIVE else
                        (pkg.formats.NAMES.get(fmts[0]) if fmts and all(x == fmts[0] for x in fmts) else "mixed")),
        "gru_forward_per_layer": [pkg.formats.NAMES.get(x) for x in fmts],
        "message_transform": ("f32 MFMA (GGNN_MATRIX=f32)" if not SPLIT_ACTIVE else
                              (pkg.formats.NAMES.get(efmts[0]) if efmts and all(x == efmts[0] for x in efmts) else "mixed")),
        "policy": pkg.formats.policy(),
        "selected_by": "formats.py, per launch: f16x2 only where |w| <= 255.875 and |a| <= 65504 are PROVEN from max|h0|, the weights' "
                       "maxima, the tanh cell and mean aggregation; bf16x3 (exact) otherwise",
        "bounds": getattr(model, "last_gru_format_bounds", None),
        "every_other_kernel": "bf16x3 (exact: every backward kernel, the dense and h = 128..256 transforms)" if SPLIT_ACTIVE else "f32 MFMA"}

    # The headline issues consecutive batches on `--streams` HIP streams (one batch's kernel tails are back-filled by the next
    # batch's launches).  The same loop on ONE stream, timed the same way: the difference is that overlap, and it is why the
    # per-launch kernel times of the roofline leg (one stream, launches strictly in sequence) can add up to more than ms_per_step.
    one_stream = None
    if not headline_train and streams is not None:
        el1, rep1 = timed_region(lambda i: fwd_step(i, multi=False), args.steps, 2, min(args.min_time, 0.25), no_grad=True)
        one_stream = el1 / (args.steps * rep1) * 1e3

    what = ("sparse GGNN TRAINING step (forward + backward + gradient all-reduce + per-variable clip + Adam)" if headline_train
            else "sparse GGNN forward propagation")
    out = {
        "metric": "node-state updates/sec on QM9-shaped graphs, h=100, 4 edge types" + (" (training step)" if headline_train else ""),
        "value": value, "unit": "node-state updates/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": elapsed / steps_timed * 1e3, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "matrix_path": ((("fused GRU forward: f16x2 split (f32 in, f32 accumulate; every f32 operand as two f16 pieces = 22 of its 24 significand "
                          "bits, 3 f16 MFMA products per f32 product; measured error against f64 below the six-product form's and the f32 "
                          "MFMA's INSIDE the operand range the host policy proves per launch, bf16x3 otherwise; GGNN_GRU_FMT=3: bf16x3 always); " if GRU_FWD_FORMAT == 2 else "") +
                         "bf16x3 split (f32 in, f32 accumulate; every f32 product = 6 exact bf16 MFMA products of 3-way split operands; "
                         "GGNN_MATRIX=f32 selects the f32 MFMA kernels)") if SPLIT_ACTIVE else "f32 MFMA"),
        "gru_forward_format": {2: "f16x2", 3: "bf16x3", 0: "f32"}.get(GRU_FWD_FORMAT), "operand_format": operand_format,
        "ranks_seen": ranks_seen(dist_ctx), "allreduce_us": None,
        "steps_timed": steps_timed, "timed_repeats": repeats, "timed_seconds": elapsed,
        "ms_per_step_one_stream": one_stream,
        # (N > 1 with --n1-value: the curve's point without a second run of arithmetic elsewhere)
        "weak_scaling_efficiency": (value / (world * args.n1_value)) if (world > 1 and args.n1_value > 0) else None,
        "config": {"workload": what + ", full-QM9-sized synthetic batches (configs[%d])" % (3 if world > 1 else 1),
                   "mode": args.mode, "hidden_size": D, "num_edge_types": T, "propagation_steps": n_prop,
                   "layer_timesteps": params["layer_timesteps"], "residual_connections": params["residual_connections"],
                   "nodes_per_batch": int(np.mean(nodes)), "messages_per_batch": int(np.mean(msgs)),
                   "graphs_per_batch": int(np.mean(graphs)), "mean_nodes_per_graph": args.mean_nodes,
                   "active_source_type_pairs_per_batch": None, "hip_streams": 1 if headline_train else max(args.streams, 1),
                   "resident_batches": len(feeds), "batch_size_param": params["batch_size"],
                   # (data-parallel epochs of run_epoch are re-cut into equal-node batches by default, data.epoch_boundaries; the bench's
                   #  ranks pack their own datasets with the reference's greedy batcher, so the headline's batches are the reference's)
                   "dp_balance_nodes": bool(params.get("dp_balance_nodes", True)),
                   "parallelism": "dp%d (independent graph batches%s)" % (world, "; one flat fp32 gradient all-reduce per step" if headline_train else "")},
        "graphs_per_sec": total_graphs / elapsed,
    }

    # ---- index prep / packing outside the timed region, and the rate with them inside (rank 0) ----------------------
    # (before the training leg: after a few seconds of training steps the same leg measures ~4 % lower -- clocks, not code)
    if rank == 0 and not headline_train and not os.environ.get("GGNN_BENCH_CHILD"):
        ops = pkg.ops
        for warm in (True, False):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for f in feeds:
                ops.prepare_message_index(ops.build_message_index(f["adjacency_lists"], f["initial_node_representation"].shape[0],
                                                                  validate=False), D)
            torch.cuda.synchronize()
            idx_ms = (time.perf_counter() - t0) / len(feeds) * 1e3
        dms = model.valid_data["molecules_dev"]
        from importlib import import_module
        dd = import_module(PKG + ".data_device")
        list(dd.pack_batches_device(dms, params, T, None))                         # warm
        torch.cuda.synchronize(); t0 = time.perf_counter()
        packed = list(dd.pack_batches_device(dms, params, T, None))
        torch.cuda.synchronize()
        pack_ms = (time.perf_counter() - t0) / len(packed) * 1e3
        del packed
        e2e_streams = [None] * max(args.streams, 1)
        e2e_packs = [None, None]
        # the end-to-end leg runs whole epochs over a FULL-QM9-sized dataset (133,885 molecules, ~25 batches: configs[1]), so that
        # the pipeline's fill and drain weigh what they weigh in an epoch of the real dataset (6-batch epochs overstate them 4x)
        ms_full = pkg.synthetic_qm9(133885, mean_nodes=args.mean_nodes, seed=2000 + rank)
        dms_e2e = dd.DeviceMoleculeSet(ms_full, dev, None)
        dms_e2e.static_tables(T, params.get("tie_fwd_bkwd", True), pkg.ops.compact_supported(D))
        e2e_batches = len(pkg.data.batch_boundaries(dms_e2e.nodes_per_graph, params["batch_size"])) - 1

        e2e_data = {"molecules": ms_full, "molecules_dev": dms_e2e, "label_mask": None}
        pool = torch.rand((max(params["batch_size"], max(nodes)), D), device=dev) * 2 - 1     # dense random states, as above

        def dense_states(fb):
            fb["initial_node_representation"] = pool[:fb["initial_node_representation"].shape[0]]
            pkg.formats.declare_h0_absmax(fb, 1.0)                                  # (uniform in (-1, 1) by construction, see above)

        def dense_states_measured(fb):                                              # ... and NOT declared: formats.h0_absmax measures it
            fb["initial_node_representation"] = pool[:fb["initial_node_representation"].shape[0]]

        def fresh_epochs(reps, pipelined, hook=None):
            hook = hook or dense_states
            """`reps` passes over the dataset, every batch packed fresh.  pipelined: SparseGGNNChemModel.forward_dataset -- batch
            i+1.. assembled on side streams while batch i's forward runs, forwards alternating over the compute streams
            (utils.StreamPrefetcher); else: packing and forward in sequence on one stream."""
            nn = 0
            with torch.no_grad():
                torch.cuda.synchronize(); t0 = time.perf_counter()
                for rep in range(reps):
                    if pipelined:
                        for fb, states, st in model.forward_dataset(e2e_data, num_streams=max(args.streams, 1), feed_hook=hook,
                                                                   consumer_streams=streams):
                            nn += states.shape[0]
                    else:
                        for fb in dd.pack_batches_device(dms_e2e, params, T, None):
                            hook(fb)
                            model.feed(fb)
                            nn += model.compute_final_node_representations().shape[0]
                    if dbg_steps:
                        torch.cuda.synchronize()
                        print("[bench] fresh epoch %d (%s): %.2f ms since start" % (rep, "pipelined" if pipelined else "one stream",
                                                                                   (time.perf_counter() - t0) * 1e3), file=sys.stderr)
                torch.cuda.synchronize()
            return nn, time.perf_counter() - t0

        gc.collect(); gc.freeze(); gc.disable()
        fresh_epochs(1, True)                                                       # warm (the side stream's allocator pool)
        nn1, e2e1 = fresh_epochs(1, False)
        reps = max(2, int(np.ceil(0.25 / max(e2e1, 1e-3))))
        nn, e2e = fresh_epochs(reps, True)
        # the same with max |h0| MEASURED per fresh batch (ggnn_absmax_f32 + one read-back: a foreign feed that declares nothing)
        nnm1, e2em1 = fresh_epochs(1, False, dense_states_measured)
        nnm, e2em = fresh_epochs(reps, True, dense_states_measured)
        gc.enable()
        del dms_e2e, e2e_data, pool
        out["index_build_ms_per_batch"] = idx_ms
        out["pack_ms_per_batch"] = pack_ms
        out["end_to_end_fresh_batch"] = {
            "value": nn * n_prop / e2e, "unit": "node-state updates/s", "hip_streams": "%d compute + %d packing" % (len(e2e_streams), len(e2e_packs)),
            "epochs_timed": reps, "batches_per_epoch": e2e_batches, "molecules": ms_full.num_graphs, "seconds": e2e,
            "one_stream_value": nn1 * n_prop / e2e1,
            "value_measured_h0": nnm * n_prop / e2em, "one_stream_value_measured_h0": nnm1 * n_prop / e2em1,
            "h0_bound": "value / one_stream_value: max |h0| declared by the packer (formats.declare_h0_absmax); *_measured_h0: measured per "
                        "fresh batch (one 9 us launch + a read-back that waits for the stream)",
            "what": "whole epochs over a full-QM9-sized synthetic dataset; every step assembles a fresh ~100k-node batch on the GPU from graph ids (chem_tensorflow_sparse.py:278-350: h0, "
                    "adjacency lists, in-degree table, graph_nodes_list, plus the message index of :120-129 and the source-pair "
                    "compaction) and runs the 8-step forward on it; batch i+1 is assembled on a side stream under batch i's forward, "
                    "forwards alternate over the compute streams (SparseGGNNChemModel.forward_dataset / utils.StreamPrefetcher; the pipeline "
                    "fills and drains once per epoch); one_stream_value: packing and forward in sequence on one stream"}

    # ---- the other mode, short: forward runs get a `train` object, train runs a `forward` object (all ranks: it holds the collective)
    if not args.no_secondary:
        o_steps = max(4, min(args.steps, 12))
        if headline_train:
            el2, rep2 = timed_region(fwd_step, o_steps, 2, 0.2, no_grad=True)
            n2 = sum(nodes[i % len(feeds)] for i in range(o_steps)) * rep2
            t2 = torch.tensor([n2], dtype=torch.float64, device=dev); dist_ctx.all_reduce_sum_(t2)
            out["forward"] = {"value": float(t2.item()) * n_prop / el2, "unit": "node-state updates/s", "ms_per_step": el2 / (o_steps * rep2) * 1e3,
                              "steps_timed": o_steps * rep2}
        else:
            el2, rep2 = timed_region(train_step, o_steps, 2, 0.3, no_grad=False)
            n2 = sum(nodes[i % len(feeds)] for i in range(o_steps)) * rep2
            t2 = torch.tensor([n2], dtype=torch.float64, device=dev); dist_ctx.all_reduce_sum_(t2)
            out["train"] = {"what": "forward + backward + flat gradient all-reduce (RCCL, N > 1) + per-variable clip + Adam, one batch per rank per step",
                            "value": float(t2.item()) * n_prop / el2, "unit": "node-state updates/s",
                            "ms_per_step": el2 / (o_steps * rep2) * 1e3, "steps_timed": o_steps * rep2, "n_gpus": world}
    # the gradient all-reduce on its own (HIP events around DataParallelContext.reduce_gradients), all ranks take part
    if world > 1 and (headline_train or not args.no_secondary):
        variables = list(model.trainable_variables.values())
        gbuf = [torch.zeros_like(v) for v in variables]
        ev = []
        for i in range(12):
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record(); dist_ctx.reduce_gradients(variables, list(gbuf)); e.record()
            ev.append((s, e))
        torch.cuda.synchronize()
        ar = [s.elapsed_time(e) * 1e3 for s, e in ev][2:]
        rec = {"allreduce_us": float(np.mean(ar)), "allreduce_min_us": float(np.min(ar)),
               "allreduce_bytes": int(sum(v.numel() for v in variables) * 4), "backend": "nccl (RCCL)"}
        out.setdefault("train", {}).update(rec) if not headline_train else out.update({"collective": rec})
        out["allreduce_us"] = rec["allreduce_us"]

    # ---- roofline leg: per-launch HIP-event timing of every kernel (rank 0) -----------------------------
    if rank == 0 and not args.no_roofline and not headline_train:
        reps = max(4, min(args.steps, 12))
        with torch.no_grad(), pkg.ops.kernel_timing():
            fwd_step(0, multi=False)          # untimed: the per-step entry points' first launches (one-off set-up)
        torch.cuda.synchronize()
        with torch.no_grad(), pkg.ops.kernel_timing() as kt:
            for i in range(reps):
                fwd_step(i, multi=False)      # single stream: launches must not overlap while they are timed
        res = kt.results()
        Vb, Mb = float(np.mean([nodes[i % len(feeds)] for i in range(reps)])), float(np.mean([msgs[i % len(feeds)] for i in range(reps)]))
        comps = [getattr(f["message_index"], "_compact", None) for f in feeds]
        Rb = float(np.mean([c.num_rows for c in comps])) if all(c is not None for c in comps) else None
        kernels, tot_ms = kernel_table(res, reps, Vb, Mb, D, T, Rb)
        # HBM traffic per launch cannot be read from inside this process: it comes from the committed rocprofv3 PMC
        # passes of tools/profile_round.sh (separate --pmc FETCH_SIZE / WRITE_SIZE runs; FETCH_SIZE doubled as
        # MI355X_MICROARCH.md prescribes for gfx950), same workload -- and only if that summary was recorded for the
        # kernel sources of this tree (csrc sha1); otherwise traffic stays null and the reason is printed.
        traffic_err = attach_traffic(kernels, "bench", D)
        # The edge-indexed scatter-add (chem_tensorflow_sparse.py:198-209) runs INSIDE the GRU launch on the timed path
        # (ggnn_gru_packed_gather_f32).  Its stand-alone kernel -- the one the training path, edge-bias layers and
        # non-fused hidden sizes use -- is timed here over all 8 timesteps of the same batches so that its HBM rate is
        # reported on its own.
        if pkg.ops.FUSE_GATHER:
            saved, pkg.ops.FUSE_GATHER = pkg.ops.FUSE_GATHER, 0
            try:
                with torch.no_grad(), pkg.ops.kernel_timing() as kt2:
                    for i in range(min(reps, 4)):
                        fwd_step(i, multi=False)
            finally:
                pkg.ops.FUSE_GATHER = saved
            t2 = kt2.results().get("gather_segment_sum")
            if t2:
                k2, _ = kernel_table({"gather_segment_sum": t2}, min(reps, 4), Vb, Mb, D, T, Rb)
                attach_traffic(k2, "bench", D)
                rec = k2["gather_segment_sum"]
                out["scatter_add"] = {"kernel": "gather_segment_sum", "bound": "hbm", "achieved": rec["achieved"], "peak": HBM_PEAK_GBPS,
                                      "unit": "GB/s", "frac": rec["frac"], "frac_of_6.29TBps_copy": rec["achieved"] / HBM_COPY_GBPS,
                                      "avg_us": rec["avg_us"], "algorithmic_bytes": rec["algorithmic_bytes"],
                                      "in_timed_path": False, "traffic": rec["traffic"], "traffic_source": rec.get("traffic_source")}
                # the yardstick at THIS size: a device-to-device copy moving the same number of bytes (half read, half written),
                # timed the same way -- a 25 us launch never reaches the 6.29 TB/s of a GB-sized copy (ramp-up and drain are ~3 us)
                nfl = int(rec["algorithmic_bytes"] / 8)
                src_c, dst_c = torch.empty(nfl, device=dev), torch.empty(nfl, device=dev)
                for _ in range(3):
                    dst_c.copy_(src_c)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(20):
                    dst_c.copy_(src_c)
                e1.record(); torch.cuda.synchronize()
                copy_us = e0.elapsed_time(e1) * 1e3 / 20
                out["scatter_add"]["copy_same_bytes_us"] = copy_us
                out["scatter_add"]["frac_of_copy_same_bytes"] = copy_us / rec["avg_us"]
                del src_c, dst_c
        dom = max(kernels, key=lambda k: kernels[k]["time_share"])
        out["roofline"] = dict(kernels[dom], kernel=dom)
        # What the matrix pipe SUSTAINS on this box: `peak` above is the data-sheet figure (2.5 PF bf16 = every CU issuing an MFMA
        # every 16 clocks at 2.4 GHz); under random operands the chip sits at its socket power limit at a lower clock.  Measured here,
        # ~1 s per pattern, with the library's probe (csrc/ggnn_probe.hip: 8 waves per CU of back-to-back v_mfma_f32_16x16x32_bf16 on
        # register operands -- nothing but MFMAs); `frac` stays the fraction of the data-sheet peak.
        if SPLIT_ACTIVE and kernels[dom].get("pipe", "").startswith(("bf16", "f16")):
            try:
                sus = measure_sustained_mfma(pkg, dev)
                out["roofline"]["sustained_mfma"] = sus
                products = F16X2_PRODUCTS if kernels[dom]["pipe"].startswith("f16") else SPLIT_PRODUCTS     # (the f16 MFMA issues at the bf16 rate)
                out["roofline"]["frac_of_sustained_split_pattern"] = kernels[dom]["mfma_achieved"] / (sus["split_pattern"]["tflops_bf16"] / products)
            except Exception as exc:                                   # (a measurement aid must never take the line down)
                out["roofline"]["sustained_mfma"] = {"error": "%s: %s" % (type(exc).__name__, exc)}
        if traffic_err:
            out["roofline"]["traffic_error"] = traffic_err
        out["config"]["active_source_type_pairs_per_batch"] = None if Rb is None else int(Rb)
        out["kernels"] = kernels
        out["kernel_time_ms_per_step"] = tot_ms / reps

    # ---- secondary configs (rank 0, N = 1) ----------------------------------------------------------------------------
    if rank == 0 and world == 1 and not args.no_secondary and not headline_train:
        sec = {}
        for key, fn in (("config3_dense_b256", secondary_dense), ("config5_large_graph_h256", secondary_large)):
            try:
                sec[key] = fn(pkg, dev)
            except Exception as exc:                                   # a secondary leg must never take the headline down
                sec[key] = {"error": "%s: %s" % (type(exc).__name__, exc)}
            torch.cuda.empty_cache()
        out["secondary"] = sec
        if SPLIT_ACTIVE and GRU_FWD_FORMAT == 2 and not os.environ.get("GGNN_BENCH_CHILD"):
            out["exact_bf16x3_gru_reference"] = exact_format_reference(args, value)
        if not os.environ.get("GGNN_BENCH_CHILD") and abs(args.mean_nodes - 9.0) > 0.5:
            out["secondary"]["config1_mean9_molecules"] = small_molecule_reference(args)

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
        # The port's best thread count is the stated baseline: on a 256-CPU host torch's default (128 threads) oversubscribes the
        # [E_t,100] x [100,100] products and runs 2-3x slower than 8-32 threads.  One forward per candidate, then the reps on the winner.
        default_threads = torch.get_num_threads()
        ncpu = os.cpu_count() or default_threads
        cands = sorted({t for t in (8, 16, 32, 64, default_threads) if 1 <= t <= max(ncpu, 1)})
        sweep = {}
        with torch.no_grad():
            torch.set_num_threads(cands[0])
            OT.sparse_propagate(h0, adj, nin, layers, params)          # warm-up (allocator, first-touch)
            for t in cands:
                torch.set_num_threads(t)
                t0 = time.perf_counter()
                OT.sparse_propagate(h0, adj, nin, layers, params)
                sweep[t] = time.perf_counter() - t0
            best = min(sweep, key=sweep.get)
            torch.set_num_threads(best)
            t0 = time.perf_counter()
            for _ in range(args.cpu_reps):
                ref = OT.sparse_propagate(h0, adj, nin, layers, params)
            cpu_t = min((time.perf_counter() - t0) / args.cpu_reps, sweep[best])
            torch.set_num_threads(default_threads)
            model.feed(f)
            got = model.compute_final_node_representations().cpu()
        out["cpu_baseline"] = {"value": nodes[0] * n_prop / cpu_t, "unit": "node-state updates/s",
                               "cores": best, "kind": "port",
                               "sample": "1 batch (%d nodes, %d messages): one 8-step forward per thread count of the sweep, then %d reps at the "
                                         "best one; torch-CPU fp32 port of chem_tensorflow_sparse.py:117-218 in reference op order" % (
                                             nodes[0], msgs[0], args.cpu_reps),
                               "thread_sweep_node_updates_per_sec": {str(t): nodes[0] * n_prop / v for t, v in sorted(sweep.items())},
                               "torch_default_threads": default_threads,
                               "host_cpus": os.cpu_count(), "graphs_per_sec": graphs[0] / cpu_t,
                               "max_abs_diff_gpu_vs_cpu": float((got - ref).abs().max())}
        if not headline_train:
            out["speedup_vs_cpu_baseline"] = value / out["cpu_baseline"]["value"]

    if rank == 0:
        emit(out)
    if world > 1:
        dist_ctx.barrier()          # rank 0 was still busy with the roofline leg: leave together
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()

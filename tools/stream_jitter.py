#!/usr/bin/env python3
"""Diagnostic: per-step completion times of the bench loop with N streams (is a slow run uniformly slow, or a few
multi-millisecond hiccups in an otherwise normal run?).   python tools/stream_jitter.py [streams] [steps]"""
import importlib
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("ggnn_amd")

n_streams = int(sys.argv[1]) if len(sys.argv) > 1 else 2
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 48
dev = torch.device("cuda:0")
ms = pkg.synthetic_qm9(5700 * 6, mean_nodes=18.0, seed=1000)
model = pkg.SparseGGNNChemModel({"--quiet": True, "--device": str(dev), "train_data": None, "valid_data": ms})
feeds = list(model.make_minibatch_iterator(model.valid_data, is_training=False))[:6]
g = torch.Generator(device="cpu").manual_seed(1234)
for f in feeds:
    f["initial_node_representation"] = (torch.rand(f["initial_node_representation"].shape, generator=g) * 2 - 1).to(dev)
streams = [torch.cuda.Stream(device=dev) for _ in range(n_streams)]
for s in streams:
    s.wait_stream(torch.cuda.current_stream())


def step(i):
    with torch.cuda.stream(streams[i % n_streams]):
        model.feed(feeds[i % len(feeds)])
        out = model.compute_final_node_representations()
        ev = torch.cuda.Event(enable_timing=True)
        ev.record()
    return ev


with torch.no_grad():
    for i in range(8):
        step(i)
    if os.environ.get("GGNN_JITTER_GC", "1") == "0":      # the fix bench.py applies: no cyclic-GC pass in the timed region
        import gc
        gc.collect(); gc.freeze(); gc.disable()
    torch.cuda.synchronize()
    start = torch.cuda.Event(enable_timing=True)
    start.record()
    for s in streams:
        s.wait_event(start)
    t0 = time.perf_counter()
    host = []
    evs = []
    for i in range(steps):
        evs.append(step(i))
        host.append(time.perf_counter() - t0)
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
done = np.array([start.elapsed_time(e) for e in evs])          # ms since start, per step
order = np.sort(done)
gaps = np.diff(np.concatenate([[0.0], order]))
print("streams=%d wall %.2f ms  %.4f ms/step | completion gaps ms: median %.3f  p90 %.3f  max %.3f  (#>2x median: %d) | host enqueue of all steps done at %.2f ms"
      % (n_streams, wall * 1e3, wall * 1e3 / steps, np.median(gaps), np.percentile(gaps, 90), gaps.max(),
         int((gaps > 2 * np.median(gaps)).sum()), host[-1] * 1e3))
print("  gaps:", " ".join("%.2f" % x for x in gaps))
hg = np.diff(np.concatenate([[0.0], np.array(host) * 1e3]))
print("  host enqueue ms per step:", " ".join("%.2f" % x for x in hg))
print("  torch allocator: reserved %.0f MB, num_alloc_retries %d, num_device_alloc %d" % (
    torch.cuda.memory_reserved() / 2**20, torch.cuda.memory_stats().get("num_alloc_retries", -1),
    torch.cuda.memory_stats().get("num_device_alloc", -1)))

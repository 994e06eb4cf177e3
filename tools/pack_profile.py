import importlib, sys, time, os
sys.path.insert(0, "/root/repo")
import numpy as np, torch
pkg = importlib.import_module("gated-graph-neural-network-samples_amd")
ms = pkg.synthetic_qm9(5700 * 6, mean_nodes=18, seed=0)
model = pkg.SparseGGNNChemModel({"--quiet": True, "--device": "cuda:0", "train_data": None, "valid_data": ms})
dd = pkg.data_device
dms = dd.DeviceMoleculeSet(ms, model.device, model.valid_data["label_mask"])
T = model.num_edge_types
for training in (False, True):
    list(dd.pack_batches_device(dms, model.params, T, None, training=training))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out = list(dd.pack_batches_device(dms, model.params, T, None, training=training))
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    n = len(out)
    print("training=%s: host %.2f ms/batch, + drain %.2f ms/batch" % (training, (t1 - t0) / n * 1e3, (t2 - t1) / n * 1e3))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    del out
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    out = list(dd.pack_batches_device(dms, model.params, T, None, training=False))
    torch.cuda.synchronize()
ka = prof.key_averages()
tot_gpu = sum(k.device_time_total for k in ka if k.device_time_total) if hasattr(ka[0], "device_time_total") else sum(k.cuda_time_total for k in ka)
print("GPU kernel time per batch: %.3f ms" % (tot_gpu / 1e3 / len(out)))
rows = sorted(ka, key=lambda k: -(getattr(k, "device_time_total", None) or getattr(k, "cuda_time_total", 0)))[:14]
for k in rows:
    print("  %-70s n=%4d gpu %.1f us total" % (k.key[:70], k.count, (getattr(k, "device_time_total", None) or getattr(k, "cuda_time_total", 0))))

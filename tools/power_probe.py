"""Sustained forward loop (seconds) so that rocm-smi samples taken meanwhile see the chip under THIS load:
    python tools/power_probe.py [seconds]      (GGNN_FWD_DATA=zero: all-zero states and weights, the same instruction stream)
tools/power_probe.sh runs it under a 1-s rocm-smi sampler for both data settings."""
import importlib, os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("gated-graph-neural-network-samples_amd")
secs = float(sys.argv[1]) if len(sys.argv) > 1 else 8.0
ms = pkg.synthetic_qm9(5700 * 2, mean_nodes=18, seed=1000)
model = pkg.SparseGGNNChemModel({"--quiet": True, "--device": "cuda:0", "train_data": None, "valid_data": ms})
feeds = list(model.make_minibatch_iterator(model.valid_data, False))[:2]
zero = os.environ.get("GGNN_FWD_DATA") == "zero"
for f in feeds:
    x = f["initial_node_representation"]
    f["initial_node_representation"] = torch.zeros_like(x) if zero else torch.rand_like(x) * 2 - 1
if zero:
    with torch.no_grad():
        for v in model.trainable_variables.values():
            v.zero_()
with torch.no_grad():
    for i in range(10):
        model.feed(feeds[i % 2]); model.compute_final_node_representations()
    torch.cuda.synchronize()
    print("LOAD-START", flush=True)
    t0 = time.perf_counter(); n = 0
    while time.perf_counter() - t0 < secs:
        for i in range(50):
            model.feed(feeds[i % 2]); model.compute_final_node_representations()
        torch.cuda.synchronize(); n += 50
    dt = time.perf_counter() - t0
print("%s data: %.3f ms per forward over %.1f s (one stream)" % ("zero" if zero else "random", dt / n * 1e3, dt), flush=True)

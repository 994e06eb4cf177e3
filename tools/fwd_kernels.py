"""Per-kernel HIP-event timing of one 8-step forward at the headline shape (bench.py's roofline leg on its own)."""
import importlib, os, sys, json
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("gated-graph-neural-network-samples_amd")
ms = pkg.synthetic_qm9(5700 * 2, mean_nodes=18, seed=1000)
cfg = {"batch_size": int(os.environ["GGNN_FWD_BATCH_NODES"])} if os.environ.get("GGNN_FWD_BATCH_NODES") else {}   # the reference's node cap per batch
model = pkg.SparseGGNNChemModel({"--quiet": True, "--device": "cuda:0", "train_data": None, "valid_data": ms, "--config": cfg})
feeds = list(model.make_minibatch_iterator(model.valid_data, False))[:2]
for f in feeds:
    f["initial_node_representation"] = torch.rand_like(f["initial_node_representation"]) * 2 - 1
# GGNN_FWD_DATA=zero: all-zero states and weights (same instructions, no operand bit toggling) -- the power / clock side of a kernel time
if os.environ.get("GGNN_FWD_DATA") == "zero":
    for f in feeds:
        f["initial_node_representation"] = torch.zeros_like(f["initial_node_representation"])
    with torch.no_grad():
        for v in model.trainable_variables.values():
            v.zero_()
elif os.environ.get("GGNN_FWD_DATA") == "small":      # weights and states scaled to 1e-3: products underflow nothing, few mantissa bits differ
    with torch.no_grad():
        for v in model.trainable_variables.values():
            v.mul_(1e-3)
with torch.no_grad():
    for i in range(6):
        model.feed(feeds[i % 2]); model.compute_final_node_representations()
    with pkg.ops.kernel_timing() as kt:
        for i in range(10):
            model.feed(feeds[i % 2]); model.compute_final_node_representations()
    res = kt.results()
    print("V = %s" % [int(f["initial_node_representation"].shape[0]) for f in feeds], {k: round(float(np.mean(v)) * 1e3, 1) for k, v in res.items()})
    torch.cuda.synchronize()
    import time
    t0 = time.perf_counter()
    for i in range(100):
        model.feed(feeds[i % 2]); model.compute_final_node_representations()
    torch.cuda.synchronize()
    print("one stream: %.3f ms per forward" % ((time.perf_counter() - t0) * 10))

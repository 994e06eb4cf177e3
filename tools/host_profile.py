import importlib, sys, time, os, json
sys.path.insert(0, "/root/repo")
import numpy as np, torch
pkg = importlib.import_module("gated-graph-neural-network-samples_amd")
DEV = torch.device("cuda:0")
ms = pkg.synthetic_qm9(5700 * 3, mean_nodes=18, seed=0)
model = pkg.SparseGGNNChemModel({"--quiet": True, "--device": str(DEV), "train_data": None, "valid_data": ms})
feeds = list(model.make_minibatch_iterator(model.valid_data, False))[:3]
for f in feeds:
    f["edge_weight_dropout_keep_prob"] = model.params["edge_weight_dropout_keep_prob"]; f["out_layer_dropout_keep_prob"] = 1.0
for i in range(6): model.train_batch(feeds[i % 3])
torch.cuda.synchronize()
N = 30
t0 = time.perf_counter()
for i in range(N): model.train_batch(feeds[i % 3])
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print("host enqueue %.2f ms/step, total %.2f ms/step (drain after last enqueue %.2f ms)" % ((t1 - t0) / N * 1e3, (t2 - t0) / N * 1e3, (t2 - t1) * 1e3))
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
for i in range(10): model.train_batch(feeds[i % 3])
pr.disable(); torch.cuda.synchronize()
st = pstats.Stats(pr); st.sort_stats("cumulative").print_stats(35)

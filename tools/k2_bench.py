"""Stand-alone segment sum (K2) at the config-2 shape for the GGNN_K2_ITEMS settings (items per thread of the slot-head kernel) (env read once per process: run per value)."""
import importlib, os, sys, torch, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("gated-graph-neural-network-samples_amd")
ms = pkg.synthetic_qm9(5700, mean_nodes=18, seed=0)
model = pkg.SparseGGNNChemModel({"--quiet": True, "--device": "cuda:0", "train_data": None, "valid_data": ms})
f = list(model.make_minibatch_iterator(model.valid_data, False))[0]
idx = f["message_index"]; comp = idx._compact; nin = f["num_incoming_edges_per_type"]
V = f["initial_node_representation"].shape[0]
Hc = torch.rand(comp.num_rows, 100, device="cuda:0")
run = lambda: pkg.ops.gather_segment_sum_compact(Hc, idx, comp, nin, None, True)
ref = run().clone()
for _ in range(5): run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(200): out = run()
e1.record(); torch.cuda.synchronize()
print("GGNN_K2_ITEMS=%s: %.2f us  (V=%d, M=%d)  checksum %.6f" % (os.environ.get("GGNN_K2_ITEMS", "default"), e0.elapsed_time(e1) * 5, V, idx.num_messages, float(out.double().sum())))

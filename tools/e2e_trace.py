"""One short pipelined fresh-batch run for rocprofv3 --kernel-trace (tools/e2e_probe.py's loop, one setting)."""
import importlib, os, sys, time, gc
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("gated-graph-neural-network-samples_amd")
dd = pkg.data_device
dev = torch.device("cuda:0")
ms = pkg.synthetic_qm9(int(os.environ.get("MOLS", "133885")), mean_nodes=18, seed=1000)
model = pkg.SparseGGNNChemModel({"--quiet": True, "--device": "cuda:0", "train_data": None, "valid_data": ms})
T, params = model.num_edge_types, model.params
model.prepare_resident_data(model.valid_data, False)
dms = model.valid_data["molecules_dev"]
pool = torch.rand((100000, params["hidden_size"]), device=dev) * 2 - 1
cs = [torch.cuda.Stream(), torch.cuda.Stream()]
kw = {"pack_streams": int(os.environ.get("PS", "2")), "depth": int(os.environ.get("DEPTH", "3")), "priority": int(os.environ.get("PRIO", "-1"))}
gc.collect(); gc.freeze(); gc.disable()
with torch.no_grad():
    for rep in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter(); nn = 0
        for fb, st in pkg.utils.StreamPrefetcher(dd.pack_batches_device(dms, params, T, None), dev, consumer_streams=cs, **kw):
            V = fb["initial_node_representation"].shape[0]
            with torch.cuda.stream(st):
                fb["initial_node_representation"] = pool[:V]
                model.feed(fb); model.compute_final_node_representations()
            nn += V
        torch.cuda.synchronize()
        print("rep", rep, nn * 8 / (time.perf_counter() - t0) / 1e6, flush=True)

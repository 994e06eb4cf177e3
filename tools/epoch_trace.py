"""One short training epoch (fresh batches) for rocprofv3 --kernel-trace: where does a step's wall time go?"""
import importlib, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("gated-graph-neural-network-samples_amd")
NB = int(os.environ.get("NB", "12"))
ms = pkg.synthetic_qm9(5700 * NB, mean_nodes=18, seed=0)
np.random.seed(0); torch.manual_seed(0)
model = pkg.SparseGGNNChemModel({"--quiet": True, "--device": "cuda:0", "train_data": ms, "valid_data": ms,
                                 "--config": {"threaded_batches": os.environ.get("THREADED", "0") != "0"}})
model.run_epoch("warm", model.train_data, True)
torch.cuda.synchronize(); t0 = time.perf_counter()
steps = model.run_epoch("train", model.train_data, True)[4]
torch.cuda.synchronize()
print("ms per step", (time.perf_counter() - t0) / steps * 1e3, "steps", steps, flush=True)

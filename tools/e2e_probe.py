"""End-to-end fresh-batch forward (pack on side streams + forward over compute streams): rate for several prefetcher settings, and
the host's share (time spent inside the pack call and the forward call without waiting for the GPU)."""
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


def run(reps, **kw):
    nn = 0; th_pack = 0.0; th_fwd = 0.0
    with torch.no_grad():
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(reps):
            it = iter(pkg.utils.StreamPrefetcher(dd.pack_batches_device(dms, params, T, None), dev, consumer_streams=cs, **kw))
            while True:
                ta = time.perf_counter()
                try:
                    fb, st = next(it)
                except StopIteration:
                    break
                tb = time.perf_counter()
                V = fb["initial_node_representation"].shape[0]
                with torch.cuda.stream(st):
                    fb["initial_node_representation"] = pool[:V]
                    model.feed(fb); model.compute_final_node_representations()
                tc = time.perf_counter()
                th_pack += tb - ta; th_fwd += tc - tb; nn += V
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    return nn * 8 / dt / 1e6, th_pack / dt, th_fwd / dt


gc.collect(); gc.freeze(); gc.disable()
for kw in ({"pack_streams": 1, "depth": 1, "priority": 0}, {"pack_streams": 1, "depth": 1, "priority": -1},
           {"pack_streams": 1, "depth": 3, "priority": -1}, {"pack_streams": 2, "depth": 2, "priority": -1},
           {"pack_streams": 2, "depth": 3, "priority": -1}, {"pack_streams": 3, "depth": 4, "priority": -1},
           {"pack_streams": 3, "depth": 4, "priority": 0}):
    run(1, **kw)
    r = run(8, **kw)
    print(kw, "-> %.1f M node-updates/s, host in pack %.0f %%, in forward %.0f %%" % (r[0], 100 * r[1], 100 * r[2]), flush=True)


def run_fd(reps):
    """the same epochs through SparseGGNNChemModel.forward_dataset"""
    nn = 0
    hook = lambda fb: fb.__setitem__("initial_node_representation", pool[:fb["initial_node_representation"].shape[0]])
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps):
        for fb, states, st in model.forward_dataset(model.valid_data, num_streams=2, feed_hook=hook):
            nn += states.shape[0]
    torch.cuda.synchronize()
    return nn * 8 / (time.perf_counter() - t0) / 1e6


run_fd(1)
print("forward_dataset: %.1f M node-updates/s" % run_fd(8), flush=True)
r = run(8, pack_streams=2, depth=3, priority=-1)
print("manual loop again: %.1f" % r[0], flush=True)

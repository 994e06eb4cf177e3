"""Where a training epoch with fresh batches spends its host time: waiting for the next batch, queueing the step, draining.
   python tools/epoch_profile.py [threaded=1]"""
import importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
pkg = importlib.import_module("gated-graph-neural-network-samples_amd")
threaded = (sys.argv[1:] or ["1"])[0] != "0"
ms = pkg.synthetic_qm9(5700 * int(os.environ.get("NB", "8")), mean_nodes=18, seed=0)
model = pkg.SparseGGNNChemModel({"--quiet": True, "--device": "cuda:0", "train_data": ms, "valid_data": ms, "--config": {"threaded_batches": threaded}})
model.run_epoch("warm", model.train_data, True)
torch.cuda.synchronize()
st0 = torch.cuda.memory_stats()
for ep in range(2):
    it = model.make_minibatch_iterator(model.train_data, True)
    if threaded:
        it = pkg.utils.ThreadedIterator(it, max_queue_size=2, device=model.device)
    it = iter(it)
    waits, queues = [], []
    t_ep = time.perf_counter()
    while True:
        t0 = time.perf_counter()
        try:
            b = next(it)
        except StopIteration:
            break
        t1 = time.perf_counter()
        b['out_layer_dropout_keep_prob'] = 1.0
        model.train_batch(b)
        t2 = time.perf_counter()
        waits.append(t1 - t0); queues.append(t2 - t1)
    torch.cuda.synchronize()
    t3 = time.perf_counter()
    n = len(waits)
    print("epoch %d (%s): %d steps, %.2f ms/step = next(batch) %.2f + train_batch %.2f (+ final drain %.2f ms total)" % (
        ep, "threaded" if threaded else "inline", n, (t3 - t_ep) / n * 1e3, np.mean(waits) * 1e3, np.mean(queues) * 1e3, (t3 - t2) * 1e3))
    print("   per step next(batch) ms:", " ".join("%.1f" % (w * 1e3) for w in waits), "| train_batch ms:", " ".join("%.1f" % (q * 1e3) for q in queues))
st1 = torch.cuda.memory_stats()
for k in ("num_device_alloc", "num_device_free", "num_alloc_retries", "num_sync_all_streams"):
    print(k, st1.get(k, 0) - st0.get(k, 0))
print("reserved MB", st1["reserved_bytes.all.current"] / 2**20, "allocated MB", st1["allocated_bytes.all.current"] / 2**20)

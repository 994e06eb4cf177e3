"""The gather-fused GRU launch at the headline shape under each ring form (ggnn_gru_form_set), on one feed of the benchmark's model:
per-launch HIP-event times of the forward's kernels and the forward's wall time per form.   python tools/wide_bench.py [forms...]"""
import importlib, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("gated-graph-neural-network-samples_amd")
lib = pkg._lib.load()
forms = [int(x) for x in sys.argv[1:]] or [-1, 0, 62, 63]
ms = pkg.synthetic_qm9(5700 * 2, mean_nodes=18, seed=1000)
cfg = {"batch_size": int(os.environ["GGNN_FWD_BATCH_NODES"])} if os.environ.get("GGNN_FWD_BATCH_NODES") else {}
model = pkg.SparseGGNNChemModel({"--quiet": True, "--device": "cuda:0", "train_data": None, "valid_data": ms, "--config": cfg})
feeds = list(model.make_minibatch_iterator(model.valid_data, False))[:2]
for f in feeds:
    f["initial_node_representation"] = torch.rand_like(f["initial_node_representation"]) * 2 - 1
ref = None
with torch.no_grad():
    for form in forms:
        lib.ggnn_gru_form_set(form)
        for i in range(6):
            model.feed(feeds[i % 2]); out = model.compute_final_node_representations()
        torch.cuda.synchronize()
        if ref is None:
            model.feed(feeds[0]); ref = model.compute_final_node_representations().clone()
        else:
            model.feed(feeds[0]); o = model.compute_final_node_representations()
            print("form %d: bit-identical to form %d: %s (max |diff| %.3g)" % (form, forms[0], bool(torch.equal(o, ref)), float((o - ref).abs().max())))
        with pkg.ops.kernel_timing() as kt:
            for i in range(10):
                model.feed(feeds[i % 2]); model.compute_final_node_representations()
        res = kt.results()
        print("form %d V=%s" % (form, [int(f["initial_node_representation"].shape[0]) for f in feeds]),
              {k: round(float(np.mean(v)) * 1e3, 1) for k, v in res.items()})
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(100):
            model.feed(feeds[i % 2]); model.compute_final_node_representations()
        torch.cuda.synchronize()
        print("form %d one stream: %.3f ms per forward" % (form, (time.perf_counter() - t0) * 10), flush=True)
lib.ggnn_gru_form_set(-1)

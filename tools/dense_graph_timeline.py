"""s_memtime timeline of workgroup 0 (waves 0 and 6) of the graph-resident dense kernel (GGNN_DG_TPTR).  python tools/dense_graph_timeline.py"""
import importlib, os, sys, torch, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("gated-graph-neural-network-samples_amd")
dev = "cuda:0"
b, v, E, D, steps = 256, 29, 4, 100, 4
rng = np.random.default_rng(0)
t = lambda a: torch.from_numpy(np.ascontiguousarray(a.astype(np.float32))).to(dev)
A = t(rng.random((b, E, v, v)) < 2.0 / v); h0 = t(rng.uniform(-1, 1, (b, v, D)))
W = t(rng.uniform(-.1, .1, (E, D, D))); Wg = t(rng.uniform(-.1, .1, (2 * D, 2 * D))); Wc = t(rng.uniform(-.1, .1, (2 * D, D)))
eb = t(rng.normal(0, .1, (E, D))); bg = t(np.ones(2 * D)); bc = t(np.zeros(D))
P = pkg.ops.PackedWeights()
run = lambda: pkg.ops.dense_propagate(h0, A, P.dense_edge(W), P.dense_gru(Wg, Wc, D), eb, bg, bc, steps)
for _ in range(3): run()
torch.cuda.synchronize()
tb = torch.zeros(8 * 2 * 8, dtype=torch.int64, device=dev)
os.environ["GGNN_DG_TPTR"] = str(tb.data_ptr())
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); run(); e1.record(); torch.cuda.synchronize()
print("launch %.1f us" % (e0.elapsed_time(e1) * 1e3))
x = tb.cpu().numpy().reshape(8, 2, 8)
names = ["step start", "transform stages done", "barrier 1 passed", "aggregation done", "acts loaded", "5 gate stages + r,u done", "r*h exchanged", "candidate stage done"]
t0 = x[0, 0, 0]
for s in range(steps):
    for w in range(2):
        r = x[s, w] - t0
        print("step %d wave %d @%6d: " % (s, 0 if w == 0 else 6, r[0]) + " | ".join("%s +%d" % (names[k], r[k] - r[k - 1]) for k in range(1, 8)))

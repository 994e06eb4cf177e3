"""s_memtime timeline of workgroup 0 (waves 0 and 4) of the fused GRU backward (debug stamps behind GGNN_BWD_TPTR).
   python tools/gru_bwd_timeline.py"""
import importlib, os, sys, torch, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("gated-graph-neural-network-samples_amd")
V, D, nx, T = 99990, 100, 1, 4
dev = "cuda:0"
r = lambda *s: torch.rand(*s, device=dev) * 2 - 1
g, h, c = r(V, D), r(V, D), r(V, D)
rr, u = torch.rand(V, D, device=dev), torch.rand(V, D, device=dev)
Wg, Wc = r((nx + 1) * D, 2 * D) * 0.2, r((nx + 1) * D, D) * 0.2
nin = torch.ones(V, T, device=dev)
packed = pkg.ops.PackedWeights().gru_bwd(Wg, Wc, nx, D)
run = lambda: pkg.ops.gru_bwd_fused(g, h, rr, u, c, packed, nin, True, nx, "tanh")
for _ in range(3): run()
torch.cuda.synchronize()
tbuf = torch.zeros(8 * 2 * 16, dtype=torch.int64, device=dev)
os.environ["GGNN_BWD_TPTR"] = str(tbuf.data_ptr())
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); run(); e1.record(); torch.cuda.synchronize()
print("launch by events: %.1f us" % (e0.elapsed_time(e1) * 1e3))
t = tbuf.cpu().numpy().astype(np.float64).reshape(8, 2, 16)
t0 = t[0, 0, 0]
names = ["pass start", "head done", "stage0 done", "epi0 done(pre-st1)", "stage1 done", "stage2 done", "dh stored", "pass end", "stage3 done", "stage4 done", "stage5 done"]
order = [0, 1, 2, 3, 4, 5, 6, 8, 9, 10, 7]
for p in range(5):
    for w in range(2):
        if t[p, w, 0] == 0: continue
        row = t[p, w] - t0
        prev = row[0]
        out = []
        for k in order:
            out.append("%s +%d" % (names[k], row[k] - prev)); prev = row[k]
        print("pass %d wave %d @%d: " % (p, w * 4, row[0]) + " | ".join(out))

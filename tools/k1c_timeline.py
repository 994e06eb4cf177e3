"""Per-wave timeline of the compacted message transform (debug stamps behind GGNN_K1C_TPTR)."""
import importlib, os, sys, torch, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("gated-graph-neural-network-samples_amd")
dev = "cuda:0"
ms = pkg.synthetic_qm9(5700, mean_nodes=18, seed=1000)
model = pkg.SparseGGNNChemModel({"--quiet": True, "--device": dev, "train_data": None, "valid_data": ms})
feed = next(iter(model.make_minibatch_iterator(model.valid_data, False)))
index = feed["message_index"]
comp = pkg.ops.build_compact_sources(index)
V, D, T = index.num_nodes, 100, index.num_edge_types
h = torch.rand(V, D, device=dev) * 2 - 1
W = (torch.rand(T, D, D, device=dev) - 0.5) * 0.3
for _ in range(3): out = pkg.ops.msg_transform_compact(h, W, comp)
torch.cuda.synchronize()
NB = 1024
NW = int(os.environ.get('NW', 8))           # waves per workgroup of the kernel build
tbuf = torch.zeros(NB * NW * 8, dtype=torch.int64, device=dev)
os.environ["GGNN_K1C_TPTR"] = str(tbuf.data_ptr())
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); out = pkg.ops.msg_transform_compact(h, W, comp); e1.record()
torch.cuda.synchronize()
print("rows %d, launch (pack pre-pass + transform) %.1f us" % (comp.num_rows, e0.elapsed_time(e1) * 1e3))
t = tbuf.cpu().numpy().astype(np.float64).reshape(NB, NW, 8)
used = t[:, 0, 0] > 0
t = t[used]
print("workgroups", len(t))
start = (t[:, :, 0] - t[:, :, 0].min()) / 100.0           # us, real-time counter
end = (t[:, :, 7] - t[:, :, 0].min()) / 100.0
print("start skew over workgroups: max %.1f us | end: median %.1f, max %.1f us | wave duration median %.1f max %.1f us"
      % (start.max(), np.median(end), end.max(), np.median(end - start), (end - start).max()))
pre = t[:, :, 1] - t[:, :, 6]
print("start -> first rows requested (clocks): median %d p90 %d max %d" % (np.median(pre), np.percentile(pre, 90), pre.max()))
print("  by wave id (median):", [int(np.median(pre[:, w])) for w in range(NW)])
print("  by workgroup index octile (median):", [int(np.median(pre[i * len(pre) // 8:(i + 1) * len(pre) // 8])) for i in range(8)])
print("  by workgroup % 8 (XCD, median):", [int(np.median(pre[i::8])) for i in range(8)])
pro = t[:, :, 2] - t[:, :, 1]
print("barrier wait (clocks): median %d max %d" % (np.median(pro), pro.max()))
tot = t[:, :, 5].copy()
for k in (4, 3, 2):
    tot = np.where(tot > 0, tot, t[:, :, k])
print("start -> last tile done (clocks): median %d p90 %d max %d" % (np.median(tot - t[:, :, 6]), np.percentile(tot - t[:, :, 6], 90), (tot - t[:, :, 6]).max()))
for k in range(3, 6):
    ok = t[:, :, k] > 0
    if ok.any():
        prev = t[:, :, k - 1]
        d = (t[:, :, k] - prev)[ok]
        print("tile %d per wave (clocks): n %d median %d p90 %d max %d" % (k - 2, ok.sum(), np.median(d), np.percentile(d, 90), d.max()))
# when does the first / last wave pass the barrier, relative to launch start (need a clock ratio: use wave duration)
clk = np.median((t[:, 0, 6][t[:, 0, 6] > 0] - t[:, 0, 1][t[:, 0, 6] > 0])) if (t[:, 0, 6] > 0).any() else 0
print("workgroup start times (us) by decile:", np.round(np.percentile(start[:, 0], [0, 10, 50, 90, 100]), 1))
print("workgroup end times   (us) by decile:", np.round(np.percentile(end.max(axis=1), [0, 10, 50, 90, 100]), 1))

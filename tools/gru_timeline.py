"""Per-stage s_memtime timeline of workgroup 0 of the fused GRU (debug instrumentation, GGNN_GRU_TPTR)."""
import importlib, os, sys, torch, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("gated-graph-neural-network-samples_amd")
V, D, nx = 99990, 100, 1
dev = "cuda:0"
tbuf = torch.zeros(4 * 6 * 8 * 4, dtype=torch.int64, device=dev)
os.environ["GGNN_GRU_TPTR"] = str(tbuf.data_ptr())
xs = [torch.rand(V, D, device=dev) * 2 - 1]
h = torch.rand(V, D, device=dev) * 2 - 1
Wg = (torch.rand(2 * D, 2 * D, device=dev) - 0.5) * 0.3; bg = torch.ones(2 * D, device=dev)
Wc = (torch.rand(2 * D, D, device=dev) - 0.5) * 0.3; bc = torch.zeros(D, device=dev)
out = torch.empty_like(h); ws = pkg.ops.gru_workspace(V, D, dev)
for _ in range(3): pkg.ops.gru(xs, h, Wg, bg, Wc, bc, "tanh", out=out, ws=ws)
torch.cuda.synchronize()
t = tbuf.cpu().numpy().reshape(4, 6, 8, 4).astype(np.float64)
t0 = t[0, 0, :, 0].min()
print("pass stage | wave0: start dma_done mma_done barrier_done (cycles since kernel start; deltas)")
for p in range(4):
    for s in range(6):
        w = t[p, s, 0] - t0
        allw_mma = t[p, s, :, 2] - t0
        print(p, s, "| %8d  dma+%5d  mma+%6d  bar+%5d | mma_done spread over waves: min %d max %d" % (
            w[0], w[1] - w[0], w[2] - w[1], w[3] - w[2], allw_mma.min() - w[0], allw_mma.max() - w[0]))
    if p < 3:
        print("   epilogue/pass gap: %d" % (t[p + 1, 0, 0, 0] - t[p, 5, 0, 3]))

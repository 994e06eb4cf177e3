"""Times ops.gru alone (V=100k, D=100, nx=1) -- used with GGNN_GRU_DBG ablation bits."""
import importlib, os, sys, torch, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("gated-graph-neural-network-samples_amd")
V, D = int(os.environ.get("V", 99990)), 100
nx = int(os.environ.get("NX", 1))
dev = "cuda:0"
xs = [torch.rand(V, D, device=dev) * 2 - 1 for _ in range(nx)]
h = torch.rand(V, D, device=dev) * 2 - 1
Wg = (torch.rand((nx + 1) * D, 2 * D, device=dev) - 0.5) * 0.3; bg = torch.ones(2 * D, device=dev)
Wc = (torch.rand((nx + 1) * D, D, device=dev) - 0.5) * 0.3; bc = torch.zeros(D, device=dev)
out = torch.empty_like(h); ws = pkg.ops.gru_workspace(V, D, dev)
for _ in range(5): pkg.ops.gru(xs, h, Wg, bg, Wc, bc, "tanh", out=out, ws=ws, fmt=int(os.environ.get("GRU_FMT", "2")))
torch.cuda.synchronize()
s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
s.record()
for _ in range(20): pkg.ops.gru(xs, h, Wg, bg, Wc, bc, "tanh", out=out, ws=ws, fmt=int(os.environ.get("GRU_FMT", "2")))
e.record(); torch.cuda.synchronize()
print("dbg=%s V=%d nx=%d: %.1f us" % (os.environ.get("GGNN_GRU_DBG", "0"), V, nx, s.elapsed_time(e) / 20 * 1e3))

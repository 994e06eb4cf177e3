"""Per-stage s_memtime timeline of workgroup 0 of the fused GRU + per-workgroup start/end stamps (debug
instrumentation behind GGNN_GRU_TPTR; not part of the product path: the stamps are compiled in only with -DGGNN_GRU_STAMPS=1).
    bash tools/variant_lib.sh tl ggnn_gru_fused.hip,ggnn_gru_fused_split.hip -DGGNN_GRU_STAMPS=1
    GGNN_LIB_VARIANT=tl python tools/gru_timeline.py [nx]"""
import importlib, os, sys, torch, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("gated-graph-neural-network-samples_amd")
V, D = 99990, 100
nx = int(sys.argv[1]) if len(sys.argv) > 1 else 1
NSTAGE = 3 * (nx + 1)
dev = "cuda:0"
tbuf = torch.zeros(4096 + 256 * 4, dtype=torch.int64, device=dev)
xs = [torch.rand(V, D, device=dev) * 2 - 1 for _ in range(nx)]
h = torch.rand(V, D, device=dev) * 2 - 1
Wg = (torch.rand((nx + 1) * D, 2 * D, device=dev) - 0.5) * 0.3; bg = torch.ones(2 * D, device=dev)
Wc = (torch.rand((nx + 1) * D, D, device=dev) - 0.5) * 0.3; bc = torch.zeros(D, device=dev)
out = torch.empty_like(h); ws = pkg.ops.gru_workspace(V, D, dev)
for _ in range(3): pkg.ops.gru(xs, h, Wg, bg, Wc, bc, "tanh", out=out, ws=ws, fmt=int(os.environ.get("GRU_FMT", "2")))
torch.cuda.synchronize()
os.environ["GGNN_GRU_TPTR"] = str(tbuf.data_ptr())
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); pkg.ops.gru(xs, h, Wg, bg, Wc, bc, "tanh", out=out, ws=ws, fmt=int(os.environ.get("GRU_FMT", "2"))); e1.record()
torch.cuda.synchronize()
print("launch (pack pre-pass + GRU) by events: %.1f us" % (e0.elapsed_time(e1) * 1e3))
raw = tbuf.cpu().numpy().astype(np.float64)
t = raw[:4 * NSTAGE * 8 * 4].reshape(4, NSTAGE, 8, 4)
t0 = t[0, 0, :, 0].min()
print("pass stage | wave0: start  late-side-work-done  mma(+early side work)-done  barrier-done (shader clocks; deltas)")
for p in range(4):
    for s in range(NSTAGE):
        w = t[p, s, 0] - t0
        allw_mma = t[p, s, :, 2] - t0
        w4 = t[p, s, 4] - t0
        print(p, s, "| %8d  pre+%5d  mma+%6d  bar+%5d | mma_done over waves: min %d max %d | wave4: +%5d +%5d bar+%5d" % (
            w[0], w[1] - w[0], w[2] - w[1], w[3] - w[2], allw_mma.min() - w[0], allw_mma.max() - w[0],
            w4[1] - w4[0], w4[2] - w4[1], w4[3] - w4[2]))
    if p < 3:
        print("   epilogue/pass gap: %d" % (t[p + 1, 0, 0, 0] - t[p, NSTAGE - 1, 0, 3]))
b = raw[4096:].reshape(256, 4)
clk = (b[:, 2] - b[:, 0]) / ((b[:, 3] - b[:, 1]) / 100.0)       # shader clocks per us (real-time counter: 100 MHz)
start_us = (b[:, 1] - b[:, 1].min()) / 100.0
end_us = (b[:, 3] - b[:, 1].min()) / 100.0
dur = end_us - start_us
print("workgroups: shader clock %.0f MHz | start skew max %.1f us | duration us min %.1f median %.1f max %.1f | last end %.1f us"
      % (np.median(clk), start_us.max(), dur.min(), np.median(dur), dur.max(), end_us.max()))
order = np.argsort(dur)
print("  slowest blocks:", [(int(i), round(float(dur[i]), 1)) for i in order[-6:]], " fastest:", [(int(i), round(float(dur[i]), 1)) for i in order[:4]])
print("  duration by block%8 (XCD):", [round(float(np.median(dur[i::8])), 1) for i in range(8)])

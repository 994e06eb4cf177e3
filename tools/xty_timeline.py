"""s_memtime timeline of workgroup (0,0,0) of ggnn_xty_f32 (debug stamps behind GGNN_XTY_TPTR), and the launch time as a
function of the row count.   python tools/xty_timeline.py"""
import importlib, os, sys, torch, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("gated-graph-neural-network-samples_amd")
dev = "cuda:0"

def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3

for nseg, N in ((2, 200), (2, 100)):
    for V in (99986,):
        xs = [torch.rand(V, 100, device=dev) for _ in range(nseg)]
        dy = torch.rand(V, N, device=dev)
        t = timeit(lambda: pkg.ops.xty(xs, dy, ones_row=True))
        print("K=%d N=%d V=%d: %.1f us (kernel + reduce), %.1f TF" % (nseg * 100, N, V, t, 2.0 * V * nseg * 100 * N / (t * 1e-6) / 1e12), flush=True)
    V = 99986
    xs = [torch.rand(V, 100, device=dev) for _ in range(nseg)]
    dy = torch.rand(V, N, device=dev)
    tbuf = torch.zeros(128, dtype=torch.int64, device=dev)
    os.environ["GGNN_XTY_TPTR"] = str(tbuf.data_ptr())
    pkg.ops.xty(xs, dy, ones_row=True); torch.cuda.synchronize()
    del os.environ["GGNN_XTY_TPTR"]
    t = tbuf.cpu().numpy().astype(np.int64).reshape(2, 64)
    for w in range(2):
        r = t[w]; t0 = t[0, 0]
        print(" wave %s: start %d, first slab landed %d, loop done %d, stored %d" % ("0" if w == 0 else "15", r[0] - t0, r[1] - t0, r[62] - t0, r[63] - t0))
        prev = r[1]
        for i in range(20):
            if not r[2 + i]: break
            print("   slab %2d: DMA of the next issued +%d, MFMAs issued +%d, barrier passed +%d" % (i, r[22 + i] - prev, r[42 + i] - r[22 + i], r[2 + i] - r[42 + i]))
            prev = r[2 + i]

"""Per-sub-stage s_memtime timeline of workgroup 0 of the gather-fused GRU in its half-stage forms (GGNN_GRU_FORM = 1 / 2):
one nx = 1 launch through the model (a one-layer, one-timestep sparse GGNN on a full-size batch).
    bash tools/variant_lib.sh tl ggnn_gru_fused.hip,ggnn_gru_fused_split.hip -DGGNN_GRU_STAMPS=1
    GGNN_LIB_VARIANT=tl GGNN_GRU_FORM=2 python tools/gru_half_timeline.py"""
import importlib, os, sys, torch, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("gated-graph-neural-network-samples_amd")
dev = "cuda:0"
NW = 4 if os.environ.get("GGNN_GRU_FORM", "2") == "1" else 8
NSTAGE = 6
ms = pkg.synthetic_qm9(5700, mean_nodes=18, seed=1000)
model = pkg.SparseGGNNChemModel({"--quiet": True, "--device": dev, "train_data": None, "valid_data": ms,
                                 "--config": {"layer_timesteps": [1], "residual_connections": {}}})
feed = next(iter(model.make_minibatch_iterator(model.valid_data, False)))
feed["initial_node_representation"] = torch.rand_like(feed["initial_node_representation"]) * 2 - 1
tbuf = torch.zeros(4096 + 1024 * 4, dtype=torch.int64, device=dev)
with torch.no_grad():
    for _ in range(3):
        model.feed(feed); model.compute_final_node_representations()
    torch.cuda.synchronize()
    os.environ["GGNN_GRU_TPTR"] = str(tbuf.data_ptr())
    model.feed(feed); model.compute_final_node_representations()
    torch.cuda.synchronize()
raw = tbuf.cpu().numpy().astype(np.float64)
A = raw[:4 * NSTAGE * NW * 4].reshape(4, NSTAGE, NW, 4)
B = raw[2048:2048 + 4 * NSTAGE * NW * 4].reshape(4, NSTAGE, NW, 4)
t0 = A[0, 0, :, 0].min()
print("form %s, %d waves.  per (pass, stage), waves 0 (early) and %d (late); shader clocks" % (os.environ.get("GGNN_GRU_FORM", "2"), NW, NW // 2))
print("   start | side A | burst A | wait+barrier A || side B | burst B | side B' | wait+barrier B")
for p in range(4):
    for s in range(NSTAGE):
        for w in (0, NW // 2):
            a, b = A[p, s, w] - t0, B[p, s, w] - t0
            if A[p, s, w, 0] == 0:
                continue
            print("p%d s%d w%d %8d | %5d | %5d | %5d || %5d | %5d | %5d | %5d" % (
                p, s, w, a[0], a[1] - a[0], a[2] - a[1], a[3] - a[2], b[0] - a[3], b[1] - b[0], (b[2] - b[1]) if b[2] else 0,
                b[3] - (b[2] if b[2] else b[1])))
    if p < 3 and A[p + 1, 0, 0, 0]:
        print("   epilogue + pass gap (wave 0): %d" % (A[p + 1, 0, 0, 0] - B[p, NSTAGE - 1, 0, 3]))
nwg = 512 if NW == 4 else 256
bb = raw[4096:4096 + nwg * 4].reshape(nwg, 4)
ok = bb[:, 3] > 0
clk = (bb[ok, 2] - bb[ok, 0]) / ((bb[ok, 3] - bb[ok, 1]) / 100.0)
dur = (bb[ok, 3] - bb[ok, 1]) / 100.0
print("workgroups: shader clock %.0f MHz | duration us min %.1f median %.1f max %.1f | span %.1f us" % (
    np.median(clk), dur.min(), np.median(dur), dur.max(), (bb[ok, 3].max() - bb[ok, 1].min()) / 100.0))

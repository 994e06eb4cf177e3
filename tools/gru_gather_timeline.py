"""All-waves s_memtime timeline of workgroup 0 of the gather-fused GRU, form 0 / 3 (whole-image ring), one nx = 1 launch through the model.
    bash tools/variant_lib.sh tl ggnn_gru_fused.hip,ggnn_gru_fused_split.hip -DGGNN_GRU_STAMPS=1
    GGNN_LIB_VARIANT=tl [GGNN_GRU_FORM_R0=3] python tools/gru_gather_timeline.py
Per (pass, stage) and wave: side work before the burst (late waves) | burst + side work after it (early waves) | wait at the barrier."""
import importlib, os, sys, torch, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("gated-graph-neural-network-samples_amd")
dev = "cuda:0"
NW, NSTAGE = 8, 6
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
t0 = A[0, 0, :, 0].min()
for p in range(4):
    print("pass %d   (per wave 0..7: pre | burst+post | barrier wait), stage start of wave 0, stage length" % p)
    for s in range(NSTAGE):
        if A[p, s, 0, 0] == 0:
            continue
        a = A[p, s] - t0
        nxt = (A[p, s + 1, 0, 0] if s + 1 < NSTAGE else (A[p + 1, 0, 0, 0] if p < 3 else 0)) - t0
        cells = " ".join("%4d|%5d|%4d" % (a[w, 1] - a[w, 0], a[w, 2] - a[w, 1], a[w, 3] - a[w, 2]) for w in range(NW))
        print("s%d %7d len %5d  %s" % (s, a[0, 0], (nxt - a[0, 0]) if nxt > 0 else -1, cells))
bb = raw[4096:4096 + 256 * 4].reshape(256, 4)
ok = bb[:, 3] > 0
clk = (bb[ok, 2] - bb[ok, 0]) / ((bb[ok, 3] - bb[ok, 1]) / 100.0)
dur = (bb[ok, 3] - bb[ok, 1]) / 100.0
print("workgroups: shader clock %.0f MHz | duration us min %.1f median %.1f max %.1f | span %.1f us" % (
    np.median(clk), dur.min(), np.median(dur), dur.max(), (bb[ok, 3].max() - bb[ok, 1].min()) / 100.0))

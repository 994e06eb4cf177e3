"""Per-stage s_memtime timeline of workgroup 0 of the wide gather-fused GRU launch (csrc/ggnn_gru_wide.hip, nx = 1) inside the
benchmark's forward.  Stamps are compiled in only with -DGGNN_WIDE_STAMPS=1:
    bash tools/variant_lib.sh wst ggnn_gru_wide.hip -DGGNN_WIDE_STAMPS=1
    GGNN_LIB_VARIANT=wst python tools/wide_timeline.py [form=63]"""
import importlib, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("gated-graph-neural-network-samples_amd")
lib = pkg._lib.load()
form = int(sys.argv[1]) if len(sys.argv) > 1 else 63
ms = pkg.synthetic_qm9(5700, mean_nodes=18, seed=1000)
model = pkg.SparseGGNNChemModel({"--quiet": True, "--device": "cuda:0", "train_data": None, "valid_data": ms, "--config": {}})
feed = list(model.make_minibatch_iterator(model.valid_data, False))[0]
feed["initial_node_representation"] = torch.rand_like(feed["initial_node_representation"]) * 2 - 1
tbuf = torch.zeros(4 * 32 * 4, dtype=torch.int64, device="cuda:0")
lib.ggnn_gru_form_set(form)
with torch.no_grad():
    for _ in range(3):
        model.feed(feed); model.compute_final_node_representations()
    torch.cuda.synchronize()
    os.environ["GGNN_GRU_TPTR"] = str(tbuf.data_ptr())
    model.feed(feed); model.compute_final_node_representations()
    torch.cuda.synchronize()
t = tbuf.cpu().numpy().astype(np.float64).reshape(4, 32, 4)
names = ["x->r", "h->r", "x->u (r epi)", "h->u", "x->c (u epi)", "rh->c"]
for p in range(4):
    if t[p, 0, 0] == 0:
        break
    t0 = t[p, 0].min()
    print("pass %d (clocks since the pass's first stamp; per wave: side work | products | barrier wait)" % p)
    for j in range(6):
        row = []
        for w in range(4):
            a, b, c, d = t[p, 4 * j:4 * j + 4, w]
            row.append("%5d|%5d|%5d" % (b - a, c - b, d - c))
        print("  %-13s start %7d   %s" % (names[j], t[p, 4 * j, 0] - t0, "   ".join(row)))
    print("  candidate epilogue + blend: %s   gather finish + split: %s   pass total: %s" % (
        [int(t[p, 24, w] - t[p, 23, w]) for w in range(4)], [int(t[p, 25, w] - t[p, 24, w]) for w in range(4)],
        [int(t[p, 25, w] - t[p, 0, w]) for w in range(4)]))
lib.ggnn_gru_form_set(-1)

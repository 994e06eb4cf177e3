"""Where the wide form differs from a ring form (debug):  python tools/wide_debug.py fmt nx V M"""
import importlib, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from conftest import random_graph_batch
pkg = importlib.import_module("gated-graph-neural-network-samples_amd")
lib = pkg._lib.load()
fmt, nx, V, M = (int(x) for x in (sys.argv[1:5] + ["3", "1", "500", "1200"][len(sys.argv) - 1:]))
D, T, cuda = 100, 4, "cuda:0"
rng = np.random.default_rng(1)
h, adj, nin = random_graph_batch(rng, V, M, T, D, sorted_src=False)
dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(cuda)
Wg = rng.uniform(-0.2, 0.2, ((nx + 1) * D, 2 * D)).astype(np.float32); Wc = rng.uniform(-0.2, 0.2, ((nx + 1) * D, D)).astype(np.float32)
bg = rng.uniform(-0.5, 1.0, 2 * D).astype(np.float32); bc = rng.uniform(-0.5, 0.5, D).astype(np.float32)
if os.environ.get('DBG_BC0'): bc[:] = 0
if os.environ.get('DBG_BG1'): bg[:] = 1
if os.environ.get('DBG_NOAVG'): nin = None
res = [dev(rng.uniform(-1, 1, (V, D)).astype(np.float32)) for _ in range(nx - 1)]
H = dev(rng.uniform(-1, 1, (V * T, D)).astype(np.float32))
index = pkg.ops.build_message_index([dev(a) for a in adj], V)
nd = dev(nin) if nin is not None else None
hd, Wgd, Wcd, bgd, bcd = (dev(x) for x in (h, Wg, Wc, bg, bc))
packed = pkg.ops.PackedWeights().gru(Wgd, Wcd, nx, D, fmt)
outs = {}
for form in (0, 6):
    lib.ggnn_gru_form_set(form)
    s = {}
    pkg.ops.gru_packed_gather(res, hd, packed, bgd, bcd, H, index, None, nd, fmt=fmt, save=s)
    outs[form] = (pkg.ops.gru_packed_gather(res, hd, packed, bgd, bcd, H, index, None, nd, fmt=fmt).cpu().numpy(), {k: v.cpu().numpy() for k, v in s.items()})
lib.ggnn_gru_form_set(-1)
ref, refs = outs[0]
for form in (6,):
    o, sv = outs[form]
    d = np.abs(o - ref)
    print("form %d vs 0: h' differing %d of %d, max %.3g; rows %s cols %s" % (form, (d > 0).sum(), d.size, d.max(), np.unique(np.nonzero(d)[0])[:12], np.unique(np.nonzero(d)[1])[:30]))
    for k in ("incoming", "r", "u", "c"):
        dk = np.abs(sv[k] - refs[k])
        print("    %s: differing %d, max %.3g, cols %s" % (k, (dk > 0).sum(), dk.max(), np.unique(np.nonzero(dk)[1])[:30]))
deg = np.diff(index.row_ptr.cpu().numpy())
o = outs[6][0]; d = np.abs(o - ref).max(1)
bad = np.nonzero(d > 0)[0]
print("bad rows %d of %d; degrees of bad rows: %s; degrees overall: %s" % (len(bad), V, np.bincount(deg[bad], minlength=8)[:12], np.bincount(deg, minlength=8)[:12]))
print("bad rows mod 16:", np.bincount(bad % 16, minlength=16), " first bad rows:", bad[:20], " max diff %.3g" % d.max())

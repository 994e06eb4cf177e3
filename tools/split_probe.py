"""Error of the fused GRU launch against an f64 evaluation in the process's matrix mode (GGNN_MATRIX=f32 | default: split) and, under
the split path, in the operand format SPLIT_PROBE_GRU_FMT asks for (2: two f16 pieces x three products; 3, the default: the exact
bf16 x 3 split -- a per-launch argument since ABI 3).  Run once per mode and compare:
    GGNN_MATRIX=f32 python tools/split_probe.py; SPLIT_PROBE_GRU_FMT=2 python tools/split_probe.py; python tools/split_probe.py
"""
import importlib, os, sys, json
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("gated-graph-neural-network-samples_amd")
ops = pkg.ops
lib = pkg._lib.load()
mode = "bf16x3" if lib.ggnn_matrix_path_is_split() else "f32"
dev = "cuda:0"
gru_fmt = int(os.environ.get("SPLIT_PROBE_GRU_FMT", "3")) if lib.ggnn_matrix_path_is_split() else 0
out = {"mode": mode, "gru_format": gru_fmt}     # 2: f16 x 2 pieces, 3 products; 3: bf16 x 3, 6 products; 0: f32 MFMA
for D, nx, V in ((100, 1, 40000), (100, 3, 20000), (64, 2, 20000), (32, 1, 20000), (256, 1, 8000), (128, 2, 8000)):     # (128 / 256: the column-panel GRU)
    g = torch.Generator(device="cpu").manual_seed(5 + D + nx)
    xs = [(torch.rand(V, D, generator=g) * 2 - 1) for _ in range(nx)]
    h = torch.rand(V, D, generator=g) * 2 - 1
    s = 1.0 / np.sqrt((nx + 1) * D)
    Wg = (torch.rand((nx + 1) * D, 2 * D, generator=g) * 2 - 1) * (3 * s)
    Wc = (torch.rand((nx + 1) * D, D, generator=g) * 2 - 1) * (3 * s)
    bg = torch.rand(2 * D, generator=g) - 0.5
    bc = torch.rand(D, generator=g) - 0.5
    # f64 evaluation
    X = torch.cat(xs + [h], 1).double()
    ru = torch.sigmoid(X @ Wg.double() + bg.double())
    r, u = ru[:, :D], ru[:, D:]
    c = torch.tanh(torch.cat([x.double() for x in xs] + [r * h.double()], 1) @ Wc.double() + bc.double())
    want = u * h.double() + (1 - u) * c
    # pre-activation magnitudes: the error of the products is relative to sum |a b|
    dx = [t.to(dev) for t in xs]; dh = h.to(dev)
    save = {}
    got = ops.gru(dx, dh, Wg.to(dev), bg.to(dev), Wc.to(dev), bc.to(dev), "tanh", save=save, fmt=gru_fmt or None)
    e = (got.double().cpu() - want).abs()
    er = (save["r"].double().cpu() - r).abs()
    ec = (save["c"].double().cpu() - c).abs()
    key = "D%d_nx%d" % (D, nx)
    out[key] = {"h_max": float(e.max()), "h_rms": float((e ** 2).mean().sqrt()), "r_max": float(er.max()),
                "c_max": float(ec.max()), "c_rms": float((ec ** 2).mean().sqrt())}
# the compacted message transform (K1): rows h[pair_node] x W_type against an f64 product, error relative to sum |a||w|
for D, V in ((100, 30000), (64, 9000)):
    T = 4
    g = torch.Generator(device="cpu").manual_seed(77 + D)
    h = torch.rand(V, D, generator=g) * 2 - 1
    W = (torch.rand(T, D, D, generator=g) * 2 - 1) * 0.3
    src = torch.randint(0, V, (4 * V,), generator=g)
    tgt = torch.randint(0, V, (4 * V,), generator=g)
    adj = [torch.stack([src[t::T], tgt[t::T]], 1).to(torch.int32).to(dev) for t in range(T)]
    index = ops.build_message_index(adj, V)
    comp = ops.build_compact_sources(index)
    Hc = ops.msg_transform_compact(h.to(dev), W.to(dev), comp).double().cpu()[:comp.num_rows]
    pn = comp.pair_node.cpu().long()[:comp.num_rows]
    want = torch.empty(comp.num_rows, D, dtype=torch.float64)
    bound = torch.empty(comp.num_rows, D, dtype=torch.float64)
    for t in range(T):
        lo, hi = int(comp.type_row_off[t]), int(comp.type_row_off[t + 1])
        rows = h[pn[lo:hi]].double()
        want[lo:hi] = rows @ W[t].double()
        bound[lo:hi] = rows.abs() @ W[t].double().abs()
    e = (Hc - want).abs()
    out["transform_D%d" % D] = {"max": float(e.max()), "rms": float((e ** 2).mean().sqrt()), "max_rel_to_sum_abs": float((e / bound).max())}
print(json.dumps(out))

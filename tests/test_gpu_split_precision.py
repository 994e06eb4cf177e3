"""The split matrix path (csrc/ggnn_split.hpp: every f32 product as six bf16 MFMA products of operands split exactly into three
bf16 pieces -- or, in the fused GRU forward since round 4, as three f16 MFMA products of operands rounded to two f16 pieces -- with
f32 accumulation) is an f32-faithful evaluation, not a reduced-precision one: against an f64 evaluation of the same
GRU update / message transform its error is no larger than that of the f32-MFMA kernels (GGNN_MATRIX=f32), which round once per k.

The matrix path is fixed per process (packed weight images are in its format), so each mode runs tools/split_probe.py in a
process of its own."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _probe(mode):
    env = dict(os.environ)
    env.pop("GGNN_MATRIX", None); env.pop("GGNN_GRU_FMT", None)
    if mode == "f32":
        env["GGNN_MATRIX"] = "f32"
    if mode == "bf16x3":
        env["GGNN_GRU_FMT"] = "3"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "split_probe.py")], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    out = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert out["mode"] == ("f32" if mode == "f32" else "bf16x3")
    assert out.pop("gru_format") == {"f32": 0, "split": 2, "bf16x3": 3}[mode]
    return out


def test_split_products_are_as_accurate_as_f32_mfma(cuda):
    """Three processes: the f32 MFMA kernels, the default split path (fused GRU forward: two f16 pieces x three products; everything
    else three bf16 pieces x six products) and the all-bf16x3 path (GGNN_GRU_FMT=3).  Both split paths must be at least as accurate
    against f64 as the f32 MFMA -- the f16 x 2 form rounds each operand to 22 bits, but rounds its sums once per 32 terms instead of
    once per term."""
    f32, split, b3 = _probe("f32"), _probe("split"), _probe("bf16x3")
    for name, got in (("f16x2 GRU (default)", split), ("bf16x3", b3)):
        for key, rec in got.items():
            if key == "mode":
                continue
            ref = f32[key]
            for field, val in rec.items():
                # mean-square errors: the split forms round once per 32-term dot product instead of once per term -- not worse than f32
                if field.endswith("rms"):
                    assert val <= 1.1 * ref[field], (name, key, field, val, ref[field])
                else:   # maxima are single samples of the same distribution: same size class
                    assert val <= 1.5 * ref[field] + 1e-9, (name, key, field, val, ref[field])
        # absolute: the bound the f32 parity tests use for a K-term f32 product chain, 4e-7 * sum_k |a_k||w_k| (tests/test_gpu_parity.py)
        for key in ("transform_D100", "transform_D64"):
            assert got[key]["max_rel_to_sum_abs"] < 4e-7, (name, key, got[key])
        for key in ("D100_nx1", "D100_nx3", "D64_nx2", "D32_nx1"):
            assert got[key]["h_max"] < 3e-6 and got[key]["c_max"] < 4e-6, (name, key, got[key])
    for key in ("transform_D100", "transform_D64"):
        assert f32[key]["max_rel_to_sum_abs"] < 4e-7, (key, f32[key])
    # the two-piece form is not the less accurate of the two split forms either (measured: 0.85-0.95 x)
    for key in ("D100_nx1", "D100_nx3", "D64_nx2", "D32_nx1"):
        assert split[key]["h_rms"] <= 1.1 * b3[key]["h_rms"] and split[key]["c_rms"] <= 1.1 * b3[key]["c_rms"], (key, split[key], b3[key])


def test_f16x2_gru_operand_range(pkg, cuda):
    """The two-piece f16 form's operand range (csrc/ggnn_split.hpp): activations beyond +-65504 are clamped before the split -- a
    finite, saturated operand (never Inf - Inf = NaN); the gates are saturated long before, so h' still equals the f64 evaluation.
    Small activations keep their absolute accuracy (the lo piece of a value below 2^-11 is an f16 subnormal, which the MFMA keeps)."""
    import numpy as np
    import torch
    lib = pkg._lib.load()
    if lib.ggnn_gru_forward_format() != 2:
        pytest.skip("the fused GRU forward is not in the f16 x 2 format in this process")
    ops = pkg.ops
    V, D = 3000, 100
    g = torch.Generator(device="cpu").manual_seed(21)
    s = 1.0 / np.sqrt(2 * D)
    Wg = (torch.rand(2 * D, 2 * D, generator=g) * 2 - 1) * (3 * s)
    Wc = (torch.rand(2 * D, D, generator=g) * 2 - 1) * (3 * s)
    bg = torch.rand(2 * D, generator=g) - 0.5
    bc = torch.rand(D, generator=g) - 0.5
    h = torch.rand(V, D, generator=g) * 2 - 1

    def run(x):
        X = torch.cat([x, h], 1).double()
        ru = torch.sigmoid(X @ Wg.double() + bg.double())
        r, u = ru[:, :D], ru[:, D:]
        c = torch.tanh(torch.cat([x.double(), r * h.double()], 1) @ Wc.double() + bc.double())
        want = u * h.double() + (1 - u) * c
        got = ops.gru([x.to(cuda)], h.to(cuda), Wg.to(cuda), bg.to(cuda), Wc.to(cuda), bc.to(cuda), "tanh")
        return got.double().cpu(), want

    # one huge entry per row (1e5 .. 1e7: beyond f16's 65504), the rest ordinary: every pre-activation it touches is saturated
    x = torch.rand(V, D, generator=g) * 2 - 1
    cols = torch.randint(0, D, (V,), generator=g)
    big = (10.0 ** (5 + 2 * torch.rand(V, generator=g))) * torch.where(torch.rand(V, generator=g) < 0.5, -1.0, 1.0)
    x[torch.arange(V), cols] = big.float()
    got, want = run(x)
    assert torch.isfinite(got).all()
    # (a clamped entry changes its pre-activations by (|x| - 65504) |w|: only where |w| of that row / column is so small that the
    #  pre-activation is NOT saturated can the result differ -- |w| < 20 / 65504 = 3e-4, a 1e-3 fraction of the weights)
    bad = (got - want).abs() > 1e-5
    assert bad.float().mean() < 2e-2, float(bad.float().mean())
    # small activations: absolute accuracy as for ordinary ones
    x = (torch.rand(V, D, generator=g) * 2 - 1) * 10.0 ** (-6 * torch.rand(V, D, generator=g))
    got, want = run(x)
    assert (got - want).abs().max() < 3e-6


def test_power_of_two_scaling_commutes_bitwise(pkg, cuda):
    """A power-of-two rescaling of the operands moves exponents only: h 2^k and W 2^-k must give bit-identical products in either
    matrix path -- on the split path this pins the exponent handling of the bf16 pieces (truncation split, exact residuals, bf16's
    f32 exponent range) on the hardware, far from the magnitudes the model's states live at."""
    import numpy as np
    import torch
    ops = pkg.ops
    V, D, T = 4000, 100, 4
    g = torch.Generator(device="cpu").manual_seed(3)
    h = torch.rand(V, D, generator=g) * 2 - 1
    W = (torch.rand(T, D, D, generator=g) * 2 - 1) * 0.3
    src = torch.randint(0, V, (3 * V,), generator=g); tgt = torch.randint(0, V, (3 * V,), generator=g)
    adj = [torch.stack([src[t::T], tgt[t::T]], 1).to(torch.int32).to(cuda) for t in range(T)]
    comp = ops.build_compact_sources(ops.build_message_index(adj, V))
    base = ops.msg_transform_compact(h.to(cuda), W.to(cuda), comp)[:comp.num_rows].clone()
    for k in (40, -40, 90):
        got = ops.msg_transform_compact((h * 2.0 ** k).to(cuda), (W * 2.0 ** -k).to(cuda), comp)[:comp.num_rows]
        assert torch.equal(got, base), k
    # and linearity in the rows: a row scaled by 2^k scales its product by 2^k
    got = ops.msg_transform_compact((h * 2.0 ** 20).to(cuda), W.to(cuda), comp)[:comp.num_rows]
    assert torch.equal(got, base * 2.0 ** 20)
    assert np.isfinite(base.cpu().numpy()).all()

"""The split matrix path (csrc/ggnn_split.hpp: every f32 product as six bf16 MFMA products of operands split exactly into three
bf16 pieces, f32 accumulation) is an f32-faithful evaluation, not a reduced-precision one: against an f64 evaluation of the same
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
    env.pop("GGNN_MATRIX", None)
    if mode == "f32":
        env["GGNN_MATRIX"] = "f32"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "split_probe.py")], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    out = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert out["mode"] == ("f32" if mode == "f32" else "bf16x3")
    return out


def test_split_products_are_as_accurate_as_f32_mfma(cuda):
    f32, split = _probe("f32"), _probe("split")
    for key, rec in split.items():
        if key == "mode":
            continue
        ref = f32[key]
        for field, val in rec.items():
            # mean-square errors: the split form rounds once per 32-term dot product instead of once per term -- not worse than f32
            if field.endswith("rms"):
                assert val <= 1.1 * ref[field], (key, field, val, ref[field])
            else:   # maxima are single samples of the same distribution: same size class
                assert val <= 1.5 * ref[field] + 1e-9, (key, field, val, ref[field])
    # absolute: the bound the f32 parity tests use for a K-term f32 product chain, 4e-7 * sum_k |a_k||w_k| (tests/test_gpu_parity.py)
    for key in ("transform_D100", "transform_D64"):
        assert split[key]["max_rel_to_sum_abs"] < 4e-7, (key, split[key])
        assert f32[key]["max_rel_to_sum_abs"] < 4e-7, (key, f32[key])
    for key in ("D100_nx1", "D100_nx3", "D64_nx2", "D32_nx1"):
        assert split[key]["h_max"] < 3e-6 and split[key]["c_max"] < 4e-6, (key, split[key])


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

"""The split matrix path (csrc/ggnn_split.hpp: every f32 product as six bf16 MFMA products of operands split exactly into three
bf16 pieces -- or, in the fused GRU forward since round 4, as three f16 MFMA products of operands rounded to two f16 pieces -- with
f32 accumulation) is an f32-faithful evaluation, not a reduced-precision one: against an f64 evaluation of the same
GRU update / message transform its error is no larger than that of the f32-MFMA kernels (GGNN_MATRIX=f32), which round once per k.

The matrix path is fixed per process (packed weight images are in its format), so each mode runs tools/split_probe.py in a
process of its own; the GRU forward's operand format is a per-launch argument since ABI 3 (SPLIT_PROBE_GRU_FMT tells the probe)."""
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
    env["SPLIT_PROBE_GRU_FMT"] = "2" if mode == "split" else "3"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "split_probe.py")], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    out = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert out["mode"] == ("f32" if mode == "f32" else "bf16x3")
    assert out.pop("gru_format") == {"f32": 0, "split": 2, "bf16x3": 3}[mode]
    return out


def test_split_products_are_as_accurate_as_f32_mfma(cuda):
    """Three processes: the f32 MFMA kernels, the split path with the fused GRU forward in the two-piece f16 format (two f16 pieces x
    three products, operands inside its range; everything else three bf16 pieces x six products) and the all-bf16x3 path.  Both
    split paths must be at least as accurate
    against f64 as the f32 MFMA -- the f16 x 2 form rounds each operand to 22 bits, but rounds its sums once per 32 terms instead of
    once per term."""
    f32, split, b3 = _probe("f32"), _probe("split"), _probe("bf16x3")
    for name, got in (("f16x2 GRU", split), ("bf16x3", b3)):
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
    """What the two-piece f16 format does OUTSIDE its operand range, and that nothing reaches it there unasked.  Called with
    fmt = F16X2 explicitly (the caller vouches for the range, csrc/ggnn_split.hpp): an activation beyond +-65504 overflows its f16 hi
    piece (Inf) and the residual piece (-Inf / NaN), so the rows it sits in come out NON-FINITE -- loud, not a plausible wrong number
    (round 4 clamped the operand to +-65504 and returned finite, slightly wrong states).  The SAME call in the exact format (what
    ops.gru runs on raw weights, and what formats.py selects whenever it cannot prove the range) matches f64 on every entry.  Small
    activations keep their absolute accuracy in both formats (the lo piece of a value below 2^-11 is an f16 subnormal, which the
    MFMA keeps)."""
    import numpy as np
    import torch
    f = pkg.formats
    if not f.split_path():
        pytest.skip("f32 matrix path: no operand formats")
    ops = pkg.ops
    V, D = 3000, 100
    g = torch.Generator(device="cpu").manual_seed(21)
    s = 1.0 / np.sqrt(2 * D)
    Wg = (torch.rand(2 * D, 2 * D, generator=g) * 2 - 1) * (3 * s)
    Wc = (torch.rand(2 * D, D, generator=g) * 2 - 1) * (3 * s)
    bg = torch.rand(2 * D, generator=g) - 0.5
    bc = torch.rand(D, generator=g) - 0.5
    h = torch.rand(V, D, generator=g) * 2 - 1

    def run(x, fmt):
        X = torch.cat([x, h], 1).double()
        ru = torch.sigmoid(X @ Wg.double() + bg.double())
        r, u = ru[:, :D], ru[:, D:]
        c = torch.tanh(torch.cat([x.double(), r * h.double()], 1) @ Wc.double() + bc.double())
        want = u * h.double() + (1 - u) * c
        got = ops.gru([x.to(cuda)], h.to(cuda), Wg.to(cuda), bg.to(cuda), Wc.to(cuda), bc.to(cuda), "tanh", fmt=fmt)
        return got.double().cpu(), want

    # one huge entry per row (1e5 .. 1e7: beyond f16's 65504), the rest ordinary
    x = torch.rand(V, D, generator=g) * 2 - 1
    cols = torch.randint(0, D, (V,), generator=g)
    big = (10.0 ** (5 + 2 * torch.rand(V, generator=g))) * torch.where(torch.rand(V, generator=g) < 0.5, -1.0, 1.0)
    x[torch.arange(V), cols] = big.float()
    got, want = run(x, f.BF16X3)                       # the exact format: f32 on every input
    assert torch.isfinite(got).all() and (got - want).abs().max() < 1e-5
    got2, _ = run(x, f.F16X2)                          # the two-piece format, forced outside its range: every row holds a huge entry
    assert not torch.isfinite(got2).all()              # ... and fails LOUDLY: non-finite states, nothing clamped
    # the host policy never selects it for such a batch
    with f.forced("auto"):
        assert f.layer_format(f.state_bound(float(x.abs().max()), "tanh"), 1.0, float(Wg.abs().max())) == f.BF16X3
    # small activations: absolute accuracy as for ordinary ones, in both formats
    x = (torch.rand(V, D, generator=g) * 2 - 1) * 10.0 ** (-6 * torch.rand(V, D, generator=g))
    for fmt in (f.F16X2, f.BF16X3):
        got, want = run(x, fmt)
        assert (got - want).abs().max() < 3e-6


def test_absmax_kernel(pkg, cuda):
    """ggnn_absmax_f32, the device half of the operand-range proof: max |x| per tensor in one launch, NaN / Inf propagated, any
    alignment and size (scalar head and tail around the 16-byte middle)."""
    import math
    import torch
    f = pkg.formats
    g = torch.Generator(device="cpu").manual_seed(2)
    base = torch.randn(1_000_003, generator=g)
    ts = [base[:0], base[:1], base[1:8], base[3:1_000_003], base.clone() * 1e-30, -base.abs() * 7.0, torch.zeros(513)]
    got = f.absmax([t.to(cuda) if t.numel() == 0 else t.to(cuda)[:] for t in ts])
    want = [0.0 if t.numel() == 0 else float(t.abs().max()) for t in ts]
    assert got == want
    # views at odd offsets of one device buffer (4-byte aligned, not 16)
    d = base.to(cuda)
    assert f.absmax([d[1:77], d[2:1003], d[3:]]) == [float(base[1:77].abs().max()), float(base[2:1003].abs().max()), float(base[3:].abs().max())]
    big = base.clone(); big[777_777] = float("inf")
    nan = base.clone(); nan[5] = float("nan"); nan[6] = float("inf")
    out = f.absmax([big.to(cuda), nan.to(cuda)])
    assert out[0] == math.inf and math.isnan(out[1])
    many = [torch.full((10 + i,), float(i), device=cuda) for i in range(70)]            # more tensors than one launch takes
    assert f.absmax(many) == [float(i) for i in range(70)]


def test_parity_suite_on_the_f32_mfma_path(cuda):
    """The third arithmetic path the library ships (GGNN_MATRIX=f32: the f32 MFMA kernels, fixed per process): the model-level
    parity tests -- oracle configs, reference-run fixtures, gradients -- in a process of their own (missing #2)."""
    env = dict(os.environ, GGNN_MATRIX="f32")
    env.pop("GGNN_GRU_FMT", None)
    sel = [("tests/test_gpu_parity.py", "test_sparse_model_matches_oracle and exact"),
           ("tests/test_gpu_reference_golden.py", "test_forward_matches_reference_run and exact"),
           ("tests/test_gpu_train.py", "test_gradients_match_oracle_autograd and exact and True")]
    for path, k in sel:
        r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, path), "-x", "-q", "-m", "gpu", "-k", k, "-p", "no:cacheprovider"],
                           env=env, capture_output=True, text=True, timeout=1500, cwd=ROOT)
        assert r.returncode == 0, (path, k, r.stdout[-3000:], r.stderr[-2000:])
        assert " passed" in r.stdout and "failed" not in r.stdout, r.stdout[-500:]


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

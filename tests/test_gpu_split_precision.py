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

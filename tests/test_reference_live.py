"""Checkpoint interchange with the reference's own code, LIVE (CPU; only where the reference checkout exists -- the build
container -- and skipped elsewhere; nothing here runs under `-m gpu`):

  * a checkpoint written by the package's save_progress is restored BY THE REFERENCE (chem_tensorflow.py:330-359, run
    over the TF-1.3 op shim in a subprocess): its params assertion passes and every global variable -- weights, Adam
    slots, beta powers -- takes the package's value;
  * a best-model pickle written by the reference's own train() is restored by the package: same weights, Adam state
    and step counters.
"""
import json
import os
import pickle
import subprocess
import sys

import numpy as np
import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
DRIVER = os.path.join(HERE, "golden", "reference_live.py")
pytestmark = pytest.mark.skipif(not os.path.isdir("/root/reference"), reason="needs the reference checkout (build container only)")

CONFIGS = {
    "sparse": {"layer_timesteps": [2, 1], "residual_connections": {"1": [0]}, "use_edge_bias": True, "batch_size": 150,
               "num_epochs": 1, "edge_weight_dropout_keep_prob": 1.0, "random_seed": 9},
    "dense": {"batch_size": 4, "num_epochs": 1, "random_seed": 9},
}


def _data_dir(pkg, tmp_path, kind):
    d = tmp_path / "data"
    d.mkdir()
    train, valid = pkg.synthetic_qm9(30, mean_nodes=8, seed=31), pkg.synthetic_qm9(16, mean_nodes=8, seed=32)
    (d / "molecules_train.json").write_text(json.dumps(train.to_json()))
    (d / "molecules_valid.json").write_text(json.dumps(valid.to_json()))
    (d / "config.json").write_text(json.dumps(CONFIGS[kind]))
    return str(d), train.to_json(), valid.to_json()


def _run(*args):
    subprocess.run([sys.executable, DRIVER, *args], check=True, stdout=subprocess.DEVNULL, timeout=600)


def _model(pkg, kind, train, valid, **extra):
    cls = pkg.SparseGGNNChemModel if kind == "sparse" else pkg.DenseGGNNChemModel
    args = {"--quiet": True, "--device": "cpu", "--config": json.dumps(CONFIGS[kind]), "train_data": train, "valid_data": valid}
    args.update(extra)
    return cls(args)


@pytest.mark.parametrize("kind", ["sparse", "dense"])
def test_reference_restores_package_checkpoint(pkg, tmp_path, kind):
    data_dir, train, valid = _data_dir(pkg, tmp_path, kind)
    m = _model(pkg, kind, train, valid)
    m.optimizer.apply_gradients([torch.randn_like(v) * 0.1 for v in m.trainable_variables.values()])   # non-trivial Adam state
    m.optimizer.apply_gradients([torch.randn_like(v) * 0.1 for v in m.trainable_variables.values()])
    ckpt, out = str(tmp_path / "pkg.pickle"), str(tmp_path / "ref_vars.npz")
    m.save_progress(ckpt, 11, 5)
    _run("restore", kind, data_dir, ckpt, out)
    z = np.load(out)
    assert (int(z["train_step"]), int(z["valid_step"])) == (11, 5)
    saved = pickle.load(open(ckpt, "rb"))["weights"]
    names = [str(n) for n in z["names"]]
    assert set(names) == set(saved) - {"ggnn_amd/adam_step:0"}          # the reference finds a value for EVERY variable
    for i, n in enumerate(names):
        np.testing.assert_array_equal(z["v%d" % i], np.asarray(saved[n], dtype=np.float32).reshape(z["v%d" % i].shape), err_msg=n)
    assert float(saved["beta1_power:0"]) == pytest.approx(0.9 ** 3) and float(saved["beta2_power:0"]) == pytest.approx(0.999 ** 3)


@pytest.mark.parametrize("kind", ["sparse", "dense"])
def test_package_restores_reference_checkpoint(pkg, tmp_path, kind):
    data_dir, train, valid = _data_dir(pkg, tmp_path, kind)
    out_dir = tmp_path / "ref_run"
    out_dir.mkdir()
    _run("train", kind, data_dir, str(out_dir))
    paths = json.loads((out_dir / "paths.json").read_text())
    blob = pickle.load(open(paths["best"], "rb"))
    m = _model(pkg, kind, train, valid, **{"--restore": paths["best"]})
    assert (m.train_step_id, m.valid_step_id) == (blob["train_step"], blob["valid_step"])
    nv = m.named_variables()
    assert set(nv) | set(m.optimizer.state_variables(m.trainable_variables)) - {"ggnn_amd/adam_step:0"} == set(blob["weights"])
    for n, t in nv.items():
        np.testing.assert_array_equal(t.numpy().reshape(np.shape(blob["weights"][n])), blob["weights"][n], err_msg=n)
    state = m.optimizer.state_variables(m.trainable_variables)
    steps = blob["train_step"]
    assert m.optimizer.t == steps                                        # recovered from beta1_power
    for n, a in state.items():
        if n != "ggnn_amd/adam_step:0":
            np.testing.assert_allclose(a, blob["weights"][n], rtol=1e-6, err_msg=n)

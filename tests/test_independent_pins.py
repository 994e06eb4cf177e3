"""Cross-checks of the restated TensorFlow-1.3 arithmetic against implementations that are NOT this repository's restatements
(round-2 review, item 8): the oracle and the TF-op shim (oracle/tf13_shim) restate GRUCell / CudnnCompatibleGRUCell / clip_by_norm /
ApplyAdam from TF's published source and had only each other as witnesses.

  * tf.contrib.cudnn_rnn.CudnnCompatibleGRUCell (chem_tensorflow_sparse.py:105-108) IS the cuDNN GRU, which torch.nn.GRUCell
    implements independently: same function after re-ordering the gate blocks (TF: [r|u] kernels over [x;h] rows, torch: r,z,n
    blocks with separate input / hidden matrices).
  * tf.nn.rnn_cell.GRUCell (chem_tensorflow_sparse.py:104, chem_tensorflow_dense.py:88) multiplies r*h BEFORE the candidate
    kernel; for a DIAGONAL hidden candidate kernel and zero hidden bias (r*h) diag(d) == r * (h diag(d)), so on that subspace it
    must coincide with the cuDNN form, i.e. with torch.nn.GRUCell: gate order (r first), bias placement, the blend u*h + (1-u)*c.
  * per-variable tf.clip_by_norm + tf.train.AdamOptimizer (chem_tensorflow.py:183-191) on a linear loss have a closed form for
    the first two steps: w_t = w_{t-1} - lr * g / (|g| + eps / sqrt(1 - beta2^t)), g = a * clip / max(||a||, clip).
"""
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SHIM = os.path.join(ROOT, "oracle", "tf13_shim")


def _torch_gru_from_tf(Wg, bg, Wcx, bcx, Wch, bch, I, H):
    cell = torch.nn.GRUCell(I, H).double()
    t = lambda a: torch.from_numpy(np.asarray(a, np.float64))
    with torch.no_grad():
        cell.weight_ih.copy_(torch.cat([t(Wg[:I, :H]).T, t(Wg[:I, H:]).T, t(Wcx).T]))
        cell.weight_hh.copy_(torch.cat([t(Wg[I:, :H]).T, t(Wg[I:, H:]).T, t(Wch).T]))
        cell.bias_ih.copy_(torch.cat([t(bg), t(bcx)]))
        cell.bias_hh.copy_(torch.cat([torch.zeros(2 * H, dtype=torch.float64), t(bch)]))
    return cell


def _shim_cell(kind, x, h, weights):
    """Evaluate one of the shim's cells (what the reference's graph construction instantiates) on numpy inputs."""
    if SHIM not in sys.path:
        sys.path.insert(0, SHIM)
    import tensorflow as tf
    assert tf.__version__.endswith("shim")
    with tf.Graph().as_default():
        xp = tf.placeholder(tf.float32, [None, x.shape[1]])
        hp = tf.placeholder(tf.float32, [None, h.shape[1]])
        cell = (tf.contrib.cudnn_rnn.CudnnCompatibleGRUCell(h.shape[1]) if kind == "cudnn" else tf.nn.rnn_cell.GRUCell(h.shape[1]))
        with tf.variable_scope("cell"):
            out, _ = cell(xp, hp)
        sess = tf.Session()
        sess.run([v.assign(np.asarray(w, np.float32)) for v, w in zip(cell._vars, weights)])
        return sess.run(out, feed_dict={xp: x.astype(np.float32), hp: h.astype(np.float32)})


@pytest.mark.parametrize("I,H,V", [(100, 100, 37), (200, 100, 5), (8, 12, 64)])
def test_cudnn_compatible_gru_is_torch_grucell(oracle, I, H, V):
    rng = np.random.default_rng(I + H)
    x, h = rng.uniform(-1, 1, (V, I)), rng.uniform(-1, 1, (V, H))
    Wg, bg = rng.uniform(-0.3, 0.3, (I + H, 2 * H)), rng.uniform(-0.5, 0.5, 2 * H)
    Wcx, bcx = rng.uniform(-0.3, 0.3, (I, H)), rng.uniform(-0.5, 0.5, H)
    Wch, bch = rng.uniform(-0.3, 0.3, (H, H)), rng.uniform(-0.5, 0.5, H)
    want = _torch_gru_from_tf(Wg, bg, Wcx, bcx, Wch, bch, I, H)(torch.from_numpy(x), torch.from_numpy(h)).detach().numpy()
    got = oracle.cudnn_compatible_gru_cell(x, h, Wg, bg, Wcx, bcx, Wch, bch)
    np.testing.assert_allclose(got, want, rtol=0, atol=1e-13)
    shim = _shim_cell("cudnn", x, h, (Wg, bg, Wcx, bcx, Wch, bch))
    np.testing.assert_allclose(shim, want, rtol=0, atol=2e-6)          # the shim computes in fp32


@pytest.mark.parametrize("I,H,V", [(100, 100, 37), (300, 100, 4), (8, 12, 64)])
def test_tf_grucell_is_torch_grucell_for_a_diagonal_hidden_candidate_kernel(oracle, I, H, V):
    rng = np.random.default_rng(7 * I + H)
    x, h = rng.uniform(-1, 1, (V, I)), rng.uniform(-1, 1, (V, H))
    Wg, bg = rng.uniform(-0.3, 0.3, (I + H, 2 * H)), rng.uniform(-0.5, 0.5, 2 * H)
    Wcx, bc = rng.uniform(-0.3, 0.3, (I, H)), rng.uniform(-0.5, 0.5, H)
    d = rng.uniform(-1, 1, H)
    Wc = np.concatenate([Wcx, np.diag(d)])                             # TF layout: [x rows ; (r*h) rows]
    want = _torch_gru_from_tf(Wg, bg, Wcx, bc, np.diag(d), np.zeros(H), I, H)(torch.from_numpy(x), torch.from_numpy(h)).detach().numpy()
    got = oracle.gru_cell(x, h, Wg, bg, Wc, bc)[0]
    np.testing.assert_allclose(got, want, rtol=0, atol=1e-13)
    shim = _shim_cell("gru", x, h, (Wg, bg, Wc, bc))
    np.testing.assert_allclose(shim, want, rtol=0, atol=2e-6)
    # ... and it is NOT the cuDNN form once the hidden kernel has off-diagonal entries (the test would notice a swap)
    Wc2 = np.concatenate([Wcx, np.diag(d) + 0.2])
    other = _torch_gru_from_tf(Wg, bg, Wcx, bc, np.diag(d) + 0.2, np.zeros(H), I, H)(torch.from_numpy(x), torch.from_numpy(h)).detach().numpy()
    assert np.abs(oracle.gru_cell(x, h, Wg, bg, Wc2, bc)[0] - other).max() > 1e-3


def _closed_form_two_steps(w0, a, clip, lr=1e-3, b1=0.9, b2=0.999, eps=1e-8):
    g = a * clip / max(np.linalg.norm(a), clip)
    w1 = w0 - lr * g / (np.abs(g) + eps / np.sqrt(1 - b2))
    w2 = w1 - lr * g / (np.abs(g) + eps / np.sqrt(1 - b2 ** 2))
    return w1, w2


@pytest.mark.parametrize("scale", [0.01, 1.0, 30.0])          # below / at / far above the clip norm of 1.0
def test_clip_by_norm_and_adam_closed_form(pkg, scale):
    rng = np.random.default_rng(int(scale * 100))
    w0 = rng.uniform(-1, 1, (6, 5)); a = rng.uniform(-1, 1, (6, 5)) * scale
    want1, want2 = _closed_form_two_steps(w0, a, 1.0)
    # the package's host-side optimiser (train.TFAdam.apply_gradients + clip_by_norm_)
    w = torch.from_numpy(w0.copy())
    opt = pkg.train.TFAdam([w])
    for want in (want1, want2):
        g = [torch.from_numpy(a.copy())]
        pkg.train.clip_by_norm_(g, 1.0)
        opt.apply_gradients(g)
        np.testing.assert_allclose(w.numpy(), want, rtol=0, atol=1e-12)
    # the TF-op shim's clip_by_norm + AdamOptimizer, driven the way chem_tensorflow.py:183-191 drives them
    if SHIM not in sys.path:
        sys.path.insert(0, SHIM)
    import tensorflow as tf
    with tf.Graph().as_default():
        v = tf.Variable(w0.astype(np.float32), name="w")
        loss = tf.reduce_sum(v * a.astype(np.float32))
        optimizer = tf.train.AdamOptimizer(0.001)
        gv = optimizer.compute_gradients(loss, var_list=[v])
        clipped = [(tf.clip_by_norm(g_, 1.0), var) for g_, var in gv]
        step = optimizer.apply_gradients(clipped)
        sess = tf.Session()
        sess.run(tf.global_variables_initializer())
        for want in (want1, want2):
            sess.run(step)
            np.testing.assert_allclose(sess.run(v), want, rtol=0, atol=3e-7)      # fp32


@pytest.mark.gpu
@pytest.mark.parametrize("scale", [0.01, 30.0])
def test_fused_clip_adam_kernel_closed_form(pkg, cuda, scale):
    """ggnn_clip_adam_f32 (all variables, two launches) against the same closed form."""
    rng = np.random.default_rng(int(scale * 100) + 1)
    shapes = [(400, 100), (200,), (7, 3)]
    w0 = [rng.uniform(-1, 1, s) for s in shapes]; a = [rng.uniform(-1, 1, s) * scale for s in shapes]
    ws = [torch.from_numpy(x.astype(np.float32)).to(cuda) for x in w0]
    opt = pkg.train.TFAdam(ws)
    assert opt.fused
    w32 = [x.astype(np.float32).astype(np.float64) for x in w0]
    for step in range(2):
        opt.load_gradients([torch.from_numpy(x.astype(np.float32)).to(cuda) for x in a])
        opt.clip_and_apply(1.0)
        for i in range(len(ws)):
            want = _closed_form_two_steps(w32[i], a[i].astype(np.float32).astype(np.float64), 1.0)[step]
            np.testing.assert_allclose(ws[i].cpu().numpy(), want, rtol=0, atol=3e-7)

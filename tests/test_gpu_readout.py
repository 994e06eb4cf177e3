"""Fused readout + masked loss: ggnn_readout_loss_{fwd,bwd}_f32 against the oracle's gated_regression / task_loss
(chem_tensorflow_sparse.py:220-231, chem_tensorflow_dense.py:119-129, chem_tensorflow.py:158-170) and, for the gradients,
torch autograd of the oracle's torch restatement in float64."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _case(rng, V, G, D, with_node_mask, empty_graphs):
    sizes = rng.multinomial(V, np.ones(G) / G)
    if empty_graphs and G > 2:
        sizes[1] += sizes[2]; sizes[2] = 0                       # a graph id without nodes
    gnl = np.repeat(np.arange(G), sizes).astype(np.int32)
    hT = rng.uniform(-1, 1, (V, D)).astype(np.float32)
    h0 = rng.uniform(-1, 1, (V, D)).astype(np.float32)
    gW = rng.uniform(-0.3, 0.3, (2 * D, 1)).astype(np.float32); gb = rng.uniform(-0.2, 0.2, 1).astype(np.float32)
    tW = rng.uniform(-0.3, 0.3, (D, 1)).astype(np.float32); tb = rng.uniform(-0.2, 0.2, 1).astype(np.float32)
    y = rng.normal(0, 1, G).astype(np.float32)
    m = (rng.random(G) < 0.8).astype(np.float32)
    nm = (rng.random(V) < 0.85).astype(np.float32) if with_node_mask else None
    return gnl, hT, h0, gW, gb, tW, tb, y, m, nm


def _reference(gnl, hT, h0, gW, gb, tW, tb, y, m, nm, G):
    """float64 torch restatement (oracle_torch.gated_regression + task_loss, with the dense model's node mask)."""
    t = lambda a: torch.tensor(a, dtype=torch.float64)
    hT_, gW_, gb_, tW_, tb_ = (t(a).requires_grad_(True) for a in (hT, gW, gb, tW, tb))
    gate = torch.sigmoid(torch.cat([hT_, t(h0)], dim=-1).matmul(gW_) + gb_)
    gated = gate * (hT_.matmul(tW_) + tb_)
    if nm is not None:
        gated = gated * t(nm)[:, None]
    out = torch.zeros(G, 1, dtype=torch.float64).index_add_(0, torch.from_numpy(gnl).long(), gated)[:, 0]
    diff = (out - t(y)) * t(m)
    num, ab, ms = (0.5 * diff * diff).sum(), diff.abs().sum(), t(m).sum()
    return out, num, ab, ms, (hT_, gW_, gb_, tW_, tb_)


@pytest.mark.parametrize("V,G,D,node_mask,empty,use_ptr", [(1000, 37, 100, False, False, True), (5003, 300, 100, False, True, False),
                                                           (29 * 64, 64, 100, True, False, True), (700, 1, 64, False, False, False),
                                                           (4000, 150, 256, True, True, False), (100000, 5500, 100, False, False, True)])
def test_readout_loss_forward_backward(pkg, cuda, V, G, D, node_mask, empty, use_ptr):
    rng = np.random.default_rng(V + G)
    gnl, hT, h0, gW, gb, tW, tb, y, m, nm = _case(rng, V, G, D, node_mask, empty)
    dev = lambda a: None if a is None else torch.from_numpy(np.ascontiguousarray(a)).to(cuda)
    gptr = np.concatenate([[0], np.cumsum(np.bincount(gnl, minlength=G))]).astype(np.int32) if use_ptr else None
    from importlib import import_module
    ag = import_module(pkg.__name__ + ".autograd")
    leaves = [dev(a).requires_grad_(True) for a in (hT, gW, gb, tW, tb)]
    out, num, ab, ms = ag.readout_loss(leaves[0], dev(h0), dev(gnl), dev(gptr), dev(nm), G, leaves[1], leaves[2], leaves[3], leaves[4],
                                       dev(y), dev(m))
    w_out = torch.from_numpy(rng.normal(0, 1, G).astype(np.float32)).to(cuda)
    total = 0.7 * num + 0.3 * ab + (out * w_out).sum()           # exercises d_num, d_abs and d_out together
    total.backward()

    r_out, r_num, r_ab, r_ms, r_leaves = _reference(gnl, hT, h0, gW, gb, tW, tb, y, m, nm, G)
    (0.7 * r_num + 0.3 * r_ab + (r_out * w_out.cpu().double()).sum()).backward()
    np.testing.assert_allclose(out.detach().cpu().numpy(), r_out.detach().numpy(), atol=2e-5, rtol=1e-5)
    assert abs(float(num) - float(r_num)) <= 1e-5 * max(1.0, float(r_num))
    assert abs(float(ab) - float(r_ab)) <= 1e-5 * max(1.0, float(r_ab))
    assert float(ms) == float(r_ms)
    names = ("d_hT", "d_gate_W", "d_gate_b", "d_transform_W", "d_transform_b")
    for n, a, b in zip(names, leaves, r_leaves):
        ref = b.grad.numpy()
        scale = max(1.0, float(np.abs(ref).max()))
        np.testing.assert_allclose(a.grad.cpu().numpy(), ref, atol=2e-5 * scale, rtol=1e-4, err_msg=n)

    # deterministic: a second evaluation is bit-identical (no atomics, fixed reduction orders)
    leaves2 = [dev(a).requires_grad_(True) for a in (hT, gW, gb, tW, tb)]
    out2, num2, ab2, _ = ag.readout_loss(leaves2[0], dev(h0), dev(gnl), dev(gptr), dev(nm), G, leaves2[1], leaves2[2], leaves2[3],
                                         leaves2[4], dev(y), dev(m))
    (0.7 * num2 + 0.3 * ab2 + (out2 * w_out).sum()).backward()
    assert torch.equal(out, out2) and float(num) == float(num2)
    for a, b in zip(leaves, leaves2):
        assert torch.equal(a.grad, b.grad)


def test_model_readout_paths_agree(pkg, oracle, cuda):
    """The model's fused readout (inference and training form) equals the oracle's gated_regression + task_loss, and the
    atomic kernel kept for unsorted graph_nodes_list."""
    ms = pkg.synthetic_qm9(300, mean_nodes=12, seed=8)
    cfg = {"task_ids": [0], "batch_size": 2000}
    model = pkg.SparseGGNNChemModel({"--quiet": True, "--device": "cuda:0", "train_data": None, "valid_data": ms, "--config": cfg})
    feeds = list(model.make_minibatch_iterator(model.valid_data, is_training=False))
    assert len(feeds) >= 2
    for feed in feeds[:2]:
        with torch.no_grad():
            loss = float(model.forward_batch(feed))
            out = model.output.cpu().numpy()
            final = model.ops['final_node_representations']
            g = model.weights['regression_gate_task0'].params; t = model.weights['regression_transform_task0'].params
            atomic = pkg.ops.gated_readout(final, feed['initial_node_representation'], feed['graph_nodes_list'], feed['num_graphs'],
                                           g["weights"][0], g["biases"][0], t["weights"][0], t["biases"][0]).cpu().numpy()
        f = lambda x: x.cpu().numpy().astype(np.float64)
        pred = oracle.gated_regression(f(final), f(feed['initial_node_representation']), feed['graph_nodes_list'].cpu().numpy(),
                                       feed['num_graphs'], f(g["weights"][0]), f(g["biases"][0]), f(t["weights"][0]), f(t["biases"][0]))
        np.testing.assert_allclose(out, pred, atol=2e-5, rtol=1e-5)
        np.testing.assert_allclose(atomic, pred, atol=2e-5, rtol=1e-5)
        wl, wa = oracle.task_loss(pred, f(feed["target_values"])[0], f(feed["target_mask"])[0])
        assert abs(loss - wl) <= 1e-5 * max(1.0, abs(wl))
        assert abs(float(model.ops['accuracy_task0']) - wa) <= 1e-5 * max(1.0, abs(wa))


@pytest.mark.parametrize("V,K,n", [(3000, 200, 1), (1500, 100, 1), (777, 128, 3), (500, 64, 4)])
def test_op_by_op_readout_linear_on_package_kernels(pkg, cuda, V, K, n):
    """utils.MLP's layer (the op-by-op readout of chem_tensorflow_sparse.py:220-231 / utils.py:64-70, used where the fused readout
    does not apply) runs on the package's GEMM / X^T dY / column-sum kernels -- no vendor BLAS: forward and all three gradients
    against torch autograd in float64."""
    rng = np.random.default_rng(V + K + n)
    x = torch.from_numpy(rng.uniform(-1, 1, (V, K)).astype(np.float32)).to(cuda).requires_grad_(True)
    W = torch.from_numpy(rng.uniform(-0.3, 0.3, (K, n)).astype(np.float32)).to(cuda).requires_grad_(True)
    b = torch.from_numpy(rng.uniform(-0.3, 0.3, (n,)).astype(np.float32)).to(cuda).requires_grad_(True)
    g = torch.from_numpy(rng.uniform(-1, 1, (V, n)).astype(np.float32)).to(cuda)
    y = pkg.utils._TallLinear.apply(x, W, b)
    y.backward(g)
    x64, W64, b64 = (t.detach().double().requires_grad_(True) for t in (x, W, b))
    y64 = x64.matmul(W64) + b64
    y64.backward(g.double())
    assert float((y.detach().double() - y64).abs().max()) < 2e-5
    for got, want in ((x.grad, x64.grad), (W.grad, W64.grad), (b.grad, b64.grad)):
        scale = float(want.abs().max()) + 1e-12
        assert float((got.double() - want).abs().max()) <= 3e-5 * scale + 1e-6

"""Host logic of the per-launch operand-format choice of the fused GRU forward (formats.py): the two-piece f16 format is selected
only where its operand range is PROVEN, the exact bf16x3 split everywhere else.  No GPU: maxima of CPU tensors."""
import math

import numpy as np
import pytest
import torch


def test_layer_format_bounds(pkg):
    f = pkg.formats
    with f.forced("auto"):
        S = f.state_bound(1.0, "tanh")
        assert S == 1.0 and f.state_bound(7.5, "tanh") == 7.5 and f.state_bound(0.2, "TANH") == 1.0
        inc = f.incoming_bound(S, 100, 0.11, 0.0, True)                     # glorot edge weights: sqrt(6/500)
        assert inc == pytest.approx(11.0) and f.layer_format(S, inc, 0.2) == f.F16X2
        # every way out of the range, or out of the proof, is the exact format
        assert f.layer_format(S, inc, 256.0) == f.BF16X3                     # a GRU weight the x 2^8 packing would saturate
        assert f.layer_format(S, inc, 255.7) == f.BF16X3                     # (inside the safety margin)
        assert f.layer_format(S, inc, 255.0) == f.F16X2
        assert f.layer_format(70000.0, inc, 0.2) == f.BF16X3                 # max |h0| beyond f16
        assert f.layer_format(S, f.incoming_bound(S, 100, 700.0, 0.0, True), 0.2) == f.BF16X3       # D max|W_edge| S > 65504
        assert f.layer_format(S, f.incoming_bound(S, 100, 0.11, 7e4, True), 0.2) == f.BF16X3        # edge bias
        assert f.layer_format(S, f.incoming_bound(S, 100, 0.11, 0.0, False), 0.2) == f.BF16X3       # sum aggregation: no bound
        assert f.state_bound(1.0, "relu") == math.inf and f.layer_format(math.inf, inc, 0.2) == f.BF16X3
        for bad in (float("nan"), math.inf):
            assert f.layer_format(f.state_bound(bad, "tanh"), inc, 0.2) == f.BF16X3
            assert f.layer_format(S, f.incoming_bound(S, 100, bad, 0.0, True), 0.2) == f.BF16X3
            assert f.layer_format(S, inc, bad) == f.BF16X3
        # dropout divides by the keep probability: weight dropout once, state dropout once per timestep
        assert f.incoming_bound(S, 100, 0.11, 0.0, True, 0.8) == pytest.approx(11.0 / 0.8)
        assert f.state_bound(1.0, "tanh", 8, 0.5) == 256.0
    with f.forced(f.BF16X3):
        assert f.policy() == "exact" and f.layer_format(1.0, 1.0, 0.1) == f.BF16X3
    with f.forced(f.F16X2):
        assert f.policy() == "force2" and f.layer_format(math.inf, math.inf, 1e9) == f.F16X2    # (unchecked: experiments only)
    with pytest.raises(ValueError):
        with f.forced("fast"):
            pass


def test_absmax_of_cpu_tensors_propagates_non_finite(pkg):
    f = pkg.formats
    t = torch.tensor([[0.5, -3.0], [2.0, 1.0]])
    assert f.absmax([t, torch.zeros(0), torch.tensor([1.0, float("inf")]), torch.tensor([float("nan"), 9.0])])[:3] == [3.0, 0.0, math.inf]
    assert math.isnan(f.absmax([torch.tensor([float("nan"), 9.0])])[0])
    w = torch.ones(4, 4)
    assert f.weight_absmax([w]) == [1.0]
    w.mul_(3.0)                                        # an in-place update bumps the version: re-measured
    assert f.weight_absmax([w, w[:2]]) == [3.0, 3.0]


def test_adam_step_bound_holds_against_adversarial_gradients(pkg):
    """|delta w| of one TF-1.3 Adam step <= lr * C for ANY gradient history (the bound the training-time weight maxima lean on)."""
    f = pkg.formats
    lr, b1, b2, eps = 1e-3, 0.9, 0.999, 1e-8
    bound = f.adam_step_bound(lr, b1, b2)
    assert 3.0 * lr < bound < 8.0 * lr
    assert f.adam_step_bound(lr, 0.99, 0.9) == math.inf                      # b1^2 >= b2: no bound -> measured every step
    rng = np.random.default_rng(0)
    worst = 0.0
    for trial in range(60):
        n = 400
        kind = trial % 4
        if kind == 0:
            g = rng.normal(size=n) * 10.0 ** rng.uniform(-6, 3)
        elif kind == 1:                                                       # long silence, then constant large gradients
            g = np.concatenate([np.full(n // 2, 1e-12), np.full(n - n // 2, 1e3)])
        elif kind == 2:                                                       # geometric ramps (the Cauchy-Schwarz extremal shape)
            g = (b1 / b2) ** -np.arange(n, dtype=np.float64) * 1e-30
        else:
            g = rng.choice([0.0, 1.0], size=n, p=[0.97, 0.03]) * rng.normal(size=n)
        m = v = 0.0
        for t, gt in enumerate(g, 1):
            m = b1 * m + (1 - b1) * gt
            v = b2 * v + (1 - b2) * gt * gt
            lr_t = lr * math.sqrt(1 - b2 ** t) / (1 - b1 ** t)
            worst = max(worst, abs(lr_t * m / (math.sqrt(v) + eps)))
    assert worst <= bound and worst > 0.5 * lr


class _FakeAdam:
    def __init__(self, vs, lr=1e-3):
        self.vars, self.lr, self.b1, self.b2, self.t, self.fused = vs, lr, 0.9, 0.999, 0, True

    def step(self):
        self.t += 1
        for v in self.vars:
            v.add_(1e-4)                                 # one in-place write per variable and step, like clip_and_apply's bump


def test_training_weight_bounds_follow_the_optimizer_and_notice_foreign_writes(pkg, monkeypatch):
    f = pkg.formats
    calls = []
    real = f.absmax
    monkeypatch.setattr(f, "absmax", lambda ts: (calls.append(len(ts)), real(ts))[1])
    ws = [torch.full((3, 3), 0.5), torch.full((2,), -2.0)]
    opt = _FakeAdam(ws)
    tb = f.TrainingWeightBounds()
    assert tb.get(ws, opt) == [0.5, 2.0] and len(calls) == 1
    step_b = f.adam_step_bound(opt.lr, opt.b1, opt.b2)
    for k in range(1, 5):
        opt.step()
        got = tb.get(ws, opt)
        assert len(calls) == 1                                               # no new measurement: bounded by k Adam steps
        assert got == pytest.approx([0.5 + k * step_b, 2.0 + k * step_b]) and all(g >= float(w.abs().max()) for g, w in zip(got, ws))
    ws[0].fill_(300.0)                                                       # a checkpoint restore: not the optimizer's write
    assert tb.get(ws, opt)[0] == 300.0 and len(calls) == 2
    for _ in range(f.REMEASURE_STEPS):
        opt.step()
    tb.get(ws, opt)
    assert len(calls) == 3                                                   # periodic re-measurement


def _model(pkg, **over):
    ms = pkg.synthetic_qm9(12, mean_nodes=6, seed=2)
    params = {"layer_timesteps": [2, 1], "residual_connections": {"1": [0]}}
    params.update(over)
    return pkg.SparseGGNNChemModel({"--quiet": True, "--device": "cpu", "train_data": None, "valid_data": ms, "--config": params})


def test_model_selects_f16x2_only_with_a_proof(pkg):
    f = pkg.formats
    if not f.split_path():
        pytest.skip("f32 matrix path: the operand format does not apply")
    m = _model(pkg)
    V = 12
    h0 = torch.zeros(V, 100); h0[:, :5] = torch.eye(5).repeat(3, 1)[:V]
    m.placeholders["initial_node_representation"] = h0
    with f.forced("auto"):
        # (mean aggregation bounds the aggregate only with an in-degree table that covers the message index: none fed yet)
        assert m.gru_formats(h0) == [f.BF16X3, f.BF16X3]
        import types
        m.placeholders["num_incoming_edges_per_type"] = torch.ones(V, 4)
        m.placeholders["message_index"] = types.SimpleNamespace(row_ptr=torch.arange(V + 1, dtype=torch.int32) * 3)   # in-degree 3 <= 4
        assert m.gru_formats(h0) == [f.F16X2, f.F16X2] and m.last_gru_format_bounds["proven"]
        m.gnn_weights.rnn_cells[1].gates_kernel[3, 7] = 400.0                # one weight outside the x 2^8 range: that layer only
        assert m.gru_formats(h0) == [f.F16X2, f.BF16X3] and not m.last_gru_format_bounds["proven"]
        m.gnn_weights.rnn_cells[1].gates_kernel[3, 7] = 0.1
        assert m.gru_formats(h0) == [f.F16X2, f.F16X2]
        h0[2, 1] = 1e5                                                        # same tensor, new version: re-measured
        assert m.gru_formats(h0) == [f.BF16X3, f.BF16X3]
        h0[2, 1] = float("nan")
        assert m.gru_formats(h0) == [f.BF16X3, f.BF16X3]
        h0[2, 1] = 1.0
        assert m.gru_formats(h0) == [f.F16X2, f.F16X2]
        m.params["use_edge_msg_avg_aggregation"] = False                      # sum aggregation: no bound without the degrees
        assert m.gru_formats(h0) == [f.BF16X3, f.BF16X3]
        m.params["use_edge_msg_avg_aggregation"] = True
        m.params["graph_rnn_activation"] = "relu"
        assert m.gru_formats(h0) == [f.BF16X3, f.BF16X3]
    m.params["graph_rnn_activation"] = "tanh"
    with f.forced(f.BF16X3):
        assert m.gru_formats(h0) == [f.BF16X3, f.BF16X3]


def test_nin_consistency_is_declared_or_checked(pkg):
    """Advisor (round 5): the bound on the aggregated messages under mean aggregation needs sum_t nin[v, t] >= in-degree(v).  The
    packers declare their table with the batch (tied to the tensor's identity and version); any other table is checked against the
    message index once -- zeros, fractions, negative or non-finite entries fail, and the GRU then runs in the exact format."""
    f = pkg.formats
    row_ptr = torch.tensor([0, 2, 2, 5], dtype=torch.int32)                 # in-degrees 2, 0, 3
    good = torch.tensor([[1., 1.], [0., 0.], [3., 0.]])
    feed = {"initial_node_representation": torch.zeros(3, 4), "num_incoming_edges_per_type": good}
    assert f.nin_consistent(feed, row_ptr)                                   # checked: consistent
    for bad in (torch.zeros(3, 2), torch.tensor([[1., 1.], [0., 0.], [1.5, 1.]]), torch.tensor([[3., -1.], [0., 0.], [3., 0.]]),
                torch.tensor([[1., 1.], [0., float("nan")], [3., 0.]])):
        assert not f.nin_consistent({"num_incoming_edges_per_type": bad}, row_ptr)
    assert not f.nin_consistent({"num_incoming_edges_per_type": good}, None)             # nothing to check against
    assert not f.nin_consistent({"num_incoming_edges_per_type": good}, row_ptr[:-1])     # another batch's index
    # a declaration holds while it is about THIS tensor at THIS version
    z = torch.zeros(3, 2)
    feed = f.declare_h0_absmax({"initial_node_representation": torch.zeros(3, 4), "num_incoming_edges_per_type": z}, 1.0)
    assert f.nin_consistent(feed, row_ptr)                                   # (the packer's word, not the contents)
    z.add_(0.0)                                                              # an in-place write bumps the version
    assert not f.nin_consistent(feed, row_ptr)
    feed["num_incoming_edges_per_type"] = torch.zeros(3, 2)                  # a replaced table
    assert not f.nin_consistent(feed, row_ptr)


def test_adjacency_absmax_follows_the_fed_tensor(pkg):
    f = pkg.formats
    A = torch.zeros(2, 4, 5, 5); A[0, 1, 2, 3] = 1.0
    assert f.adjacency_absmax(A) == 1.0
    A[1, 0, 0, 0] = -50.0                                                    # a weighted edge: new version, measured again
    assert f.adjacency_absmax(A) == 50.0
    A[0, 0, 1, 1] = float("nan")
    assert f.adjacency_absmax(A) != f.adjacency_absmax(A)                    # NaN stays NaN (fails every <= of the policy)
    assert f.adjacency_absmax(torch.zeros(0, 4, 5, 5)) == 0.0

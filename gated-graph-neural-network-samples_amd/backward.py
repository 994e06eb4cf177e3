"""Backward pass of one propagation timestep (what TF autodiff derives from
chem_tensorflow_sparse.py:153-216 via optimizer.compute_gradients, chem_tensorflow.py:184) -- every launch a
hand-written HIP kernel of libggnn_hip.so; torch only allocates the buffers.

Forward (training) = the inference kernels: compacted message transform on the active (node, type) pairs, then the fused GRU
with the segment sum gathered inside the launch, in its training form (r, u, c and the gathered segment written for the backward
pass); layers with an edge bias keep the stand-alone segment sum (bias epilogue).

Backward, per timestep (nx = residual inputs + 1, K = (nx+1) D):
  1. ggnn_gru_bwd_stage1_f32      g, h, r, u, c -> dpc, dpu (= dpg[:, D:]), dh = g*u, r*h           (one element-wise pass)
  2. ggnn_xty_f32                 dWc = [x.. | incoming | r*h]^T dpc        (segment pointers: no [V,K] concat)
     ggnn_colsum_f32              dbc
  3. ggnn_gru_bwd_dx_cand_f32     dpc Wc^T with the stage-2 algebra in the epilogue: dx (x part), dh += drh*r, dpr -> dpg[:, :D]
  4. ggnn_xty_f32 / ggnn_colsum_f32   dWg = [x.. | incoming | h]^T dpg, dbg
  5. ggnn_gru_bwd_dx_gates_f32    dpg Wg^T: dx += (residual columns), d_incoming = (dx + .) / (deg + 1e-7), dh += (h columns)
  6. ggnn_gather_segment_sum_f32 on the transpose index: dHc[r] = sum of d_incoming[dst] over the messages leaving pair r
  7. ggnn_msg_transform_compact_f32 on W^T: Z = dHc W_t^T;  ggnn_gather_segment_sum_acc_f32: dh[v] += sum_t Z[row(v,t)]
  8. ggnn_xty_f32 (row-gathered, one batch per edge type): dW_t = h[pair_node[rows_t]]^T dHc[rows_t]
No vendor-BLAS call and no torch arithmetic kernel is left on this path (hidden sizes without a compacted transform keep the
dense-form fallback at the end of this file).
"""
from __future__ import annotations

import contextlib
import os
import weakref

import torch

from . import _lib, ops
from ._lib import check
from .utils import SMALL_NUMBER

# Training forward/backward on the compacted message transform (default wherever the hidden size has one) or on the dense
# [V, T*D] form (GGNN_TRAIN_COMPACT=0; always for other hidden sizes).
USE_COMPACT_TRANSFORM = os.environ.get("GGNN_TRAIN_COMPACT", "1") != "0"

# (dense-form fallback: weight gradients X^T dY on the row-split kernel ggnn_gemm_tn_f32 -- no vendor BLAS anywhere in the package)


# ---- weight gradients on a side stream -----------------------------------------------------------------------------------
# The weight-gradient products (2., 4., 8. above) are LEAVES of the backward graph: nothing in the backward pass reads them.  The
# fused training step (train.train_step) therefore hands the propagation steps a SINK -- for every trainable variable the view of
# the optimizer's flat gradient buffer that belongs to it -- and the products of a timestep are then launched on a second HIP
# stream, added into those views there, and not returned to autograd at all.  They run next to the gate/transform/segment-sum
# chain of the next timestep's backward on the main stream, whose phases are bound by different things (the fused GRU backward
# spends a third of each pass waiting for HBM; the products are matrix-pipe bound).  The sums are the ones autograd would have
# formed, in the same order (timesteps in backward order, one in-order stream), so the result is bit-identical.
USE_WGRAD_STREAM = os.environ.get("GGNN_WGRAD_STREAM", "1") != "0"
# The whole optimisation step of the default model on the native launch sequences (train_native.py / csrc/ggnn_train.hip);
# GGNN_NATIVE_STEP=0: through torch.autograd (this file), which stays the path of every other model variant
USE_NATIVE_STEP = os.environ.get("GGNN_NATIVE_STEP", "1") != "0"
# Training forward with the segment sum gathered inside the GRU launch (ggnn_gru_packed_gather_train_f32); 0: separate launch
TRAIN_GATHER_IN_GRU = os.environ.get("GGNN_TRAIN_GATHER_IN_GRU", "1") != "0"


class _WeightGradSink:
    def __init__(self):
        self.targets = None          # {data_ptr of a variable: float32 buffer of its shape}
        self.stream = None
        self.used = set()            # data_ptrs that received a gradient while the sink was active
        self.masks = {}              # data_ptr -> (keep_prob, seed): dropout mask to apply to the accumulated gradient at the end

    def target(self, ptr, shape):
        """The buffer registered for the variable at `ptr`, if it holds as many elements as `shape` (the propagation step sees the
        edge weights as a [T,D,D] view of the [T*D, D] variable: same memory, same element order)."""
        if self.targets is None or ptr is None:
            return None
        t = self.targets.get(ptr)
        n = 1
        for d in shape:
            n *= int(d)
        if t is None or t.numel() != n:
            return None
        return t

    def finish(self):
        """After the side stream's products have been ordered before the current stream: gradients that were accumulated in
        front of a weight-dropout mask (chem_tensorflow_sparse.py:91: d variable = mask/keep * d masked weights, one mask per layer
        and step) are masked in place, once per variable instead of once per timestep."""
        for ptr, mask in self.masks.items():
            if mask is not None:                       # (None: the variable's contributions were masked one by one, see defer_mask)
                t = self.targets[ptr]
                ops.dropout(t, mask[0], mask[1], out=t)
        self.masks = {}

    def defer_mask(self, ptr, keep, seed) -> bool:
        """May the contribution about to be added to the buffer of `ptr` stay unmasked until finish()?  Only while every
        contribution of the step shares ONE (keep, seed): the sparse model draws one mask per layer and step
        (chem_tensorflow_sparse.py:91), the dense model one per TIMESTEP on the same shared variable
        (chem_tensorflow_dense.py:104) -- sum_i mask_i * dW_i is not mask * sum_i dW_i.  On the first different mask the pending
        one is applied to what has accumulated so far (on the side stream, behind the products that wrote it) and the variable
        switches to per-contribution masking for the rest of the step."""
        state = self.masks.get(ptr, "unset")
        mask = (float(keep), int(seed))
        if state == "unset" or state == mask:
            self.masks[ptr] = mask
            return True
        if state is not None:
            t = self.targets[ptr]
            _on_side_stream([], lambda: ops.dropout(t, state[0], state[1], out=t))
            self.masks[ptr] = None
        return False

    def add(self, ptr, target, value):
        target.add_(value.view_as(target))
        self.used.add(ptr)

    def mark(self, *ptrs):
        self.used.update(ptrs)


_SINK = _WeightGradSink()


def side_stream(device=None) -> "torch.cuda.Stream":
    """THE side stream of the training step (one per process): weight-gradient products of the autograd path and of the native
    step, and the read-back of a step's statistics.  One shared object because streams are multiplexed onto few hardware queues."""
    if _SINK.stream is None:
        _SINK.stream = torch.cuda.Stream(device)        # (a high-priority stream measured the same: 6.55-6.77 ms either way)
    return _SINK.stream


@contextlib.contextmanager
def weight_gradient_sink(targets):
    """While active, PropagationStepFn.backward adds the gradients of the variables in `targets` ({variable.data_ptr(): buffer})
    into the buffers on a side stream and returns None for them; the caller zeroes the buffers beforehand (on the current stream)
    and must make the current stream wait for `sink.stream` before it reads them.  Yields the sink (`.used`, `.stream`)."""
    if not (USE_WGRAD_STREAM and torch.cuda.is_available()):
        yield None
        return
    side_stream()
    _SINK.targets, _SINK.used, _SINK.masks = dict(targets), set(), {}
    try:
        yield _SINK
    except BaseException:
        _SINK.targets, _SINK.masks = None, {}          # the body's own error is the one to report
        raise
    else:
        pending = any(m is not None for m in _SINK.masks.values())
        _SINK.targets, _SINK.masks = None, {}
        if pending:
            raise RuntimeError("weight_gradient_sink left without sink.finish(): masked gradients pending")


def _on_side_stream(tensors, fn):
    """Run fn() on the sink's stream after everything queued so far on the current stream; `tensors` (allocated on the current
    stream, read by fn) are kept from being recycled until the side stream is done with them."""
    side = _SINK.stream
    ev = torch.cuda.Event()
    ev.record()
    with torch.cuda.stream(side):
        side.wait_event(ev)
        fn()
    for t in tensors:
        t.record_stream(side)


def _tn(x: torch.Tensor, dy: torch.Tensor) -> torch.Tensor:
    """x^T @ dy over ~1e5 rows on ggnn_gemm_tn_f32 (rows split over the whole GPU, deterministic reduction); dy wider than the
    kernel's 512 columns goes through in column blocks (views: the kernel takes any row stride)."""
    N = dy.shape[1]
    if N <= 512:
        return ops.gemm_tn(x, dy)
    step = 512
    return torch.cat([ops.gemm_tn(x, dy[:, c:min(c + step, N)]) for c in range(0, N, step)], dim=1)


def weight_grad(x_segs, dy: torch.Tensor):
    """(dW [K, N], db [N]) = (concat(x_segs, dim=1)^T dy, column sums of dy) for a weight matrix applied to [x_0 | x_1 | ..].
    Shapes of the fused kernels' hidden sizes: ONE ggnn_xty_f32 launch, the bias gradient as its ones row.  Wider ones (hidden
    128 / 192 / 256): per segment on the row-split kernel ggnn_gemm_tn_f32, the bias gradient by ggnn_colsum_f32."""
    D, N = x_segs[0].shape[1], dy.shape[1]
    if D <= 104 and N <= 208:
        K = len(x_segs) * D
        w = ops.xty(list(x_segs), dy, ones_row=True)
        return w[:K], w[K]
    return torch.cat([_tn(x, dy) for x in x_segs], dim=0), ops.colsum(dy)


def _source_index(index: "ops.MessageIndex", num_nodes: int) -> "ops.MessageIndex":
    src_index = getattr(index, "_source_index", None)
    if src_index is None:
        lists = [index.adj[index.type_off[t]:index.type_off[t + 1]] for t in range(index.num_edge_types)]
        src_index = ops.build_source_index(lists, num_nodes)
        index._source_index = src_index
    return src_index


class _TransposeCache:
    """W^T (contiguous) per weight tensor version: the backward of every timestep of a layer multiplies by the same
    transposed weights, so the transpose copy is made once per layer and optimisation step."""

    def __init__(self):
        self._t = {}

    def get(self, W: torch.Tensor, dims=(0, 1)) -> torch.Tensor:
        key = (id(W), W._version, dims)
        hit = self._t.get(key)
        if hit is not None and hit[0]() is W:
            return hit[1]
        if len(self._t) > 256:
            self._t.clear()
        Wt = W.transpose(*dims).contiguous()
        self._t[key] = (weakref.ref(W), Wt)
        return Wt


_TRANSPOSED = _TransposeCache()


class _MaskedWeights:
    """The dropped-out edge weights of a layer (chem_tensorflow_sparse.py:91), formed once per (variable version, mask) and shared
    by the layer's timesteps."""

    def __init__(self):
        self._t = {}

    def get(self, W: torch.Tensor, keep: float, seed: int) -> torch.Tensor:
        base = W._base if W._base is not None else W
        key = (id(base), W.data_ptr(), tuple(W.shape))
        hit = self._t.get(key)
        if hit is not None and hit[0]() is base and hit[1] == (base._version, keep, seed):
            return hit[2]
        if len(self._t) > 64:
            self._t.clear()
        Wm = ops.dropout(W.detach().contiguous(), keep, seed)
        self._t[key] = (weakref.ref(base), (base._version, keep, seed), Wm)
        return Wm


_MASKED = _MaskedWeights()


class PropagationStepFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, h, index, nin, edge_weights, edge_biases, use_avg, activation, Wg, bg, Wc, bc, ew_mask, gru_fmt, *residuals):
        h = h.contiguous()
        D = h.shape[1]
        ctx.comp = None
        ctx.ew_mask = ew_mask
        # (addresses of the variables as they were passed in: keys of the weight-gradient sink)
        ctx.var_ptrs = tuple(None if t is None else t.data_ptr() for t in (edge_weights, edge_biases, Wg, bg, Wc, bc))
        ctx.var_shapes = tuple(None if t is None else tuple(t.shape) for t in (edge_weights, edge_biases, Wg, bg, Wc, bc))
        if ew_mask is not None:
            # `edge_weights` is the variable: multiply by its masked copy; backward() sends the gradient back through the mask
            edge_weights = _MASKED.get(edge_weights, float(ew_mask[0]), int(ew_mask[1]))
        if USE_COMPACT_TRANSFORM and ops.compact_supported(D) and D <= 104:
            # transform only the (node, type) pairs that emit a message (~1.2 V rows instead of T V)
            from .autograd import _PACKED
            comp = getattr(index, "_compact", None)
            if comp is None:
                comp = index._compact = ops.build_compact_sources(index)
            ctx.comp = comp
            ew = edge_weights if edge_weights.is_contiguous() else edge_weights.contiguous()
            H = ops.msg_transform_compact_packed(h, _PACKED.edge(ew), ew.shape[0], comp)
            edge_weights = ew
            if TRAIN_GATHER_IN_GRU and edge_biases is None and ops.gru_gather_fused(D) and len(residuals) + 1 <= ops.GRU_FUSED_MAX_INPUTS:
                # the segment sum gathered inside the GRU launch, as on the inference path; the kernel also writes r, u, c and the
                # gathered segment, which the backward pass needs (one launch and one pass over `incoming` less per timestep)
                save = {}
                h_new = ops.gru_packed_gather(list(residuals), h, _PACKED.gru(Wg, Wc, len(residuals) + 1, D, gru_fmt), bg, bc, H, index,
                                              comp.gather_row, nin if use_avg else None, activation, save=save, fmt=gru_fmt)
                del H
                ctx.index, ctx.use_avg, ctx.activation, ctx.has_bias = index, use_avg, activation.lower(), False
                ctx.save_for_backward(h, nin, edge_weights, Wg, Wc, save["incoming"], save["r"], save["u"], save["c"], *residuals)
                return h_new
            incoming = ops.gather_segment_sum_compact(H, index, comp, nin, edge_biases, use_avg)
        else:
            edge_weights = edge_weights.contiguous()
            H = ops.msg_transform(h, edge_weights)
            incoming = ops.gather_segment_sum(H, index, nin, edge_biases, use_avg)
        del H
        save = {}
        h_new = ops.gru(list(residuals) + [incoming], h, Wg, bg, Wc, bc, activation, save=save, fmt=gru_fmt)
        ctx.index, ctx.use_avg, ctx.activation, ctx.has_bias = index, use_avg, activation.lower(), edge_biases is not None
        ctx.save_for_backward(h, nin, edge_weights, Wg, Wc, incoming, save["r"], save["u"], save["c"], *residuals)
        return h_new

    @staticmethod
    def backward(ctx, g):
        if ctx.comp is None:
            return _backward_dense_form(ctx, g)
        lib = _lib.load()
        h, nin, W, Wg, Wc, incoming, r, u, c, *residuals = ctx.saved_tensors
        V, D = h.shape
        T = W.shape[0]
        nx = len(residuals) + 1
        K = (nx + 1) * D
        g = g.contiguous()
        dev = h.device
        st = torch.cuda.current_stream().cuda_stream
        act = ops.ACT_IDS[ctx.activation]
        xs = list(residuals) + [incoming]

        if ops.gru_bwd_is_fused(D) and nx <= ops.GRU_FUSED_MAX_INPUTS:
            # ---- 1.-5. in ONE launch: gate algebra, dX products, mean aggregation; dpc / dpg / r*h written for the dW products
            from .autograd import _PACKED
            dpc, dpg, rh, dh, dxs = ops.gru_bwd_fused(g, h, r, u, c, _PACKED.gru_bwd(Wg, Wc, nx, D), nin, ctx.use_avg, nx, ctx.activation)
            dinc = dxs[-1]
            d_res = dxs[:-1]
            Kx = (nx + 1) * D
            pW, pb, pWg, pbg, pWc, pbc = ctx.var_ptrs
            sW, sb, sWg, sbg, sWc, sbc = ctx.var_shapes
            tg = [_SINK.target(p_, s_) for p_, s_ in ((pWg, sWg), (pbg, sbg), (pWc, sWc), (pbc, sbc))]
            if all(t is not None for t in tg):
                def gru_weight_products():             # (the reduction kernel adds into the gradient buffers; bias = the ones row)
                    ops.xty(xs + [rh], dpc, ones_row=True, add_to=tg[2], add_bias_to=tg[3])
                    ops.xty(xs + [h], dpg, ones_row=True, add_to=tg[0], add_bias_to=tg[1])
                    _SINK.mark(pWg, pbg, pWc, pbc)
                _on_side_stream(xs + [rh, h, dpc, dpg], gru_weight_products)
                dWc = dbc = dWg = dbg = None
            else:
                wc = ops.xty(xs + [rh], dpc, ones_row=True); dWc, dbc = wc[:Kx], wc[Kx]      # bias gradient = the ones row
                wg = ops.xty(xs + [h], dpg, ones_row=True); dWg, dbg = wg[:Kx], wg[Kx]
        else:
            dpc, dpg, dh, dinc, d_res, dWc, dbc, dWg, dbg = _gru_backward_unfused(lib, g, h, r, u, c, Wg, Wc, nin, xs, nx, T, act, ctx.use_avg, st)
        dbias = None
        if ctx.has_bias:                                                       # :202-204  incoming += nin @ edge_biases
            tb = _SINK.target(ctx.var_ptrs[1], ctx.var_shapes[1])
            if tb is not None:
                _on_side_stream([dinc, nin], lambda: _SINK.add(ctx.var_ptrs[1], tb, ops.xty([dinc], nin).t()))
            else:
                dbias = ops.xty([dinc], nin).t().contiguous()                  # (dinc^T nin)^T = nin^T dinc   [T, D]

        # ---- 6.-8. segment sum and compacted transform  Hc[r] = h[node(r)] W_type(r)  on the same R rows
        tW = _SINK.target(ctx.var_ptrs[0], ctx.var_shapes[0])
        # the raw product may go straight into the sink's buffer when no mask applies, or when the mask can wait for sink.finish()
        direct = tW is not None and (ctx.ew_mask is None or _SINK.defer_mask(ctx.var_ptrs[0], ctx.ew_mask[0], ctx.ew_mask[1]))
        dW = transform_backward(ctx.index, ctx.comp, h, W, dinc, dh, sink=(ctx.var_ptrs[0], tW) if direct else None)
        if ctx.ew_mask is not None and not direct:
            dW = ops.dropout(dW.contiguous(), float(ctx.ew_mask[0]), int(ctx.ew_mask[1]))
            if tW is not None:                          # this timestep's own mask (dense model), then into the buffer
                masked, ptr0 = dW, ctx.var_ptrs[0]
                _on_side_stream([masked], lambda: _SINK.add(ptr0, tW, masked))
                dW = None
        return (dh, None, None, dW, dbias, None, None, dWg, dbg, dWc, dbc, None, None, *d_res)


def transform_backward(index, comp, h, W, dinc, dh, message_weights=None, sink=None):
    """Backward of  incoming[v] = sum over the messages into v of (w_m *) h[src_m] W_type(m)  given dinc = dL/d incoming:
    adds the state gradient to `dh` in place and returns dW [T,D,D].  On the compact (source node, type) rows whatever form
    the forward transform had:  dHc[r] = sum over the messages leaving pair r of (w_m *) dinc[dst_m]  (transpose gather),
    Z = dHc W_t^T (the compacted transform kernel on W^T),  dh[v] += sum_t Z[row(v,t)],  dW_t = h[pair_node[rows_t]]^T dHc[rows_t].
    message_weights: per-message weights w (by message id; the attention coefficients) or None.
    sink: (variable address, buffer) of the active weight-gradient sink: dW is then added to the buffer on the side stream and
    None is returned."""
    from .autograd import _PACKED
    T = W.shape[0]
    bwd = ops.compact_backward(index, comp)
    R = comp.num_rows
    if not R:
        if sink is not None:
            _SINK.mark(sink[0])                      # a zero gradient is still a gradient: Adam's m / v decay (TF semantics)
            return None
        return torch.zeros_like(W)
    if message_weights is None:
        dHc = ops.segment_sum_rows_by_index(dinc, bwd.rows_index)                              # [R,D], transpose gather
    else:
        dHc = ops.weighted_segment_sum(dinc, bwd.rows_index, bwd.rows_index.msg, message_weights)
    D_ = h.shape[1]
    if ops.compact_supported(D_):
        Z = ops.msg_transform_compact_packed(dHc, _PACKED.edge(_TRANSPOSED.get(W, (1, 2))), T, bwd.identity)   # dHc W_t^T
    else:
        # hidden sizes without a compacted transform kernel (96, 160, 200, 224, 288, 300 ...: the reference accepts any,
        # chem_tensorflow_sparse.py:46-50): the rows of a type are one contiguous range -> one generic GEMM per type
        WT = _TRANSPOSED.get(W, (1, 2))
        Z = torch.empty((max(R, 1), D_), dtype=torch.float32, device=h.device)
        for t in range(T):
            lo, hi = int(comp.type_row_off[t]), int(comp.type_row_off[t + 1])
            if hi > lo:
                ops.gemm([dHc[lo:hi]], WT[t], out=Z[lo:hi])
    ops.segment_sum_rows_acc(Z[:R], bwd.node_index, dh)                                      # sum over a node's types
    D = h.shape[1]
    if D > 128:
        # the row-gathered product kernel takes <= 128 columns of each operand: wider blocks go through in 128-column views
        dW = torch.empty((T, D, D), dtype=torch.float32, device=h.device)
        for k0 in range(0, D, 128):
            for n0 in range(0, D, 128):
                dW[:, k0:k0 + 128, n0:n0 + 128] = ops.xty([h[:, k0:k0 + 128]], dHc[:, n0:n0 + 128], x_rows=comp.pair_node,
                                                          row_off=comp.type_row_off)
        if sink is not None:
            _on_side_stream([dW], lambda: _SINK.add(sink[0], sink[1], dW))
            return None
        return dW
    if sink is not None:
        def edge_weight_products():
            ops.xty([h], dHc, x_rows=comp.pair_node, row_off=comp.type_row_off, add_to=sink[1])
            _SINK.mark(sink[0])
        _on_side_stream([h, dHc], edge_weight_products)
        return None
    return ops.xty([h], dHc, x_rows=comp.pair_node, row_off=comp.type_row_off)                 # [T, D, D]


def _gru_backward_unfused(lib, g, h, r, u, c, Wg, Wc, nin, xs, nx, T, act, use_avg, st):
    """Steps 1-5 as separate launches (hidden sizes without the fused backward kernel)."""
    V, D = h.shape
    dev = h.device
    # ---- 1. GRU blend and candidate:  h' = u*h + (1-u)*c,  c = act([x | r*h] Wc + bc)
    dpc = torch.empty_like(h)
    dpg = torch.empty((V, 2 * D), dtype=torch.float32, device=dev)         # [d pre-r | d pre-u]
    dh = torch.empty_like(h)
    rh = torch.empty_like(h)
    ops._launch("gru_bwd_stage1", lambda: lib.ggnn_gru_bwd_stage1_f32(
        g.data_ptr(), h.data_ptr(), r.data_ptr(), u.data_ptr(), c.data_ptr(), act, dpc.data_ptr(), dpg.data_ptr(),
        dh.data_ptr(), rh.data_ptr(), D, 0, V, D, st))
    # ---- 2. candidate weights (+ bias: the ones row of the same product)
    Kx = (nx + 1) * D
    dWc, dbc = weight_grad(xs + [rh], dpc)
    # ---- 3. dpc Wc^T; gates pre-activation gradients ([r|u] = sigmoid([x | h] Wg + bg))
    dx = torch.empty((V, nx * D), dtype=torch.float32, device=dev)
    ops._launch("gru_bwd_dx_cand[nx=%d]" % nx, lambda: lib.ggnn_gru_bwd_dx_cand_f32(
        dpc.data_ptr(), _TRANSPOSED.get(Wc).data_ptr(), h.data_ptr(), r.data_ptr(), dx.data_ptr(), dh.data_ptr(), dpg.data_ptr(),
        nx, V, D, st))
    # ---- 4. gate weights
    dWg, dbg = weight_grad(xs + [h], dpg)
    # ---- 5. dpg Wg^T; mean aggregation (chem_tensorflow_sparse.py:206-209)
    dinc = torch.empty_like(h)
    ops._launch("gru_bwd_dx_gates[nx=%d]" % nx, lambda: lib.ggnn_gru_bwd_dx_gates_f32(
        dpg.data_ptr(), _TRANSPOSED.get(Wg).data_ptr(), dx.data_ptr(), dinc.data_ptr(), nin.data_ptr(), T, 1 if use_avg else 0,
        dh.data_ptr(), nx, V, D, st))
    d_res = [dx[:, i * D:(i + 1) * D] for i in range(nx - 1)]
    return dpc, dpg, dh, dinc, d_res, dWc, dbc, dWg, dbg


def _backward_dense_form(ctx, g):
    """Hidden sizes without a compacted transform (or GGNN_TRAIN_COMPACT=0): the dense [V, T*D] message transform; its
    weight gradients are plain tall-skinny products on ggnn_gemm_tn_f32, the bias gradients column sums (ggnn_colsum_f32)."""
    lib = _lib.load()
    h, nin, W, Wg, Wc, incoming, r, u, c, *residuals = ctx.saved_tensors
    V, D = h.shape
    T = W.shape[0]
    nx = len(residuals) + 1
    K = (nx + 1) * D
    g = g.contiguous()
    st = torch.cuda.current_stream().cuda_stream
    act = ops.ACT_IDS[ctx.activation]

    a_c = torch.empty((V, K), dtype=torch.float32, device=h.device)        # [x_0 | .. | incoming | r*h]
    for i, x in enumerate(list(residuals) + [incoming]):
        a_c[:, i * D:(i + 1) * D] = x
    dpc = torch.empty_like(h)
    dpg = torch.empty((V, 2 * D), dtype=torch.float32, device=h.device)    # [d pre-r | d pre-u]
    dh = torch.empty_like(h)
    check(lib.ggnn_gru_bwd_stage1_f32(g.data_ptr(), h.data_ptr(), r.data_ptr(), u.data_ptr(), c.data_ptr(), act,
                                      dpc.data_ptr(), dpg.data_ptr(), dh.data_ptr(), a_c.data_ptr(), K, nx * D, V, D, st))
    dWc = _tn(a_c, dpc)
    dbc = ops.colsum(dpc)
    dxrh = ops.gemm([dpc], _TRANSPOSED.get(Wc))                            # [V, (nx+1)D] = dpc Wc^T
    drh = dxrh[:, nx * D:]
    check(lib.ggnn_gru_bwd_stage2_f32(drh.data_ptr(), K, h.data_ptr(), r.data_ptr(), dh.data_ptr(), dpg.data_ptr(), V, D, st))
    a_c[:, nx * D:] = h                                                    # reuse the buffer as [x | h]
    dWg = _tn(a_c, dpg)
    dbg = ops.colsum(dpg)
    dxh = ops.gemm([dpg[:, :D], dpg[:, D:]], _TRANSPOSED.get(Wg))          # [V, (nx+1)D] = dpg Wg^T
    dh += dxh[:, nx * D:]
    dx = dxrh[:, :nx * D] + dxh[:, :nx * D]
    d_res = [dx[:, i * D:(i + 1) * D] for i in range(nx - 1)]
    dinc = dx[:, (nx - 1) * D:]
    if ctx.use_avg:
        dinc = dinc / (nin.sum(dim=-1, keepdim=True) + SMALL_NUMBER)
    dinc = dinc.contiguous()
    dbias = ops.gemm_tn(nin, dinc) if ctx.has_bias else None              # nin^T dinc   [T, D]
    dH = ops.segment_sum_rows_by_index(dinc, _source_index(ctx.index, V)).view(V, T * D)
    WT = W.transpose(1, 2).reshape(T * D, D).contiguous()                  # rows t*D..: W_t^T
    for t0 in range(0, T, 4):                                              # (the GEMM takes <= 4 K segments)
        t1 = min(t0 + 4, T)
        dh += ops.gemm([dH[:, t * D:(t + 1) * D] for t in range(t0, t1)], WT[t0 * D:t1 * D])
    dW = _tn(h, dH).view(D, T, D).transpose(0, 1)
    if ctx.ew_mask is not None:
        dW = ops.dropout(dW.contiguous(), float(ctx.ew_mask[0]), int(ctx.ew_mask[1]))
    return (dh, None, None, dW, dbias, None, None, dWg, dbg, dWc, dbc, None, None, *d_res)

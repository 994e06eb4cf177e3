"""Which operand format the fused GRU forward multiplies in -- decided PER LAUNCH, from bounds that are PROVEN on the host.

The reference computes the GRUCell of chem_tensorflow_sparse.py:215-216 in plain f32 on arbitrary inputs (:65-73; ReLU cell
:109-110; sum aggregation when :206 is off).  libggnn_hip.so has two operand formats for that product under its split matrix path
(include/ggnn_hip.h "Operand formats", csrc/ggnn_split.hpp):

  BF16X3  the exact three-piece bf16 split, six MFMA products per f32 product: f32 semantics on every input;
  F16X2   two f16 pieces, three products: half the MFMAs -- f32-faithful (its error against f64 is below the f32 FMA chain's) only
          INSIDE its operand range: every GRU weight |w| <= 255.875 (packed x 2^8 into f16), every activation |a| <= 65504.

F16X2 is selected for a layer only when both bounds follow from quantities the host KNOWS -- never on trust, never checked after the
fact.  What is known, and what follows from it:

  * max |h0| of the batch (the packers of this package know it from the dataset's node features; a foreign feed is measured once per
    tensor by ggnn_absmax_f32, one 8 us launch + one read-back), max |W| of every weight tensor (measured once per weight version;
    during training re-measured every REMEASURE_STEPS optimizer steps and in between bounded by Adam's step bound, below);
  * tanh cell:  h' = u*h + (1-u)*c with u in [0,1], |c| <= 1 is a convex combination: |h'| <= max(|h|, 1), so every state of every
    layer is bounded by S = max(1, max|h0|) (state dropout divides by keep: S / keep per application);  ReLU cell: no bound -> BF16X3;
  * mean aggregation (:206-209):  incoming[v] = (sum of messages + sum_t nin[v,t] b_t) / (deg + 1e-7), |message_j| = |sum_i h_i W_t[i,j]|
    <= D max|W_edge| S, so |incoming| <= D max|W_edge| S / keep_w + max|b_edge| (keep_w: the weight-dropout mask divides by it);
    sum aggregation: the bound needs the largest in-degree -> BF16X3 (a hub graph is exactly where the states grow);
  * the GRU's activations are its x segments (residual states, incoming), h and r*h (|r*h| <= |h|): bounded by max(S, incoming bound).

A NaN or Inf anywhere (h0 or a weight) makes its maximum non-finite, every `<=` below false, and the layer runs in BF16X3, where
non-finite values stay non-finite.  Everything here is host arithmetic on a handful of floats; the only device work is the maxima.

GGNN_GRU_FMT (process default of this policy, also ggnn_gru_forward_format()): unset / "auto" / "2" = as above; "3" / "exact" =
BF16X3 always; "force2" = F16X2 without the proof (kernel experiments and the operand-range test only -- outside the range the f16
pieces overflow and the results are non-finite: nothing is clamped).  `with formats.forced(fmt):` overrides it for a block (the parity suite runs under both formats).
"""
from __future__ import annotations

import contextlib
import ctypes
import functools
import math
import os
import threading
import weakref
from typing import Any, Dict, List, Optional, Sequence

import torch

from . import _lib
from ._lib import check

F16X2, BF16X3 = 2, 3
NAMES = {F16X2: "f16x2", BF16X3: "bf16x3", 0: "f32"}
MAX_WEIGHT = 255.875          # GGNN_F16X2_MAX_WEIGHT: 65504 / 2^8
MAX_ACTIVATION = 65504.0      # GGNN_F16X2_MAX_ACTIVATION
_MARGIN = 1.0 + 1e-3          # the bounds are real-number arguments; fp32 rounding of the bounded quantities stays far inside this
REMEASURE_STEPS = 64          # training: optimizer steps between two measurements of the weights' maxima

_local = threading.local()


def _env_policy() -> str:
    e = (os.environ.get("GGNN_GRU_FMT") or "auto").strip().lower()
    if e in ("3", "exact", "bf16x3"):
        return "exact"
    if e in ("force2", "f16x2!", "unchecked"):
        return "force2"
    return "auto"


def policy() -> str:
    """'auto' (F16X2 where proven safe), 'exact' (BF16X3 always) or 'force2' (F16X2 unchecked; experiments only)."""
    return getattr(_local, "policy", None) or _env_policy()


@contextlib.contextmanager
def forced(fmt):
    """Run a block under a fixed policy: BF16X3 / 'exact', F16X2 / 'force2' (UNCHECKED two-piece format), or 'auto'."""
    name = {BF16X3: "exact", F16X2: "force2"}.get(fmt, fmt)
    if name not in ("auto", "exact", "force2"):
        raise ValueError("unknown format policy %r" % (fmt,))
    prev = getattr(_local, "policy", None)
    _local.policy = name
    try:
        yield
    finally:
        _local.policy = prev


def split_path() -> bool:
    return bool(_lib.load().ggnn_matrix_path_is_split())


# ---- maxima ---------------------------------------------------------------------------------------------------------------------
def absmax(tensors: Sequence[torch.Tensor]) -> List[float]:
    """[max |x|] of each tensor (ggnn_absmax_f32: one launch, one read-back = one stream synchronisation).  NaN if the tensor holds
    a NaN, inf if it holds an Inf; 0.0 for an empty tensor.  CPU tensors are measured with torch (host-side tests)."""
    if not tensors:
        return []
    if not all(t.is_cuda for t in tensors):
        out = []
        for t in tensors:
            t = t.detach().float()
            out.append(0.0 if t.numel() == 0 else (float("nan") if bool(torch.isnan(t).any()) else float(t.abs().max())))
        return out
    ts = [t.detach() if t.is_contiguous() else t.detach().contiguous() for t in tensors]
    for t in ts:
        if t.dtype != torch.float32:
            raise TypeError("absmax takes float32 tensors")
    n = len(ts)
    out = torch.empty(n, dtype=torch.float32, device=ts[0].device)
    ptrs = (ctypes.c_void_p * n)(*[t.data_ptr() if t.numel() else None for t in ts])
    numel = (ctypes.c_int64 * n)(*[t.numel() for t in ts])
    check(_lib.load().ggnn_absmax_f32(ptrs, numel, n, out.data_ptr(), torch.cuda.current_stream().cuda_stream))
    return [float(v) for v in out.tolist()]


class _MaxCache:
    """max|x| per tensor, valid while the same tensor object sits at the same version (the key of ops.PackedWeights)."""

    def __init__(self):
        self._t: Dict[Any, Any] = {}

    @staticmethod
    def _base(t):
        return t._base if t._base is not None else t

    def get(self, tensors: Sequence[torch.Tensor]) -> List[float]:
        out: List[Optional[float]] = []
        miss = []
        for i, t in enumerate(tensors):
            key = (id(self._base(t)), t.data_ptr(), tuple(t.shape))
            hit = self._t.get(key)
            if hit is not None and hit[0]() is self._base(t) and hit[1] == t._version:
                out.append(hit[2])
            else:
                out.append(None)
                miss.append((i, key, t))
        if miss:
            vals = absmax([t for _, _, t in miss])
            if len(self._t) > 4096:
                self._t.clear()
            for (i, key, t), v in zip(miss, vals):
                self._t[key] = (weakref.ref(self._base(t)), t._version, v)
                out[i] = v
        return out  # type: ignore[return-value]


_WEIGHT_MAX = _MaxCache()
_H0_MAX = _MaxCache()


def weight_absmax(tensors: Sequence[torch.Tensor]) -> List[float]:
    """Cached per weight version (one measurement per tensor and version; every in-place update bumps the version)."""
    return _WEIGHT_MAX.get(tensors)


def declare_h0_absmax(feed: Dict[str, Any], value: float) -> Dict[str, Any]:
    """A packer's statement "max |h0| of THIS tensor is `value`" (known from the dataset's node features, or from how the producer
    drew the states): stored next to the tensor it is about, with the tensor's identity and version, so that a feed whose h0 was
    replaced or overwritten afterwards is measured instead of trusted."""
    h0 = feed['initial_node_representation']
    feed['h0_absmax'] = float(value)
    feed['_h0_absmax_of'] = (weakref.ref(h0), h0._version)
    # ... and, with it, the packer's second statement: the in-degree table of this feed was counted from the very adjacency lists its
    # message index was built from (nin_consistent below), tied to the table's identity and version the same way
    nin = feed.get('num_incoming_edges_per_type')
    if isinstance(nin, torch.Tensor):
        feed['_nin_consistent_of'] = (weakref.ref(nin), nin._version)
    return feed


_NIN_OK: Dict[Any, Any] = {}


def nin_consistent(placeholders: Dict[str, Any], row_ptr: Optional[torch.Tensor]) -> bool:
    """Mean aggregation divides by sum_t nin[v, t] + 1e-7 (chem_tensorflow_sparse.py:206-209), and the bound on the aggregated messages
    (incoming_bound) holds only if that divisor is at least the number of messages summed into v.  `num_incoming_edges_per_type` is a
    fed placeholder: a foreign feed may hold anything (zeros, fractions), and sum / 1e-7 then leaves the two-piece f16 format's range
    where the reference's f32 stays finite.  True iff the packer declared the table (declare_h0_absmax) and it is still that tensor,
    or -- measured once per (table, version, index) with one read-back -- every entry is finite and >= 0 and every node's row sum
    reaches its in-degree in the message index."""
    nin = placeholders.get('num_incoming_edges_per_type')
    if not isinstance(nin, torch.Tensor):
        return False
    of = placeholders.get('_nin_consistent_of')
    if of is not None and of[0]() is nin and of[1] == nin._version:
        return True
    if row_ptr is None or nin.dim() != 2 or row_ptr.numel() != nin.shape[0] + 1:
        return False
    key = (id(nin), nin._version, id(row_ptr), row_ptr._version)
    hit = _NIN_OK.get(key)
    if hit is not None and hit[0]() is nin and hit[1]() is row_ptr:
        return hit[2]
    deg = (row_ptr[1:] - row_ptr[:-1]).to(torch.float32)
    ok = bool((torch.isfinite(nin).all() & (nin >= 0).all() & ((nin.sum(1) + 1e-3) >= deg).all()).item()) if nin.shape[0] else True
    if len(_NIN_OK) > 1024:
        _NIN_OK.clear()
    _NIN_OK[key] = (weakref.ref(nin), weakref.ref(row_ptr), ok)
    return ok


_ADJ_MAX = _MaxCache()


def adjacency_absmax(A: torch.Tensor) -> float:
    """max |A| of a dense model's fed adjacency tensor (cached per tensor and version): the bound on the aggregated activations
    assumes entries of magnitude <= 1 -- the reference's 0 / 1 matrices; a foreign feed may weight its edges."""
    if A.numel() == 0:
        return 0.0
    if A.dtype != torch.float32:
        return float(A.abs().max())                   # (not the reference's feed dtype: measured outright)
    return float(_ADJ_MAX.get([A])[0])


def h0_absmax(placeholders: Dict[str, Any], h0: Optional[torch.Tensor] = None) -> float:
    """max |h0| of the fed batch: the packers' declaration (declare_h0_absmax) while it is about the tensor that is fed, else
    measured once per h0 tensor and version (ggnn_absmax_f32 + one read-back)."""
    if h0 is None:
        h0 = placeholders['initial_node_representation']
    v, of = placeholders.get('h0_absmax'), placeholders.get('_h0_absmax_of')
    if v is not None and of is not None and of[0]() is h0 and of[1] == h0._version:
        return float(v)
    v = _H0_MAX.get([h0])[0]                      # (per tensor object and version: a loop over a few resident feeds measures each once)
    placeholders['h0_absmax'] = v
    placeholders['_h0_absmax_of'] = (weakref.ref(h0), h0._version)
    return float(v)


# ---- the bounds -------------------------------------------------------------------------------------------------------------------
def state_bound(h0_max: float, activation: str, applications: int = 0, state_keep: float = 1.0) -> float:
    """Bound on |state| of every layer and timestep (tanh cell), inf where there is none.  applications: number of timesteps over
    which the state-dropout division by state_keep compounds."""
    if activation.lower() != 'tanh' or not (h0_max == h0_max):
        return math.inf
    s = max(1.0, float(h0_max))
    if state_keep < 1.0:
        if state_keep <= 0.0:
            return math.inf
        s = s / (state_keep ** max(int(applications), 0))
    return s


def incoming_bound(state_b: float, D: int, edge_w_max: float, edge_b_max: float, use_avg: bool, weight_keep: float = 1.0) -> float:
    """Bound on |aggregated messages| (the GRU's last input segment) under mean aggregation; inf for sum aggregation."""
    if not use_avg or weight_keep <= 0.0:
        return math.inf
    return D * (edge_w_max / min(weight_keep, 1.0)) * state_b + edge_b_max


def layer_format(state_b: float, inc_b: float, gru_w_max: float) -> int:
    """F16X2 iff every operand of the layer's GRU launches is provably inside the format's range."""
    pol = policy()
    if pol == "exact":
        return BF16X3
    if pol == "force2":
        return F16X2
    # (each bound tested on its own: a NaN fails its test, whereas Python's max() would drop it)
    ok = gru_w_max * _MARGIN <= MAX_WEIGHT and state_b * _MARGIN <= MAX_ACTIVATION and inc_b * _MARGIN <= MAX_ACTIVATION
    return F16X2 if ok else BF16X3


def nanmax(*values: float) -> float:
    """max() that keeps a NaN (Python's max drops or keeps it depending on the argument order)."""
    out = -math.inf
    for v in values:
        if v != v:
            return float("nan")
        out = max(out, v)
    return out


@functools.lru_cache(maxsize=64)
def adam_step_bound(lr: float, b1: float, b2: float) -> float:
    """|delta w| of ONE step of tf.train.AdamOptimizer (train.TFAdam) is at most lr * C:  with m = (1-b1) sum b1^k g, v = (1-b2) sum
    b2^k g^2, Cauchy-Schwarz gives m^2 <= (1-b1)^2 / ((1-b2)(1 - b1^2/b2)) v for b1^2 < b2, and the bias-correction factor
    sqrt(1-b2^t)/(1-b1^t) stays <= 1 / (1-b1) ... bounded here by its supremum over t (computed, not assumed).  inf when b1^2 >= b2."""
    if not (0.0 <= b1 < 1.0 and 0.0 < b2 < 1.0) or b1 * b1 >= b2:
        return math.inf
    c = (1.0 - b1) / math.sqrt((1.0 - b2) * (1.0 - b1 * b1 / b2))
    corr = max(math.sqrt(1.0 - b2 ** t) / (1.0 - b1 ** t) for t in list(range(1, 2000)) + [10 ** 9])
    return lr * c * max(corr, 1.0)


class TrainingWeightBounds:
    """max|W| of a model's weight tensors during training without a read-back per step: measured every REMEASURE_STEPS optimizer
    steps (one synchronisation), in between  max|W|_now <= max|W|_measured + steps_since * adam_step_bound  -- valid as long as the
    optimizer is the only writer, which the tensors' version counters prove (every optimizer step bumps each by exactly one,
    train.TFAdam.clip_and_apply; anything else -- a restored checkpoint, set_graph_weights -- breaks the count and forces a new
    measurement)."""

    def __init__(self):
        self._at = None          # (optimizer step, [versions], [maxima])

    def get(self, tensors: Sequence[torch.Tensor], optimizer) -> List[float]:
        t_now = int(getattr(optimizer, "t", 0))
        versions = [t._version for t in tensors]
        step_b = adam_step_bound(float(optimizer.lr), float(optimizer.b1), float(optimizer.b2)) if optimizer is not None else math.inf
        if self._at is not None:
            t0, v0, m0 = self._at
            k = t_now - t0
            if (len(v0) == len(versions) and 0 <= k < REMEASURE_STEPS and math.isfinite(step_b)
                    and all(v - w == k for v, w in zip(versions, v0))):
                return [m + k * step_b for m in m0]
        m = absmax(list(tensors))
        self._at = (t_now, versions, m)
        return m

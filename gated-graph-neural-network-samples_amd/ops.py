"""Operator layer: torch-ROCm tensors in, libggnn_hip.so HIP kernels underneath (via ctypes).

One function per TF op call site of the reference's sparse hot path
(chem_tensorflow_sparse.py:117-218):

    build_message_index   S1 (+ our bucketing by target)     :120-129
    msg_transform         S3 (transform-first)                :160-164
    gather_segment_sum    S2,S4,S5,S6,S7                      :160-162,168,198-209
    gru                   S8,S9                               :211-216
    unsorted_segment_sum  general tf.unsorted_segment_sum     :198-200, :226-228

PyTorch is plumbing only (device memory + streams).  Every function requires CUDA(HIP) tensors and
raises if the extension is missing -- there is no CPU path here.
"""
from __future__ import annotations

import ctypes
import os
from dataclasses import dataclass
from typing import List, Optional, Sequence

import torch

from . import _lib
from ._lib import check

ACT_IDS = {"tanh": 0, "relu": 1}
GRU_FUSED_MAX_INPUTS = 3        # input segments (residual inputs + aggregated messages) of the single-launch GRU kernels
GRU_MAX_INPUTS = 7              # ... of the generic two-launch GRU (8 K segments of the generic GEMM: 7 inputs + h)
GRU_FMT_F16X2, GRU_FMT_EXACT = 2, 3   # operand formats of the fused GRU forward (GGNN_GRU_FMT_F16X2 / _BF16X3; policy: formats.py)


def kernel_width(D: int) -> int:
    """Width at which node states of hidden size D live in HBM.  Hidden sizes the kernels take as they are (multiples of 32, 64
    or 100) keep their width; any other size -- the reference accepts every hidden_size (chem_tensorflow_sparse.py:46-50) -- is
    zero-padded to the next width that has single-launch kernels (32, 64, 100, 128, 192, 256), beyond that to a multiple of 32.
    Zero columns stay zero through every propagation step when the weights are padded with zeros (padded rows contribute 0 to
    every product; a padded state column is u*0 + (1-u)*act(0) = 0 for tanh and ReLU), so the model is unchanged."""
    D = int(D)
    if D <= 0:
        raise ValueError("hidden_size must be positive")
    if D % 32 == 0 or D % 100 == 0:
        return D
    for w in (32, 64, 100, 128, 192, 256):
        if D <= w:
            return w
    return (D + 31) // 32 * 32
# sparse_propagate: gather the segment sum inside the fused GRU (2 launches per timestep instead of 3) for layers with at
# most FUSE_GATHER concatenated GRU inputs (3 = every layer (default: +1.8 % over 1 on MI355X), 1 = only layers without
# residual inputs, 0 = never)
FUSE_GATHER = int(os.environ.get("GGNN_FUSE_GATHER", "3"))


# ---- optional per-launch timing (bench.py's roofline leg) ---------------------------------------------
_timing = None


class kernel_timing:
    """Context manager: while active every op brackets its launch(es) with HIP events recorded on the
    stream the kernel is launched on (torch's current stream == the stream handed to the C ABI) and the
    GRU is issued as its two separately addressable launches.  `.results()` -> {name: [ms, ...]}."""

    def __enter__(self):
        global _timing
        self.records = _timing = []
        return self

    def __exit__(self, *exc):
        global _timing
        _timing = None

    def results(self):
        torch.cuda.synchronize()
        out = {}
        for name, s, e in self.records:
            out.setdefault(name, []).append(s.elapsed_time(e))
        return out


def _launch(name: str, fn):
    if _timing is None:
        return check(fn())
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    rc = fn()
    e.record()
    _timing.append((name, s, e))
    check(rc)


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def _req(t: torch.Tensor, dtype, name: str) -> torch.Tensor:
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise TypeError("%s must be a CUDA/HIP tensor (the GGNN hot path has no CPU implementation)" % name)
    if t.dtype != dtype:
        raise TypeError("%s must be %s, got %s" % (name, dtype, t.dtype))
    if not t.is_contiguous():
        raise ValueError("%s must be contiguous" % name)
    return t


def _ptr(t: Optional[torch.Tensor]):
    return None if t is None else t.data_ptr()


@dataclass
class MessageIndex:
    """Per-batch message index (built once, reused by all propagation steps and epochs).

    adj         [M,2] int32  concatenated adjacency lists (type-major)   (:124-129)
    type_off    list[T+1]    host offsets of each type in adj
    row_ptr     [V+1] int32  slots of the messages INTO node v
    gather_row  [M]   int32  slot -> src*T + type
    msg_perm    [M]   int32  slot -> original message index
    """
    adj: torch.Tensor
    type_off: List[int]
    row_ptr: torch.Tensor
    gather_row: torch.Tensor
    msg_perm: torch.Tensor
    num_nodes: int
    num_edge_types: int

    @property
    def num_messages(self) -> int:
        return int(self.type_off[-1])


def _build_index(adjacency_lists: Sequence[torch.Tensor], num_nodes: int, validate: bool, by_source: bool) -> MessageIndex:
    lib = _lib.load()
    T = len(adjacency_lists)
    if T == 0:
        raise ValueError("need at least one edge type")
    lists = [_req(a.reshape(-1, 2), torch.int32, "adjacency_lists[%d]" % i) for i, a in enumerate(adjacency_lists)]
    dev = lists[0].device
    type_off = [0]
    for a in lists:
        type_off.append(type_off[-1] + a.shape[0])
    M = type_off[-1]
    adj = torch.cat(lists, dim=0).contiguous() if M else torch.zeros((0, 2), dtype=torch.int32, device=dev)
    nseg = num_nodes * T if by_source else num_nodes
    row_ptr = torch.empty(nseg + 1, dtype=torch.int32, device=dev)
    gather_row = torch.empty(M, dtype=torch.int32, device=dev)
    msg_perm = torch.empty(M, dtype=torch.int32, device=dev)
    err = torch.zeros(1, dtype=torch.int32, device=dev)
    ws_bytes = lib.ggnn_csr_workspace_bytes(M, num_nodes)
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
    off = (ctypes.c_int64 * (T + 1))(*type_off)
    fn = lib.ggnn_build_source_csr if by_source else lib.ggnn_build_target_csr
    check(fn(_ptr(adj), off, T, num_nodes, M, _ptr(row_ptr), _ptr(gather_row), _ptr(msg_perm), _ptr(err), _ptr(ws),
             ws_bytes, _stream()))
    if validate and M and int(err.item()) != 0:
        raise IndexError("adjacency list holds a node id outside [0, %d)" % num_nodes)
    return MessageIndex(adj, type_off, row_ptr, gather_row, msg_perm, nseg, T)


def build_message_index(adjacency_lists: Sequence[torch.Tensor], num_nodes: int,
                        validate: bool = True) -> MessageIndex:
    """chem_tensorflow_sparse.py:120-129 plus the stable bucketing by target that makes the segment
    sum atomics-free.  adjacency_lists: T tensors int32 [E_t,2] (src,dst) on the GPU; E_t may be 0
    (:346-347).  With validate=True an out-of-range src/dst raises IndexError (TF-CPU raises
    InvalidArgument at the gather / segment_sum); this costs one device->host sync, once per batch."""
    return _build_index(adjacency_lists, num_nodes, validate, by_source=False)


def build_source_index(adjacency_lists: Sequence[torch.Tensor], num_nodes: int) -> MessageIndex:
    """The transpose index for the backward pass: V*T segments keyed by (src*T + type), gathering dst rows."""
    return _build_index(adjacency_lists, num_nodes, False, by_source=True)


USE_SLOT_HEADS = os.environ.get("GGNN_SLOT_HEADS", "1") != "0"


def slot_heads(owner, row_ptr: torch.Tensor, gather_row: torch.Tensor, num_segments: int) -> Optional[torch.Tensor]:
    """[num_segments, 4] int32 gather rows of the first four slots of every segment (-1 padded), built once per index and
    cached on `owner` (ggnn_build_slot_heads): the segment-sum kernels then need one dependent load level less."""
    if not USE_SLOT_HEADS or num_segments == 0:
        return None
    cached = getattr(owner, "_slot_heads", None)
    if cached is not None and cached[0] is gather_row:
        return cached[1]
    lib = _lib.load()
    heads = torch.empty((num_segments, 4), dtype=torch.int32, device=row_ptr.device)
    check(lib.ggnn_build_slot_heads(_ptr(row_ptr), _ptr(gather_row), _ptr(heads), num_segments, _stream()))
    owner._slot_heads = (gather_row, heads)
    return heads


def _segment_sum(name, Hrows, row_ptr, gather_row, heads, nin, bias, use_avg, out, V, D, T, accumulate=False):
    lib = _lib.load()
    if heads is not None:
        _launch(name, lambda: lib.ggnn_gather_segment_sum_heads_f32(
            _ptr(Hrows), _ptr(row_ptr), _ptr(gather_row), _ptr(heads), _ptr(nin), _ptr(bias), 1 if use_avg else 0, _ptr(out), V, D, T,
            1 if accumulate else 0, _stream()))
    elif accumulate:
        _launch(name, lambda: lib.ggnn_gather_segment_sum_acc_f32(_ptr(Hrows), _ptr(row_ptr), _ptr(gather_row), _ptr(out), V, D, _stream()))
    else:
        _launch(name, lambda: lib.ggnn_gather_segment_sum_f32(_ptr(Hrows), _ptr(row_ptr), _ptr(gather_row), _ptr(nin), _ptr(bias),
                                                              1 if use_avg else 0, _ptr(out), V, D, T, _stream()))
    return out


def msg_transform(h: torch.Tensor, edge_weights: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """H[v, t*D:(t+1)*D] = h[v] @ edge_weights[t]   (chem_tensorflow_sparse.py:160-164, all types
    in one FP32-MFMA GEMM).  h [V,D], edge_weights [T,D,D] -> H [V, T*D]."""
    lib = _lib.load()
    _req(h, torch.float32, "h"); _req(edge_weights, torch.float32, "edge_weights")
    V, D = h.shape
    T = edge_weights.shape[0]
    if edge_weights.shape != (T, D, D):
        raise ValueError("edge_weights must be [T,D,D]")
    if out is None:
        out = torch.empty((V, T * D), dtype=torch.float32, device=h.device)
    else:
        _req(out, torch.float32, "out")
        if out.shape != (V, T * D):
            raise ValueError("out must be [V, T*D]")
    _launch("msg_transform", lambda: lib.ggnn_msg_transform_f32(_ptr(h), D, _ptr(edge_weights), _ptr(out), V, D, T, _stream()))
    return out


def gather_segment_sum(H: torch.Tensor, index: MessageIndex, num_incoming_edges_per_type: Optional[torch.Tensor],
                       edge_biases: Optional[torch.Tensor], use_avg: bool,
                       out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """incoming[v] = (sum of H rows of the messages into v [+ nin[v] @ edge_biases]) [/ (sum_t nin[v,t] + 1e-7)]
    (chem_tensorflow_sparse.py:160-162,168,198-209).  H [V, T*D] from msg_transform."""
    lib = _lib.load()
    _req(H, torch.float32, "H")
    V, T = index.num_nodes, index.num_edge_types
    if H.shape[0] != V or H.shape[1] % T:
        raise ValueError("H must be [V, T*D]")
    D = H.shape[1] // T
    nin = num_incoming_edges_per_type
    if nin is not None:
        _req(nin, torch.float32, "num_incoming_edges_per_type")
        if nin.shape != (V, T):
            raise ValueError("num_incoming_edges_per_type must be [V,T]")
    if edge_biases is not None:
        _req(edge_biases, torch.float32, "edge_biases")
        if edge_biases.shape != (T, D):
            raise ValueError("edge_biases must be [T,D]")
    if out is None:
        out = torch.empty((V, D), dtype=torch.float32, device=H.device)
    else:
        _req(out, torch.float32, "out")
    return _segment_sum("gather_segment_sum", H, index.row_ptr, index.gather_row, slot_heads(index, index.row_ptr, index.gather_row, V),
                        nin, edge_biases, use_avg, out, V, D, T)


def gru_workspace(V: int, D: int, device) -> torch.Tensor:
    lib = _lib.load()
    return torch.empty(lib.ggnn_gru_workspace_bytes(V, D) // 4, dtype=torch.float32, device=device)


def gru(x_segs: Sequence[torch.Tensor], h: torch.Tensor, Wg: torch.Tensor, bg: torch.Tensor, Wc: torch.Tensor,
        bc: torch.Tensor, activation: str = "tanh", out: Optional[torch.Tensor] = None,
        ws: Optional[torch.Tensor] = None, save: Optional[dict] = None, two_launch: bool = False,
        fmt: Optional[int] = None) -> torch.Tensor:
    """TF-1.3 GRUCell on x = concat(x_segs) without materialising the concat
    (chem_tensorflow_sparse.py:211-216).  save: optional dict receiving 'r','u','c' [V,D] tensors.
    two_launch=True forces the un-fused gates + candidate kernels (the path large D takes).
    On raw weights the products run in the exact BF16X3 operand format (ggnn_gru_f32); fmt = formats.F16X2 packs the weights
    in the two-piece f16 format and runs the fused launch in it (the CALLER vouches for the operand range, see formats.py)."""
    lib = _lib.load()
    if fmt is not None and int(fmt) != GRU_FMT_EXACT and not two_launch and lib.ggnn_gru_is_fused(h.shape[1]) \
            and len(x_segs) <= GRU_FUSED_MAX_INPUTS:
        _req(Wg, torch.float32, "Wg"); _req(Wc, torch.float32, "Wc")
        # (the images are cached per weight version and format: this branch runs once per timestep on the non-gather training path)
        packed = _gru_images().gru(Wg, Wc, len(x_segs), h.shape[1], int(fmt))
        return gru_packed(x_segs, h, packed, bg, bc, activation, out=out, fmt=int(fmt), save=save)
    _req(h, torch.float32, "h")
    V, D = h.shape
    nx = len(x_segs)
    for i, x in enumerate(x_segs):
        _req(x, torch.float32, "x_segs[%d]" % i)
        if x.shape != (V, D):
            raise ValueError("x_segs[%d] must be [V,D]" % i)
    for n, w, shp in (("Wg", Wg, ((nx + 1) * D, 2 * D)), ("bg", bg, (2 * D,)), ("Wc", Wc, ((nx + 1) * D, D)), ("bc", bc, (D,))):
        _req(w, torch.float32, n)
        if tuple(w.shape) != shp:
            raise ValueError("%s must have shape %s, got %s" % (n, shp, tuple(w.shape)))
    act = ACT_IDS.get(activation.lower())
    if act is None:
        raise Exception("Unknown activation function type '%s'." % activation)
    if out is None:
        out = torch.empty_like(h)
    if ws is None:
        ws = gru_workspace(V, D, h.device)
    sr = su = sc = None
    if save is not None:
        sr = save["r"] = torch.empty_like(h)
        su = save["u"] = torch.empty_like(h)
        sc = save["c"] = torch.empty_like(h)
    segs = (ctypes.c_void_p * nx)(*[x.data_ptr() for x in x_segs])
    if not two_launch and (_timing is None or lib.ggnn_gru_is_fused(D)):
        _launch("gru_fused[nx=%d]" % nx, lambda: lib.ggnn_gru_f32(
            segs, nx, _ptr(h), _ptr(Wg), _ptr(bg), _ptr(Wc), _ptr(bc), _ptr(out), _ptr(ws), ws.numel() * 4,
            _ptr(sr), _ptr(su), _ptr(sc), V, D, act, _stream()))
    else:   # un-fused sizes: the same two launches, individually bracketed by events
        if ws.numel() < 2 * V * D:
            raise ValueError("GRU workspace too small")
        rh = ws[:V * D]
        u = su if su is not None else ws[V * D:2 * V * D]
        _launch("gru_gates[nx=%d]" % nx, lambda: lib.ggnn_gru_gates_f32(segs, nx, _ptr(h), _ptr(Wg), _ptr(bg), _ptr(rh),
                                                                         _ptr(u), _ptr(sr), V, D, _stream()))
        _launch("gru_candidate[nx=%d]" % nx, lambda: lib.ggnn_gru_candidate_f32(
            segs, nx, _ptr(rh), _ptr(h), _ptr(u), _ptr(Wc), _ptr(bc), _ptr(out), _ptr(sc), V, D, act, _stream()))
    return out


def gemm(a_segs: Sequence[torch.Tensor], B: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """C = concat(a_segs, dim=1) @ B on the FP32-MFMA kernel (no concat materialised).  Segments are
    [M,D] tensors (contiguous, or equal-stride column slices of one wider row-major matrix)."""
    lib = _lib.load()
    M, D = a_segs[0].shape
    lda = a_segs[0].stride(0) if M > 1 else D
    for i, a in enumerate(a_segs):
        if not a.is_cuda or a.dtype != torch.float32:
            raise TypeError("a_segs[%d] must be a float32 CUDA/HIP tensor" % i)
        if a.shape != (M, D) or a.stride(1) != 1 or (M > 1 and a.stride(0) != lda):
            raise ValueError("all segments must be [M,D] row-major with one common row stride")
    _req(B, torch.float32, "B")
    K, N = B.shape
    if K != len(a_segs) * D:
        raise ValueError("B must be [nseg*D, N]")
    if out is None:
        out = torch.empty((M, N), dtype=torch.float32, device=B.device)
    segs = (ctypes.c_void_p * len(a_segs))(*[a.data_ptr() for a in a_segs])
    _launch("gemm[K=%d,N=%d]" % (K, N), lambda: lib.ggnn_gemm_f32(segs, len(a_segs), D, lda, _ptr(B), N, _ptr(out), N, M, N, _stream()))
    return out


def gemm_tn(x: torch.Tensor, dy: torch.Tensor) -> torch.Tensor:
    """x^T @ dy for tall operands ([M,K]^T [M,N] -> [K,N], M ~ 1e5): the weight-gradient product of the backward
    pass on ggnn_gemm_tn_f32 (rows split over the whole GPU, deterministic reduction).  x, dy: row-major (any row
    stride), dy with a row stride and N that are multiples of 4."""
    lib = _lib.load()
    for name, t in (("x", x), ("dy", dy)):
        if not t.is_cuda or t.dtype != torch.float32 or t.dim() != 2 or (t.shape[0] > 0 and t.stride(1) != 1):
            raise TypeError("%s must be a 2-D float32 CUDA/HIP tensor with unit column stride" % name)
    M, K = x.shape
    N = dy.shape[1]
    if dy.shape[0] != M:
        raise ValueError("x and dy must have the same number of rows")
    if M == 0:
        return torch.zeros((K, N), dtype=torch.float32, device=x.device)
    if M > 1 and (dy.stride(0) % 4 != 0 or dy.data_ptr() % 16 != 0):
        # the kernel reads dy in float4s: a row stride that is no multiple of 4 floats (a contiguous [M, 6] or [M, 513]) or a
        # misaligned view goes through ONE copy whose rows are padded to a multiple of 4 (zero columns add nothing to the product)
        n_pad = (N + 3) // 4 * 4
        buf = torch.zeros((M, n_pad), dtype=torch.float32, device=dy.device)
        buf[:, :N] = dy
        return gemm_tn(x, buf)[:, :N].contiguous() if n_pad != N else gemm_tn(x, buf)
    if N > 512 or N % 4 != 0:
        # the kernel takes N <= 512, N % 4 == 0: wider products go through in 512-column views (it takes any row stride), a ragged
        # tail (N % 4 columns: readout widths like 1 or 2) through a zero-padded copy of those columns
        n4 = N // 4 * 4
        parts = [gemm_tn(x, dy[:, c:min(c + 512, n4)]) for c in range(0, n4, 512)]
        if n4 < N:
            tail = torch.zeros((M, 4), dtype=torch.float32, device=dy.device)
            tail[:, :N - n4] = dy[:, n4:]
            parts.append(gemm_tn(x, tail)[:, :N - n4])
        return torch.cat(parts, dim=1)
    lda = x.stride(0) if M > 1 else K
    ldb = dy.stride(0) if M > 1 else N
    out = torch.empty((K, N), dtype=torch.float32, device=x.device)
    ws_bytes = lib.ggnn_gemm_tn_workspace_bytes(M, K, N)
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=x.device)
    _launch("gemm_tn[K=%d,N=%d]" % (K, N), lambda: lib.ggnn_gemm_tn_f32(_ptr(x), lda, _ptr(dy), ldb, _ptr(out), M, K, N,
                                                                       _ptr(ws), ws_bytes, _stream()))
    return out


def xty(x_segs: Sequence[torch.Tensor], dy: torch.Tensor, x_rows: Optional[torch.Tensor] = None,
        row_off: Optional[Sequence[int]] = None, ones_row: bool = False, add_to: Optional[torch.Tensor] = None,
        add_bias_to: Optional[torch.Tensor] = None) -> Optional[torch.Tensor]:
    """Weight-gradient product  concat(x_segs, dim=1)^T @ dy  on ggnn_xty_f32 (no concat materialised; deterministic).
    x_segs: [M', Dseg] float32 tensors with unit column stride (any row stride: column slices are fine); dy [M, N], N <= 208.
    x_rows (int32 [M]): row r of the X operand is x_segs[.][x_rows[r]] (edge-weight gradients on compact rows).
    row_off (host ints [B+1]): B independent products over the row ranges -> [B, K, N]; default one product -> [K, N].
    ones_row: the result has K + 1 rows, the last one the column sums of dy (the bias gradient next to the weight gradient).
    add_to (contiguous float32, B*K*N elements) [, add_bias_to (N or B*N elements; needs ones_row)]: the product (and the column
    sums) are ADDED into these buffers by the reduction kernel itself and nothing is returned (gradient accumulation)."""
    lib = _lib.load()
    nseg = len(x_segs)
    Dseg = x_segs[0].shape[1]
    if nseg > 4:
        # the kernel takes 4 column segments of X: more (layers with more than 2 residual inputs) are row blocks of the product
        if row_off is not None or x_rows is not None:
            raise ValueError("batched / row-gathered products take at most 4 segments")
        N = dy.shape[1]
        groups = [list(x_segs[i:i + 4]) for i in range(0, nseg, 4)]
        if add_to is not None:
            dst = add_to.view(nseg * Dseg, N)
            k0 = 0
            for gi, grp in enumerate(groups):
                last = gi == len(groups) - 1
                xty(grp, dy, ones_row=ones_row and last, add_to=dst[k0:k0 + len(grp) * Dseg], add_bias_to=add_bias_to if last else None)
                k0 += len(grp) * Dseg
            return None
        parts = [xty(grp, dy, ones_row=ones_row and gi == len(groups) - 1) for gi, grp in enumerate(groups)]
        return torch.cat(parts, dim=0)
    for i, x in enumerate(x_segs):
        if not x.is_cuda or x.dtype != torch.float32 or x.dim() != 2 or x.shape[1] != Dseg or (x.shape[0] > 1 and x.stride(1) != 1):
            raise TypeError("x_segs[%d] must be a [M, %d] float32 CUDA/HIP tensor with unit column stride" % (i, Dseg))
    if not dy.is_cuda or dy.dtype != torch.float32 or dy.dim() != 2 or (dy.shape[0] > 1 and dy.stride(1) != 1):
        raise TypeError("dy must be a 2-D float32 CUDA/HIP tensor with unit column stride")
    M, N = dy.shape
    K = nseg * Dseg
    batched = row_off is not None
    offs = [0, M] if row_off is None else [int(o) for o in row_off]
    B = len(offs) - 1
    Kout = K + 1 if ones_row else K
    if add_to is not None:
        if ones_row != (add_bias_to is not None):
            raise ValueError("add_to with ones_row needs add_bias_to (and vice versa)")
        for t, n in ((add_to, B * K * N), (add_bias_to, B * N)):
            if t is not None and (not t.is_cuda or t.dtype != torch.float32 or not t.is_contiguous() or t.numel() != n):
                raise TypeError("accumulation buffers must be contiguous float32 CUDA/HIP tensors of %d elements" % n)
        out = None
    else:
        out = torch.empty((B, Kout, N), dtype=torch.float32, device=dy.device)
    m_max = max([offs[b + 1] - offs[b] for b in range(B)] + [0])
    ws_bytes = lib.ggnn_xty_workspace_bytes(m_max, K, N, B)
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dy.device)
    segs = (ctypes.c_void_p * nseg)(*[x.data_ptr() for x in x_segs])
    ldx = (ctypes.c_int32 * nseg)(*[(x.stride(0) if x.shape[0] > 1 else Dseg) for x in x_segs])
    ro = (ctypes.c_int32 * (B + 1))(*offs)
    ldy = dy.stride(0) if M > 1 else N
    if x_rows is not None:
        _req(x_rows, torch.int32, "x_rows")
    if out is None:
        _launch("xty[K=%d,N=%d%s]" % (K, N, ",x%d" % B if batched else ""), lambda: lib.ggnn_xty_acc_f32(
            segs, nseg, Dseg, ldx, _ptr(x_rows), _ptr(dy), ldy, _ptr(add_to), _ptr(add_bias_to), 1, K, N, 1 if ones_row else 0, ro, B,
            _ptr(ws), ws_bytes, _stream()))
        return None
    _launch("xty[K=%d,N=%d%s]" % (K, N, ",x%d" % B if batched else ""), lambda: lib.ggnn_xty_f32(
        segs, nseg, Dseg, ldx, _ptr(x_rows), _ptr(dy), ldy, _ptr(out), K, N, 1 if ones_row else 0, ro, B, _ptr(ws), ws_bytes, _stream()))
    return out if batched else out[0]


def colsum(dy: torch.Tensor) -> torch.Tensor:
    """Deterministic column sums of a [M, N] float32 matrix (bias gradients)."""
    lib = _lib.load()
    M, N = dy.shape
    if N > 1024:                                       # (ggnn_colsum_f32 takes N <= 1024: wider matrices in 1024-column views)
        return torch.cat([colsum(dy[:, c:min(c + 1024, N)]) for c in range(0, N, 1024)])
    out = torch.empty(N, dtype=torch.float32, device=dy.device)
    ws_bytes = lib.ggnn_colsum_workspace_bytes(N)
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dy.device)
    _launch("colsum[N=%d]" % N, lambda: lib.ggnn_colsum_f32(_ptr(dy), dy.stride(0) if M > 1 else N, M, N, _ptr(out), _ptr(ws), ws_bytes, _stream()))
    return out


def segment_sum_rows_acc(rows: torch.Tensor, index, out: torch.Tensor) -> torch.Tensor:
    """out[s,:] += sum of rows[index.gather_row[slot],:] over the slots of segment s (ggnn_gather_segment_sum_acc_f32)."""
    lib = _lib.load()
    _req(rows, torch.float32, "rows"); _req(out, torch.float32, "out")
    return _segment_sum("gather_segment_sum_acc", rows, index.row_ptr, index.gather_row,
                        slot_heads(index, index.row_ptr, index.gather_row, index.num_nodes), None, None, False, out, index.num_nodes,
                        rows.shape[1], 1, accumulate=True)


def gru_bwd_is_fused(D: int) -> bool:
    return bool(_lib.load().ggnn_gru_bwd_is_fused(D))


def gru_bwd_fused(g, h, r, u, c, packed, nin, use_avg: bool, nx: int, activation: str, gather=None):
    """The GRU backward of one timestep in one launch (ggnn_gru_bwd_fused_f32) -> (dpc, dpg, rh, dh, [dx_0 .. dx_{nx-1}]);
    the last dx is d_incoming (already divided by the in-degree for mean aggregation).
    gather = (rows, heads): the incoming gradient is g[v] + the sum of the rows of `rows` named by the slot-head record heads[v]
    (ggnn_gru_bwd_fused_gather_f32)."""
    lib = _lib.load()
    V, D = h.shape
    T = nin.shape[1] if nin is not None else 1
    dev = h.device
    dpc = torch.empty_like(h); rh = torch.empty_like(h); dh = torch.empty_like(h)
    dpg = torch.empty((V, 2 * D), dtype=torch.float32, device=dev)
    dx = [torch.empty_like(h) for _ in range(nx)]
    dxp = (ctypes.c_void_p * nx)(*[t.data_ptr() for t in dx])
    if gather is not None:
        rows, heads = gather
        _req(rows, torch.float32, "rows"); _req(heads, torch.int32, "heads")
        _launch("gru_bwd_fused_gather[nx=%d]" % nx, lambda: lib.ggnn_gru_bwd_fused_gather_f32(
            _ptr(g), _ptr(rows), _ptr(heads), _ptr(h), _ptr(r), _ptr(u), _ptr(c), _ptr(packed), _ptr(dpc), _ptr(dpg), _ptr(rh), _ptr(dh),
            dxp, _ptr(nin), T, 1 if use_avg else 0, nx, V, D, ACT_IDS[activation.lower()], _stream()))
        return dpc, dpg, rh, dh, dx
    _launch("gru_bwd_fused[nx=%d]" % nx, lambda: lib.ggnn_gru_bwd_fused_f32(
        _ptr(g), _ptr(h), _ptr(r), _ptr(u), _ptr(c), None, None, _ptr(packed), _ptr(dpc), _ptr(dpg), _ptr(rh), _ptr(dh), dxp,
        _ptr(nin), T, 1 if use_avg else 0, nx, V, D, ACT_IDS[activation.lower()], _stream()))
    return dpc, dpg, rh, dh, dx


def weighted_segment_sum(rows: torch.Tensor, index: "SegmentIndex", weight_id: torch.Tensor, weights: torch.Tensor,
                         out: Optional[torch.Tensor] = None, accumulate: bool = False) -> torch.Tensor:
    """out[s] (+)= sum over the slots of segment s of weights[weight_id[slot]] * rows[index.gather_row[slot]]."""
    lib = _lib.load()
    _req(rows, torch.float32, "rows"); _req(weights, torch.float32, "weights"); _req(weight_id, torch.int32, "weight_id")
    D = rows.shape[1]
    nseg = index.num_nodes
    if out is None:
        out = torch.empty((nseg, D), dtype=torch.float32, device=rows.device)
    _launch("weighted_segment_sum", lambda: lib.ggnn_weighted_segment_sum_f32(
        _ptr(rows), _ptr(index.row_ptr), _ptr(index.gather_row), _ptr(weight_id), _ptr(weights), _ptr(out), 1 if accumulate else 0,
        nseg, D, _stream()))
    return out


def attn_backward_target(H: torch.Tensor, h: torch.Tensor, d_att: torch.Tensor, index: MessageIndex, type_factors: torch.Tensor,
                         dh: torch.Tensor):
    """Target-side pass of the propagation-attention backward (ggnn_attn_bwd_target_f32): adds sum_e ds_e f_t h[src_e] to dh[v]
    and returns the per-message (coef_a, coef_s, dfac), indexed by message id."""
    lib = _lib.load()
    V, D = h.shape
    T, M = index.num_edge_types, index.num_messages
    dev = h.device
    ca = torch.empty(max(M, 1), dtype=torch.float32, device=dev); cs = torch.empty_like(ca); df = torch.empty_like(ca)
    _launch("attn_bwd_target", lambda: lib.ggnn_attn_bwd_target_f32(
        _ptr(H), _ptr(h), _ptr(d_att), _ptr(index.row_ptr), _ptr(index.gather_row), _ptr(index.msg_perm), _ptr(type_factors), _ptr(ca),
        _ptr(cs), _ptr(df), _ptr(dh), 1, V, D, T, _stream()))
    return ca, cs, df


def range_sum(values: torch.Tensor, offsets: Sequence[int]) -> torch.Tensor:
    """out[b] = sum(values[offsets[b]:offsets[b+1]]), deterministic (ggnn_range_sum_f32)."""
    lib = _lib.load()
    B = len(offsets) - 1
    out = torch.empty(B, dtype=torch.float32, device=values.device)
    off = (ctypes.c_int64 * (B + 1))(*[int(o) for o in offsets])
    check(lib.ggnn_range_sum_f32(_ptr(values), off, B, _ptr(out), _stream()))
    return out


def bwd_dx(dY: torch.Tensor, nseg_y: int, WT: torch.Tensor, xcols: int, split_inc: bool, dx: Optional[torch.Tensor],
           dinc: Optional[torch.Tensor], nin: Optional[torch.Tensor], use_avg: bool, dh: Optional[torch.Tensor], acc_dx: bool,
           acc_dh: bool, D: int) -> None:
    """Q = dY WT with the backward epilogue of ggnn_bwd_dx_f32 (x columns -> dx / dinc, h columns -> dh)."""
    lib = _lib.load()
    V = dY.shape[0]
    K = WT.shape[1]
    T = nin.shape[1] if nin is not None else 1
    _launch("bwd_dx[K=%d]" % K, lambda: lib.ggnn_bwd_dx_f32(
        _ptr(dY), dY.stride(0) if V > 1 else dY.shape[1], nseg_y, _ptr(WT), K, _ptr(dx), xcols, 1 if split_inc else 0, _ptr(dinc),
        _ptr(nin), T, 1 if use_avg else 0, _ptr(dh), 1 if acc_dx else 0, 1 if acc_dh else 0, V, D, _stream()))


def act_bwd(g: torch.Tensor, out: torch.Tensor, activation: str) -> torch.Tensor:
    """dP = g * act'(out) (BasicRNNCell backward)."""
    lib = _lib.load()
    V, D = out.shape
    dP = torch.empty_like(out)
    _launch("act_bwd", lambda: lib.ggnn_act_bwd_f32(_ptr(g), _ptr(out), ACT_IDS[activation.lower()], _ptr(dP), V, D, _stream()))
    return dP


def cudnn_gru_train(x_segs: Sequence[torch.Tensor], h: torch.Tensor, Wg, bg, Wcx, bcx, Wch, bch):
    """ops.cudnn_gru that keeps what the backward pass needs: -> (h', r, u, c, hc)."""
    lib = _lib.load()
    V, D = h.shape
    nx = len(x_segs)
    out = torch.empty_like(h); c = torch.empty_like(h)
    ws_bytes = lib.ggnn_cudnn_gru_workspace_bytes(V, D)
    ws = torch.empty(max(ws_bytes // 4, 1), dtype=torch.float32, device=h.device)
    segs = (ctypes.c_void_p * nx)(*[_req(x, torch.float32, "x").data_ptr() for x in x_segs])
    _launch("cudnn_gru[nx=%d]" % nx, lambda: lib.ggnn_cudnn_gru_train_f32(segs, nx, _ptr(h), _ptr(Wg), _ptr(bg), _ptr(Wcx), _ptr(bcx),
                                                                          _ptr(Wch), _ptr(bch), _ptr(out), _ptr(c), _ptr(ws), ws_bytes,
                                                                          V, D, _stream()))
    n = V * D
    return out, ws[2 * n:3 * n].view(V, D), ws[n:2 * n].view(V, D), c, ws[3 * n:4 * n].view(V, D)


def cudnn_gru_bwd_stage(g, h, r, u, c, hc):
    """Element-wise part of the CudnnCompatibleGRUCell backward -> (dpc, dpg = [dpr|dpu], dh = g*u, dhc)."""
    lib = _lib.load()
    V, D = h.shape
    dpc = torch.empty_like(h); dh = torch.empty_like(h); dhc = torch.empty_like(h)
    dpg = torch.empty((V, 2 * D), dtype=torch.float32, device=h.device)
    _launch("cudnn_gru_bwd_stage", lambda: lib.ggnn_cudnn_gru_bwd_stage_f32(_ptr(g), _ptr(h), _ptr(r), _ptr(u), _ptr(c), _ptr(hc),
                                                                             _ptr(dpc), _ptr(dpg), _ptr(dh), _ptr(dhc), V, D, _stream()))
    return dpc, dpg, dh, dhc


def dropout(x: torch.Tensor, keep_prob: float, seed: int, row_key: Optional[torch.Tensor] = None, row_key_base: int = 0,
            out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """tf.nn.dropout with the counter-based mask of ggnn_dropout_f32: x / keep * floor(keep + U(seed, row key, column)).
    x [rows, cols] (or [cols]: one row) contiguous float32; row_key: optional int64 [rows] (default row_key_base + row)."""
    lib = _lib.load()
    _req(x, torch.float32, "x")
    if x.dim() < 1:
        raise ValueError("x must have at least one dimension")
    cols = x.shape[-1]
    rows = x.numel() // cols if cols else 0
    if row_key is not None:
        _req(row_key, torch.int64, "row_key")
        if row_key.numel() != rows:
            raise ValueError("row_key must have one entry per row")
    if out is None:
        out = torch.empty_like(x)
    _launch("dropout", lambda: lib.ggnn_dropout_f32(_ptr(x), _ptr(out), _ptr(row_key), int(row_key_base),
                                                    int(seed) & 0xFFFFFFFFFFFFFFFF, float(keep_prob), rows, cols, _stream()))
    return out


def unsorted_segment_sum(data: torch.Tensor, segment_ids: torch.Tensor, num_segments: int) -> torch.Tensor:
    """tf.unsorted_segment_sum (fp32 atomics; any id order).  data [M,D] or [M], ids [M] int32."""
    lib = _lib.load()
    _req(data, torch.float32, "data"); _req(segment_ids, torch.int32, "segment_ids")
    M = data.shape[0]
    D = 1 if data.dim() == 1 else data.shape[1]
    out = torch.empty((num_segments,) if data.dim() == 1 else (num_segments, D), dtype=torch.float32, device=data.device)
    check(lib.ggnn_unsorted_segment_sum_f32(_ptr(data), _ptr(segment_ids), _ptr(out), M, D, num_segments, _stream()))
    return out


def segment_sum_rows_by_index(rows: torch.Tensor, index: MessageIndex, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """out[s,:] = sum of rows[index.gather_row[slot],:] over the slots of segment s -- the plain
    gather/segment-sum kernel without epilogue.  With a source index (build_source_index) and
    rows = d_incoming [V,D] this is the backward of the forward gather: out = dH viewed as [V*T, D]."""
    lib = _lib.load()
    _req(rows, torch.float32, "rows")
    D = rows.shape[1]
    nseg = index.num_nodes
    if out is None:
        out = torch.empty((nseg, D), dtype=torch.float32, device=rows.device)
    return _segment_sum("gather_segment_sum_bwd", rows, index.row_ptr, index.gather_row,
                        slot_heads(index, index.row_ptr, index.gather_row, nseg), None, None, False, out, nseg, D, 1)


def dense_aggregate(adjacency: torch.Tensor, Hm: torch.Tensor, edge_biases: Optional[torch.Tensor],
                    out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """acts = sum_e A_e (h W_e + b_e)  (chem_tensorflow_dense.py:103-112).
    adjacency [b,e,v,v] f32, Hm [b*v, e*D] (msg_transform output), edge_biases [e,D] or None -> [b*v, D]."""
    lib = _lib.load()
    _req(adjacency, torch.float32, "adjacency"); _req(Hm, torch.float32, "Hm")
    b, E, v, v2 = adjacency.shape
    if v != v2 or Hm.shape[0] != b * v or Hm.shape[1] % E:
        raise ValueError("shape mismatch between adjacency [b,e,v,v] and Hm [b*v, e*D]")
    D = Hm.shape[1] // E
    if edge_biases is not None:
        _req(edge_biases, torch.float32, "edge_biases")
        if edge_biases.shape != (E, D):
            raise ValueError("edge_biases must be [e,D]")
    if out is None:
        out = torch.empty((b * v, D), dtype=torch.float32, device=Hm.device)
    _launch("dense_aggregate", lambda: lib.ggnn_dense_aggregate_f32(_ptr(adjacency), _ptr(Hm), _ptr(edge_biases), _ptr(out),
                                                                   b, v, E, D, _stream()))
    return out


def dense_propagate_supported(v: int, E: int, D: int) -> bool:
    return bool(_lib.load().ggnn_dense_propagate_supported(int(v), int(E), int(D)))


def dense_propagate(h0: torch.Tensor, adjacency: torch.Tensor, edge_packed: torch.Tensor, gru_packed: torch.Tensor,
                    edge_biases: Optional[torch.Tensor], bg: torch.Tensor, bc: torch.Tensor, steps: int,
                    fmt: int = GRU_FMT_EXACT) -> torch.Tensor:
    """The whole dense forward (chem_tensorflow_dense.py:93-117) in one launch, graph-resident (ggnn_dense_propagate_f32).
    h0 [b,v,D], adjacency [b,e,v,v]; edge_packed = PackedWeights.dense_edge(W [e,D,D]); gru_packed = PackedWeights.dense_gru(Wg, Wc).
    fmt: operand format of the kernel's products (exact by default; formats.F16X2 when the caller has proven its range)."""
    lib = _lib.load()
    _req(h0, torch.float32, "h0"); _req(adjacency, torch.float32, "adjacency")
    b, v, D = h0.shape
    E = adjacency.shape[1]
    if adjacency.shape != (b, E, v, v):
        raise ValueError("adjacency must be [b,e,v,v]")
    for t, n, name in ((bg, 2 * D, "bg"), (bc, D, "bc")):
        _req(t, torch.float32, name)
        if t.numel() != n:
            raise ValueError("%s must have %d elements" % (name, n))
    if edge_biases is not None:
        _req(edge_biases, torch.float32, "edge_biases")
        if edge_biases.numel() != E * D:
            raise ValueError("edge_biases must be [e,D]")
    out = torch.empty_like(h0)
    _launch("dense_propagate[steps=%d]" % steps, lambda: lib.ggnn_dense_propagate_f32(
        _ptr(h0), _ptr(adjacency), _ptr(edge_packed), _ptr(gru_packed), _ptr(edge_biases), _ptr(bg), _ptr(bc), _ptr(out), b, v, E, D,
        int(steps), int(fmt), _stream()))
    return out


# ---- source-compacted message transform -----------------------------------------------------------------
@dataclass
class CompactSources:
    """Active (source node, edge type) pairs of a batch, type-major / node-ascending (built once per batch).

    pair_node     [R] int32   compact row -> source node
    type_row_off  list[T+1]   host: rows of type t are type_row_off[t] .. type_row_off[t+1]-1
    gather_row    [M] int32   message slot (MessageIndex order) -> compact row
    """
    pair_node: torch.Tensor
    type_row_off: List[int]
    gather_row: torch.Tensor

    @property
    def num_rows(self) -> int:
        return int(self.type_row_off[-1])


def compact_supported(D: int) -> bool:
    return bool(_lib.load().ggnn_msg_transform_compact_supported(D))


def gru_is_fused(D: int) -> bool:
    return bool(_lib.load().ggnn_gru_is_fused(D))


def gru_gather_fused(D: int) -> bool:
    """True when the fused GRU of this hidden size has the variant that gathers the segment sum inside the kernel
    (ggnn_gru_is_fused == 1: the whole-block kernels; the column-panel kernels of D = 128/192/256 report 2)."""
    return _lib.load().ggnn_gru_is_fused(D) == 1


def build_compact_sources(index: MessageIndex, type_row_off: Optional[Sequence[int]] = None) -> CompactSources:
    """Enumerate the (node, type) pairs that emit at least one message and re-target the segment-sum's
    gather rows at them.  index: the by-target MessageIndex of the batch (build_message_index).
    type_row_off (host ints [T+1]): the per-type row ranges, when the caller knows them (data_device sums per-molecule
    tables); otherwise they are read back from the device (one device->host sync)."""
    lib = _lib.load()
    T, V, M = index.num_edge_types, index.num_nodes, index.num_messages
    dev = index.adj.device
    lists = [index.adj[index.type_off[t]:index.type_off[t + 1]] for t in range(T)]
    src_index = build_source_index(lists, V)
    index._source_index = src_index                       # the backward pass wants the same structure
    pair_node = torch.empty(max(min(M, V * T), 1), dtype=torch.int32, device=dev)
    pair_id = torch.empty(max(V * T, 1), dtype=torch.int32, device=dev)
    off = torch.zeros(T + 1, dtype=torch.int32, device=dev)
    ws_bytes = lib.ggnn_compact_workspace_bytes(V, T)
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
    check(lib.ggnn_build_compact_sources(_ptr(src_index.row_ptr), V, T, _ptr(pair_node), _ptr(pair_id), _ptr(off), _ptr(ws),
                                         ws_bytes, _stream()))
    gather_c = torch.empty(M, dtype=torch.int32, device=dev)
    check(lib.ggnn_remap_gather_rows(_ptr(index.gather_row), _ptr(pair_id), _ptr(gather_c), M, _stream()))
    if type_row_off is None:
        type_row_off = [int(x) for x in off.cpu().tolist()]   # one device->host sync, once per batch
    else:
        type_row_off = [int(x) for x in type_row_off]
        if len(type_row_off) != T + 1:
            raise ValueError("type_row_off needs T + 1 entries")
        if os.environ.get("GGNN_CHECK_HOST_COUNTS", "0") != "0" and type_row_off != [int(x) for x in off.cpu().tolist()]:
            raise AssertionError("host-side pair counts %s differ from the device's %s" % (type_row_off, off.cpu().tolist()))
    return CompactSources(pair_node[:max(type_row_off[-1], 1)], type_row_off, gather_c)


def prepare_message_index(index: MessageIndex, hidden_size: int, compact: bool = True, training: bool = False,
                          type_row_off: Optional[Sequence[int]] = None) -> MessageIndex:
    """Everything the propagation derives from a batch's message index, built EAGERLY when the batch is packed (it
    used to be built lazily on the first forward of each batch, which put a cold pass -- three small launches and a
    device->host sync -- inside whatever region timed that forward): the active (source node, edge type) pairs of the
    compacted message transform for the hidden sizes that have one; training: also the transpose structures of the backward
    pass (CompactBackward and the slot heads of its segment sums) -- ~25 small launches that would otherwise run inside the
    first backward pass of every fresh batch, on the training stream, instead of on the packer's."""
    if compact and index.num_messages and compact_supported(hidden_size) and getattr(index, "_compact", None) is None:
        index._compact = build_compact_sources(index, type_row_off)
        slot_heads(index._compact, index.row_ptr, index._compact.gather_row, index.num_nodes)
    comp = getattr(index, "_compact", None)
    if training and comp is not None and comp.num_rows:
        bwd = compact_backward(index, comp)
        slot_heads(index, index.row_ptr, index.gather_row, index.num_nodes)
        for si in (bwd.rows_index, bwd.node_index, bwd.source_node_index):
            slot_heads(si, si.row_ptr, si.gather_row, si.num_nodes)
    return index


class SegmentIndex:
    """row_ptr / gather_row pair for segment_sum_rows_by_index (the fields of MessageIndex that kernel reads)."""

    def __init__(self, row_ptr: torch.Tensor, gather_row: torch.Tensor, num_nodes: int, msg: Optional[torch.Tensor] = None):
        self.row_ptr, self.gather_row, self.num_nodes = row_ptr, gather_row, num_nodes
        self.msg = msg                     # slot -> original message id (for per-message weights), where known


class CompactBackward:
    """Index structures of the compacted transform's backward pass (built once per batch, torch index arithmetic):

    rows_index   segments = compact rows (type-major): the message slots leaving (node,type) pair r, gather_row = dst
                 -> dHc[r] = sum of d_incoming[dst]            (the transpose of the forward gather)
    node_index   segments = nodes: the compact rows of node v (type ascending)
                 -> dh[v] += sum_t (dHc W_t^T)[row(v,t)]
    identity     CompactSources whose pair_node is arange(R): runs the compact transform kernel on [R,D] rows as they are
    """

    def __init__(self, index: MessageIndex, comp: CompactSources):
        V, T, M = index.num_nodes, index.num_edge_types, index.num_messages
        R = comp.num_rows
        dev = index.adj.device
        src = getattr(index, "_source_index", None)
        if src is None:
            lists = [index.adj[index.type_off[t]:index.type_off[t + 1]] for t in range(T)]
            src = index._source_index = build_source_index(lists, V)
        pn = comp.pair_node[:R].long()
        typ = torch.repeat_interleave(torch.arange(T, device=dev), torch.as_tensor(
            [comp.type_row_off[t + 1] - comp.type_row_off[t] for t in range(T)], device=dev), output_size=R)
        seg = pn * T + typ                                              # segment of the by-(src*T+type) index
        rp = src.row_ptr.long()
        start, length = rp[seg], rp[seg + 1] - rp[seg]
        ends = torch.cumsum(length, 0)
        rp_c = torch.zeros(R + 1, dtype=torch.int32, device=dev)
        rp_c[1:] = ends.to(torch.int32)
        slots = torch.repeat_interleave(start - (ends - length), length, output_size=M) + torch.arange(M, device=dev)
        self.rows_index = SegmentIndex(rp_c, src.gather_row[slots].contiguous(), R, src.msg_perm[slots].contiguous())
        # messages leaving a NODE (all its types: the by-(src,type) segments of a node are consecutive)
        self.source_node_index = SegmentIndex(src.row_ptr[::T].contiguous(), src.gather_row, V, src.msg_perm)
        order = torch.sort(pn, stable=True)[1]                          # rows by node, type ascending inside a node
        rp_n = torch.zeros(V + 1, dtype=torch.int32, device=dev)
        per_node = torch.zeros(V, dtype=torch.int64, device=dev).scatter_add_(0, pn, torch.ones(1, dtype=torch.int64, device=dev).expand(R))
        rp_n[1:] = torch.cumsum(per_node, 0).to(torch.int32)                # (scatter_add, not bincount: no host read-back)
        self.node_index = SegmentIndex(rp_n, order.to(torch.int32).contiguous(), V)
        self.identity = CompactSources(torch.arange(max(R, 1), dtype=torch.int32, device=dev), comp.type_row_off, comp.gather_row)


def compact_backward(index: MessageIndex, comp: CompactSources) -> CompactBackward:
    bwd = getattr(comp, "_bwd", None)
    if bwd is None:
        bwd = comp._bwd = CompactBackward(index, comp)
    return bwd


def msg_transform_compact(h: torch.Tensor, edge_weights: torch.Tensor, comp: CompactSources,
                          out: Optional[torch.Tensor] = None, fmt: int = GRU_FMT_EXACT) -> torch.Tensor:
    """Hc[r] = h[pair_node[r]] @ edge_weights[type(r)] for the active (node,type) pairs only
    (chem_tensorflow_sparse.py:160-164 without the duplicate / unused rows).  -> [R, D].
    fmt: operand format of the products (formats.BF16X3 exact, the default; F16X2 when the caller has proven its range)."""
    lib = _lib.load()
    _req(h, torch.float32, "h"); _req(edge_weights, torch.float32, "edge_weights")
    V, D = h.shape
    T = edge_weights.shape[0]
    if edge_weights.shape != (T, D, D) or len(comp.type_row_off) != T + 1:
        raise ValueError("edge_weights must be [T,D,D] matching the compact index")
    R = comp.num_rows
    if out is None:
        out = torch.empty((max(R, 1), D), dtype=torch.float32, device=h.device)
    ws_bytes = lib.ggnn_msg_transform_compact_workspace_bytes(D, T)
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=h.device)
    off = (ctypes.c_int64 * (T + 1))(*comp.type_row_off)
    _launch("msg_transform_compact", lambda: lib.ggnn_msg_transform_compact_f32(
        _ptr(h), _ptr(edge_weights), _ptr(comp.pair_node), off, _ptr(out), _ptr(ws), ws_bytes, V, D, T, int(fmt), _stream()))
    return out


def gather_segment_sum_compact(Hc: torch.Tensor, index: MessageIndex, comp: CompactSources,
                               num_incoming_edges_per_type: Optional[torch.Tensor], edge_biases: Optional[torch.Tensor],
                               use_avg: bool, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """gather_segment_sum over compact transformed rows (Hc from msg_transform_compact)."""
    lib = _lib.load()
    _req(Hc, torch.float32, "Hc")
    V, T, D = index.num_nodes, index.num_edge_types, Hc.shape[1]
    nin = num_incoming_edges_per_type
    if nin is not None:
        _req(nin, torch.float32, "num_incoming_edges_per_type")
    if edge_biases is not None:
        _req(edge_biases, torch.float32, "edge_biases")
    if out is None:
        out = torch.empty((V, D), dtype=torch.float32, device=Hc.device)
    return _segment_sum("gather_segment_sum", Hc, index.row_ptr, comp.gather_row, slot_heads(comp, index.row_ptr, comp.gather_row, V),
                        nin, edge_biases, use_avg, out, V, D, T)


# ---- pre-packed weights (inference) ---------------------------------------------------------------------
_GRU_IMAGES_CACHE = None


def _gru_images():
    """ops.gru's own image cache (raw-weight callers that ask for a packed operand format)."""
    global _GRU_IMAGES_CACHE
    if _GRU_IMAGES_CACHE is None:
        _GRU_IMAGES_CACHE = PackedWeights()
    return _GRU_IMAGES_CACHE


class PackedWeights:
    """Cache of the kernels' LDS stage images of weight tensors, keyed by (storage pointer, version counter):
    during inference the weights do not change between batches, so the pack pre-pass runs once per weight
    version instead of once per launch.  Any in-place update of a weight bumps tensor._version and repacks."""

    def __init__(self):
        self._gru = {}
        self._edge = {}

    @staticmethod
    def _base(t: torch.Tensor) -> torch.Tensor:
        return t._base if t._base is not None else t

    @classmethod
    def _key(cls, *tensors):
        # identity of the owning tensor objects (views share their base's version counter) + view geometry
        return tuple((id(cls._base(t)), t.data_ptr(), tuple(t.shape)) for t in tensors)

    @classmethod
    def _lookup(cls, table, key, tensors):
        """A hit is valid only while the SAME tensor objects are alive at the same version: a pointer/version
        pair alone can be recycled by the allocator for a different model's weights."""
        entry = table.get(key)
        if entry is None:
            return None
        refs, versions, packed, ready, stream_id = entry
        for r, v, t in zip(refs, versions, tensors):
            if r() is not cls._base(t) or v != t._version:
                return None
        # the image was written on another stream: order this stream after the pack launch (an event wait on the device --
        # a host synchronisation here used to drain the GPU queue once per layer and training step)
        if ready is not None:
            cur = torch.cuda.current_stream()
            if cur.cuda_stream != stream_id:
                if not ready.query():
                    cur.wait_event(ready)
                packed.record_stream(cur)                    # (its memory belongs to the packing stream's pool)
        return packed

    @classmethod
    def _store(cls, table, key, tensors, packed):
        import weakref
        if len(table) >= 64:     # drop the images of weights that no longer exist; never evict a live model's
            for k in [k for k, e in table.items() if any(r() is None for r in e[0])]:
                del table[k]
            if len(table) >= 1024:
                table.clear()
        ready, stream_id = None, 0
        if packed.is_cuda:
            ready = torch.cuda.Event()
            ready.record()                                   # after the pack launch, on the stream that ran it
            stream_id = torch.cuda.current_stream().cuda_stream
        table[key] = ([weakref.ref(cls._base(t)) for t in tensors], [t._version for t in tensors], packed, ready, stream_id)
        return packed

    def gru(self, Wg: torch.Tensor, Wc: torch.Tensor, nx: int, D: int, fmt: int = GRU_FMT_EXACT) -> torch.Tensor:
        """The fused GRU forward's stage images in the operand format `fmt` (formats.F16X2 / BF16X3; the launch that consumes
        them must be given the same format)."""
        lib = _lib.load()
        key = self._key(Wg, Wc) + (int(fmt),)
        hit = self._lookup(self._gru, key, (Wg, Wc))
        if hit is None:
            _req(Wg, torch.float32, "Wg"); _req(Wc, torch.float32, "Wc")
            packed = torch.empty(lib.ggnn_gru_packed_bytes(D, nx) // 4, dtype=torch.float32, device=Wg.device)
            check(lib.ggnn_gru_pack_weights_f32(_ptr(Wg), _ptr(Wc), nx, D, int(fmt), _ptr(packed), _stream()))
            hit = self._store(self._gru, key, (Wg, Wc), packed)
        return hit

    def dense_edge(self, W: torch.Tensor) -> torch.Tensor:
        """The E f32 stage images of the graph-resident dense kernel (ggnn_dense_edge_pack_f32), once per weight version."""
        lib = _lib.load()
        if not hasattr(self, "_dense_edge"):
            self._dense_edge = {}
        key = self._key(W)
        hit = self._lookup(self._dense_edge, key, (W,))
        if hit is None:
            _req(W, torch.float32, "edge_weights")
            T, D = W.shape[0], W.shape[1]
            packed = torch.empty(lib.ggnn_dense_edge_packed_bytes(D, T) // 4, dtype=torch.float32, device=W.device)
            check(lib.ggnn_dense_edge_pack_f32(_ptr(W), T, D, _ptr(packed), _stream()))
            hit = self._store(self._dense_edge, key, (W,), packed)
        return hit

    def dense_gru(self, Wg: torch.Tensor, Wc: torch.Tensor, D: int) -> torch.Tensor:
        """The six plain stage images of the graph-resident dense kernel (ggnn_dense_gru_pack_f32), once per weight version."""
        lib = _lib.load()
        if not hasattr(self, "_dense_gru"):
            self._dense_gru = {}
        key = self._key(Wg, Wc)
        hit = self._lookup(self._dense_gru, key, (Wg, Wc))
        if hit is None:
            _req(Wg, torch.float32, "Wg"); _req(Wc, torch.float32, "Wc")
            packed = torch.empty(lib.ggnn_dense_gru_packed_bytes(D) // 4, dtype=torch.float32, device=Wg.device)
            check(lib.ggnn_dense_gru_pack_f32(_ptr(Wg), _ptr(Wc), D, _ptr(packed), _stream()))
            hit = self._store(self._dense_gru, key, (Wg, Wc), packed)
        return hit

    def gru_bwd(self, Wg: torch.Tensor, Wc: torch.Tensor, nx: int, D: int) -> torch.Tensor:
        """Transposed-block stage images of the fused GRU backward (ggnn_gru_bwd_fused_f32), once per weight version."""
        lib = _lib.load()
        if not hasattr(self, "_gru_bwd"):
            self._gru_bwd = {}
        key = self._key(Wg, Wc)
        hit = self._lookup(self._gru_bwd, key, (Wg, Wc))
        if hit is None:
            _req(Wg, torch.float32, "Wg"); _req(Wc, torch.float32, "Wc")
            packed = torch.empty(lib.ggnn_gru_bwd_packed_bytes(D, nx) // 4, dtype=torch.float32, device=Wg.device)
            check(lib.ggnn_gru_bwd_fused_f32(None, None, None, None, None, _ptr(Wg), _ptr(Wc), _ptr(packed), None, None, None, None,
                                             None, None, 0, 0, nx, 0, D, 0, _stream()))
            hit = self._store(self._gru_bwd, key, (Wg, Wc), packed)
        return hit

    def edge(self, W: torch.Tensor, fmt: int = GRU_FMT_EXACT) -> torch.Tensor:
        """The compacted transform's stage images of W [T, D, D] in the operand format `fmt` (the launch that consumes them must be
        given the same format)."""
        lib = _lib.load()
        key = self._key(W) + (int(fmt),)
        hit = self._lookup(self._edge, key, (W,))
        if hit is None:
            _req(W, torch.float32, "edge_weights")
            T, D = W.shape[0], W.shape[1]
            packed = torch.empty(lib.ggnn_msg_transform_compact_workspace_bytes(D, T) // 4, dtype=torch.float32, device=W.device)
            check(lib.ggnn_edge_weights_pack_f32(_ptr(W), T, D, int(fmt), _ptr(packed), _stream()))
            hit = self._store(self._edge, key, (W,), packed)
        return hit


def gru_packed(x_segs: Sequence[torch.Tensor], h: torch.Tensor, packed: torch.Tensor, bg: torch.Tensor, bc: torch.Tensor,
               activation: str = "tanh", out: Optional[torch.Tensor] = None,
               tile_counter: Optional[torch.Tensor] = None, fmt: int = GRU_FMT_EXACT, save: Optional[dict] = None) -> torch.Tensor:
    """ops.gru with pre-packed weight images (fused hidden sizes only).  tile_counter: optional int32 device tensor
    holding 0 (one element, consumed by this launch): dynamic tile hand-out, see include/ggnn_hip.h.
    fmt: the operand format `packed` was packed in (PackedWeights.gru).  save: optional dict receiving 'r','u','c'."""
    lib = _lib.load()
    _req(h, torch.float32, "h")
    V, D = h.shape
    nx = len(x_segs)
    for i, x in enumerate(x_segs):
        _req(x, torch.float32, "x_segs[%d]" % i)
        if x.shape != (V, D):
            raise ValueError("x_segs[%d] must be [V,D]" % i)
    act = ACT_IDS.get(activation.lower())
    if act is None:
        raise Exception("Unknown activation function type '%s'." % activation)
    if out is None:
        out = torch.empty_like(h)
    segs = (ctypes.c_void_p * nx)(*[x.data_ptr() for x in x_segs])
    sr = su = sc = None
    if save is not None:
        sr = save["r"] = torch.empty_like(h)
        su = save["u"] = torch.empty_like(h)
        sc = save["c"] = torch.empty_like(h)
    _launch("gru_fused[nx=%d]" % nx, lambda: lib.ggnn_gru_packed_f32(
        segs, nx, _ptr(h), _ptr(packed), _ptr(bg), _ptr(bc), _ptr(out), _ptr(sr), _ptr(su), _ptr(sc), V, D, act, int(fmt),
        _ptr(tile_counter), _stream()))
    return out


def msg_transform_compact_packed(h: torch.Tensor, packed: torch.Tensor, T: int, comp: CompactSources,
                                 out: Optional[torch.Tensor] = None, fmt: int = GRU_FMT_EXACT) -> torch.Tensor:
    """ops.msg_transform_compact with pre-packed edge-weight images (PackedWeights.edge(W, fmt): same fmt here)."""
    lib = _lib.load()
    _req(h, torch.float32, "h")
    V, D = h.shape
    R = comp.num_rows
    if out is None:
        out = torch.empty((max(R, 1), D), dtype=torch.float32, device=h.device)
    off = (ctypes.c_int64 * (T + 1))(*comp.type_row_off)
    _launch("msg_transform_compact", lambda: lib.ggnn_msg_transform_compact_f32(
        _ptr(h), None, _ptr(comp.pair_node), off, _ptr(out), _ptr(packed), packed.numel() * 4, V, D, T, int(fmt), _stream()))
    return out


def _ptr_array(tensors):
    if tensors is None:
        return None
    return (ctypes.c_void_p * len(tensors))(*[None if t is None else t.data_ptr() for t in tensors])


def sparse_propagate(h0: torch.Tensor, index: MessageIndex, comp: Optional[CompactSources], nin: torch.Tensor, use_avg: bool,
                     layer_timesteps: Sequence[int], residuals: Sequence[Sequence[int]],
                     edge_w: Sequence[torch.Tensor], edge_packed: Optional[Sequence[torch.Tensor]],
                     edge_bias: Optional[Sequence[Optional[torch.Tensor]]],
                     Wg: Sequence[torch.Tensor], bg: Sequence[torch.Tensor], Wc: Sequence[torch.Tensor], bc: Sequence[torch.Tensor],
                     gru_packed: Optional[Sequence[torch.Tensor]], activation: str,
                     fuse_gather: Optional[bool] = None, gru_fmt: Optional[Sequence[int]] = None,
                     edge_fmt: Optional[Sequence[int]] = None) -> List[torch.Tensor]:
    """chem_tensorflow_sparse.py:131-218 in ONE native call (ggnn_sparse_propagate_f32): returns
    node_states_per_layer[1:], the last entry being the final node representations.
    gru_fmt / edge_fmt: per layer, the operand format gru_packed[l] / edge_packed[l] was packed in (None: BF16X3 for every layer).
    fuse_gather (default FUSE_GATHER): gather the segment sum inside the GRU kernel where the layer allows it."""
    if fuse_gather is None:
        fuse_gather = FUSE_GATHER
    lib = _lib.load()
    _req(h0, torch.float32, "h0")
    V, D = h0.shape
    T, L = index.num_edge_types, len(layer_timesteps)
    act = ACT_IDS.get(activation.lower())
    if act is None:
        raise Exception("Unknown activation function type '%s'." % activation)
    rows = comp.num_rows if comp is not None else -1
    ws_bytes = lib.ggnn_sparse_propagate_workspace_bytes(V, D, T, rows)
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=h0.device)
    outs = [torch.empty_like(h0) for _ in range(L)]
    res_ptr = [0]
    res_idx = []
    for r in residuals:
        res_idx.extend(int(i) for i in r)
        res_ptr.append(len(res_idx))
    i32 = lambda xs: (ctypes.c_int32 * max(len(xs), 1))(*xs)
    off = None if comp is None else (ctypes.c_int64 * (T + 1))(*comp.type_row_off)
    gather = index.gather_row if comp is None else comp.gather_row
    _launch("sparse_propagate", lambda: lib.ggnn_sparse_propagate_f32(
        _ptr(h0), V, D, T, _ptr(index.row_ptr), _ptr(gather), None if comp is None else _ptr(comp.pair_node), off,
        _ptr(nin), 1 if use_avg else 0, L, i32([int(x) for x in layer_timesteps]), i32(res_ptr), i32(res_idx),
        _ptr_array(edge_w), _ptr_array(edge_packed), _ptr_array(edge_bias), _ptr_array(Wg), _ptr_array(bg), _ptr_array(Wc),
        _ptr_array(bc), _ptr_array(gru_packed), None if gru_fmt is None else i32([int(f) for f in gru_fmt]),
        None if edge_fmt is None else i32([int(f) for f in edge_fmt]), act, int(fuse_gather),
        _ptr_array(outs), _ptr(ws), ws_bytes, _stream()))
    return outs


def gru_packed_gather(residual_segs: Sequence[torch.Tensor], h: torch.Tensor, packed: torch.Tensor, bg: torch.Tensor,
                      bc: torch.Tensor, H: torch.Tensor, index: MessageIndex, gather_row: Optional[torch.Tensor],
                      num_incoming_edges_per_type: Optional[torch.Tensor], activation: str = "tanh",
                      tile_counter: Optional[torch.Tensor] = None, save: Optional[dict] = None,
                      fmt: int = GRU_FMT_EXACT) -> torch.Tensor:
    """chem_tensorflow_sparse.py:198-216 in one launch (ggnn_gru_packed_gather_f32): the GRU whose last input
    segment -- the aggregated messages -- is summed from the transformed rows `H` inside the kernel.
    save (dict): filled with r, u, c and the gathered segment "incoming" (training: what the backward pass needs)."""
    lib = _lib.load()
    _req(h, torch.float32, "h"); _req(H, torch.float32, "H")
    V, D = h.shape
    act = ACT_IDS.get(activation.lower())
    if act is None:
        raise Exception("Unknown activation function type '%s'." % activation)
    segs = (ctypes.c_void_p * max(len(residual_segs), 1))(*[s.data_ptr() for s in residual_segs])
    out = torch.empty_like(h)
    nin = num_incoming_edges_per_type
    gather = index.gather_row if gather_row is None else gather_row
    if save is not None:
        for k in ("r", "u", "c", "incoming"):
            save[k] = torch.empty_like(h)
        _launch("gru_fused_gather_train[nx=%d]" % (len(residual_segs) + 1), lambda: lib.ggnn_gru_packed_gather_train_f32(
            segs, len(residual_segs) + 1, _ptr(h), _ptr(packed), _ptr(bg), _ptr(bc), _ptr(out), _ptr(H), _ptr(index.row_ptr),
            _ptr(gather), None if nin is None else _ptr(nin), index.num_edge_types, 0 if nin is None else 1, _ptr(save["r"]),
            _ptr(save["u"]), _ptr(save["c"]), _ptr(save["incoming"]), V, D, act, int(fmt), _ptr(tile_counter), _stream()))
        return out
    _launch("gru_fused_gather[nx=%d]" % (len(residual_segs) + 1), lambda: lib.ggnn_gru_packed_gather_f32(
        segs, len(residual_segs) + 1, _ptr(h), _ptr(packed), _ptr(bg), _ptr(bc), _ptr(out), _ptr(H), _ptr(index.row_ptr),
        _ptr(gather), None if nin is None else _ptr(nin), index.num_edge_types, 0 if nin is None else 1, V, D, act, int(fmt),
        _ptr(tile_counter), _stream()))
    return out


def gated_readout(last_h: torch.Tensor, h0: torch.Tensor, graph_nodes_list: torch.Tensor, num_graphs: int,
                  gate_W: torch.Tensor, gate_b: torch.Tensor, transform_W: torch.Tensor, transform_b: torch.Tensor) -> torch.Tensor:
    """Fused gated_regression (chem_tensorflow_sparse.py:220-231): per-graph sum of sigmoid(gate)*transform."""
    lib = _lib.load()
    _req(last_h, torch.float32, "last_h"); _req(h0, torch.float32, "h0"); _req(graph_nodes_list, torch.int32, "graph_nodes_list")
    V, D = last_h.shape
    for n, w, numel in (("gate_W", gate_W, 2 * D), ("gate_b", gate_b, 1), ("transform_W", transform_W, D), ("transform_b", transform_b, 1)):
        _req(w, torch.float32, n)
        if w.numel() != numel:
            raise ValueError("%s must have %d elements" % (n, numel))
    out = torch.empty(int(num_graphs), dtype=torch.float32, device=last_h.device)
    _launch("gated_readout", lambda: lib.ggnn_gated_readout_f32(_ptr(last_h), _ptr(h0), _ptr(graph_nodes_list), _ptr(gate_W),
                                                                _ptr(gate_b), _ptr(transform_W), _ptr(transform_b), _ptr(out),
                                                                V, D, int(num_graphs), _stream()))
    return out


def readout_loss_fwd(last_h: torch.Tensor, h0: torch.Tensor, graph_nodes_list: torch.Tensor, graph_ptr: Optional[torch.Tensor],
                     node_mask: Optional[torch.Tensor], num_graphs: int, gate_W: torch.Tensor, gate_b: torch.Tensor,
                     transform_W: torch.Tensor, transform_b: torch.Tensor, target: Optional[torch.Tensor],
                     mask: Optional[torch.Tensor]):
    """Fused gated_regression + masked loss sums (chem_tensorflow_sparse.py:220-231, chem_tensorflow.py:158-170), forward:
    -> (out [G], node_gate [V], node_val [V], stats [3] = (sum 0.5 diff^2, sum |diff|, sum mask) or None without target).
    graph_nodes_list must be non-decreasing (batcher output); deterministic."""
    lib = _lib.load()
    _req(last_h, torch.float32, "last_h"); _req(h0, torch.float32, "h0"); _req(graph_nodes_list, torch.int32, "graph_nodes_list")
    V, D = last_h.shape
    G = int(num_graphs)
    dev = last_h.device
    out = torch.empty(G, dtype=torch.float32, device=dev)
    gate = torch.empty(max(V, 1), dtype=torch.float32, device=dev)
    val = torch.empty(max(V, 1), dtype=torch.float32, device=dev)
    stats = torch.empty(3, dtype=torch.float32, device=dev) if target is not None else None
    ws_bytes = lib.ggnn_readout_workspace_bytes(V, D, G)
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
    for n, t in (("gate_W", gate_W), ("gate_b", gate_b), ("transform_W", transform_W), ("transform_b", transform_b)):
        _req(t, torch.float32, n)
    _launch("readout_loss_fwd", lambda: lib.ggnn_readout_loss_fwd_f32(
        _ptr(last_h), _ptr(h0), _ptr(graph_nodes_list), _ptr(graph_ptr), _ptr(node_mask), _ptr(gate_W), _ptr(gate_b), _ptr(transform_W),
        _ptr(transform_b), _ptr(target), _ptr(mask), _ptr(out), _ptr(gate), _ptr(val), _ptr(stats), _ptr(ws), ws_bytes, V, D, G, _stream()))
    return out, gate, val, stats


def readout_loss_bwd(last_h, h0, graph_nodes_list, node_mask, num_graphs, gate_W, transform_W, gate, val, out, target, mask,
                     d_out: Optional[torch.Tensor], d_stats: Optional[torch.Tensor], d_last_h: Optional[torch.Tensor] = None,
                     grad_out: Optional[Sequence[torch.Tensor]] = None):
    """Backward of readout_loss_fwd: -> (d_last_h [V,D], d_gate_W [2D], d_gate_b [1], d_transform_W [D], d_transform_b [1]).
    d_last_h given: the gradient is ADDED to it (further tasks of a multi-task model).
    grad_out: four contiguous float32 buffers (2D, 1, D, 1 elements) that receive the weight gradients (the optimizer's flat
    gradient views) instead of fresh tensors."""
    lib = _lib.load()
    V, D = last_h.shape
    G = int(num_graphs)
    dev = last_h.device
    accumulate = d_last_h is not None
    if d_last_h is None:
        d_last_h = torch.empty_like(last_h) if V and G else torch.zeros_like(last_h)
    if grad_out is not None:
        dgW, dgb, dtW, dtb = grad_out
        for t, n in ((dgW, 2 * D), (dgb, 1), (dtW, D), (dtb, 1)):
            if not t.is_cuda or t.dtype != torch.float32 or not t.is_contiguous() or t.numel() != n:
                raise TypeError("grad_out buffers must be contiguous float32 CUDA/HIP tensors of 2D, 1, D and 1 elements")
    else:
        dgW = torch.empty(2 * D, dtype=torch.float32, device=dev); dgb = torch.empty(1, dtype=torch.float32, device=dev)
        dtW = torch.empty(D, dtype=torch.float32, device=dev); dtb = torch.empty(1, dtype=torch.float32, device=dev)
    ws_bytes = lib.ggnn_readout_workspace_bytes(V, D, G)
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
    _launch("readout_loss_bwd", lambda: lib.ggnn_readout_loss_bwd_f32(
        _ptr(last_h), _ptr(h0), _ptr(graph_nodes_list), _ptr(node_mask), _ptr(gate_W), _ptr(transform_W), _ptr(gate), _ptr(val), _ptr(out),
        _ptr(target), _ptr(mask), _ptr(d_out), _ptr(d_stats), _ptr(d_last_h), 1 if accumulate else 0, _ptr(dgW), _ptr(dgb), _ptr(dtW),
        _ptr(dtb), _ptr(ws), ws_bytes, V, D, G, _stream()))
    return d_last_h, dgW, dgb, dtW, dtb


# ---- the remaining switches of the same function: attention, RNN cell, cudnn-compatible GRU cell ------------------
def gather_segment_sum_attn(H: torch.Tensor, h: torch.Tensor, index: MessageIndex, type_factors: torch.Tensor,
                            num_incoming_edges_per_type: Optional[torch.Tensor], edge_biases: Optional[torch.Tensor],
                            use_avg: bool, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """use_propagation_attention (chem_tensorflow_sparse.py:147-149, 170-196) fused into the segment sum.
    H [V, T*D] from msg_transform (dense form), h [V,D] current states, type_factors [T]."""
    lib = _lib.load()
    _req(H, torch.float32, "H"); _req(h, torch.float32, "h"); _req(type_factors, torch.float32, "type_factors")
    V, T = index.num_nodes, index.num_edge_types
    D = h.shape[1]
    if H.shape != (V, T * D) or type_factors.shape != (T,):
        raise ValueError("shape mismatch: H [V,T*D], h [V,D], type_factors [T]")
    nin = num_incoming_edges_per_type
    if out is None:
        out = torch.empty((V, D), dtype=torch.float32, device=h.device)
    _launch("gather_segment_sum_attn", lambda: lib.ggnn_gather_segment_sum_attn_f32(
        _ptr(H), _ptr(h), _ptr(index.row_ptr), _ptr(index.gather_row), _ptr(type_factors), _ptr(nin), _ptr(edge_biases),
        1 if use_avg else 0, _ptr(out), V, D, T, _stream()))
    return out


def rnn(x_segs: Sequence[torch.Tensor], h: torch.Tensor, W: torch.Tensor, b: torch.Tensor, activation: str = "tanh") -> torch.Tensor:
    """tf.nn.rnn_cell.BasicRNNCell (chem_tensorflow_sparse.py:109-110): act([x|h] W + b)."""
    lib = _lib.load()
    _req(h, torch.float32, "h"); _req(W, torch.float32, "W"); _req(b, torch.float32, "b")
    V, D = h.shape
    nx = len(x_segs)
    if tuple(W.shape) != ((nx + 1) * D, D) or tuple(b.shape) != (D,):
        raise ValueError("W must be [(nx+1)D, D] and b [D]")
    out = torch.empty_like(h)
    segs = (ctypes.c_void_p * nx)(*[_req(x, torch.float32, "x").data_ptr() for x in x_segs])
    act = ACT_IDS.get(activation.lower())
    if act is None:
        raise Exception("Unknown activation function type '%s'." % activation)
    _launch("rnn[nx=%d]" % nx, lambda: lib.ggnn_rnn_f32(segs, nx, _ptr(h), _ptr(W), _ptr(b), _ptr(out), V, D, act, _stream()))
    return out


def cudnn_gru(x_segs: Sequence[torch.Tensor], h: torch.Tensor, Wg, bg, Wcx, bcx, Wch, bch) -> torch.Tensor:
    """tf.contrib.cudnn_rnn.CudnnCompatibleGRUCell (chem_tensorflow_sparse.py:105-108)."""
    lib = _lib.load()
    V, D = h.shape
    nx = len(x_segs)
    for n, w, shp in (("Wg", Wg, ((nx + 1) * D, 2 * D)), ("bg", bg, (2 * D,)), ("Wcx", Wcx, (nx * D, D)), ("bcx", bcx, (D,)),
                      ("Wch", Wch, (D, D)), ("bch", bch, (D,))):
        _req(w, torch.float32, n)
        if tuple(w.shape) != shp:
            raise ValueError("%s must have shape %s, got %s" % (n, shp, tuple(w.shape)))
    out = torch.empty_like(h)
    ws_bytes = lib.ggnn_cudnn_gru_workspace_bytes(V, D)
    ws = torch.empty(max(ws_bytes // 4, 1), dtype=torch.float32, device=h.device)
    segs = (ctypes.c_void_p * nx)(*[_req(x, torch.float32, "x").data_ptr() for x in x_segs])
    _launch("cudnn_gru[nx=%d]" % nx, lambda: lib.ggnn_cudnn_gru_f32(segs, nx, _ptr(h), _ptr(Wg), _ptr(bg), _ptr(Wcx), _ptr(bcx),
                                                                    _ptr(Wch), _ptr(bch), _ptr(out), _ptr(ws), ws_bytes, V, D, _stream()))
    return out

"""ctypes binding of libggnn_hip.so (the C ABI declared in include/ggnn_hip.h).

The library is built in-tree by build.py (hipcc --offload-arch=gfx950) and sits next to this file.
There is NO fallback: if the shared object is missing or a symbol is absent, importing an op fails
loudly -- the product path never routes through a CPU implementation.
"""
from __future__ import annotations

import ctypes
import os
from ctypes import POINTER, c_char_p, c_float, c_int, c_int32, c_int64, c_size_t, c_uint64, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
# GGNN_LIB_VARIANT=<tag>: load libggnn_hip_<tag>.so (kernel experiments built by tools/variant_lib.sh; never set in production)
LIB_PATH = os.path.join(_HERE, "libggnn_hip%s.so" % ("_" + os.environ["GGNN_LIB_VARIANT"] if os.environ.get("GGNN_LIB_VARIANT") else ""))
ABI_VERSION = 3

# every symbol include/ggnn_hip.h declares: name -> (restype, argtypes)
SYMBOLS = {
    "ggnn_abi_version": (c_int, []),
    "ggnn_matrix_path_is_split": (c_int, []),
    "ggnn_gru_forward_format": (c_int, []),
    "ggnn_gru_form_set": (c_int, [c_int]),
    "ggnn_absmax_f32": (c_int, [POINTER(c_void_p), POINTER(c_int64), c_int, c_void_p, c_void_p]),
    "ggnn_last_error": (c_char_p, []),
    "ggnn_csr_workspace_bytes": (c_size_t, [c_int64, c_int]),
    "ggnn_build_target_csr": (c_int, [c_void_p, POINTER(c_int64), c_int, c_int, c_int64, c_void_p, c_void_p,
                                      c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "ggnn_build_source_csr": (c_int, [c_void_p, POINTER(c_int64), c_int, c_int, c_int64, c_void_p, c_void_p,
                                      c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "ggnn_msg_transform_f32": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p]),
    "ggnn_msg_transform_compact_supported": (c_int, [c_int]),
    "ggnn_compact_workspace_bytes": (c_size_t, [c_int, c_int]),
    "ggnn_build_compact_sources": (c_int, [c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "ggnn_remap_gather_rows": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_void_p]),
    "ggnn_msg_transform_compact_workspace_bytes": (c_size_t, [c_int, c_int]),
    "ggnn_msg_transform_compact_f32": (c_int, [c_void_p, c_void_p, c_void_p, POINTER(c_int64), c_void_p, c_void_p, c_size_t,
                                               c_int, c_int, c_int, c_int, c_void_p]),
    "ggnn_gather_segment_sum_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p,
                                            c_int, c_int, c_int, c_void_p]),
    "ggnn_build_slot_heads": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_void_p]),
    "ggnn_gather_segment_sum_heads_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p,
                                                  c_int, c_int, c_int, c_int, c_void_p]),
    "ggnn_attn_bwd_target_f32": (c_int, [c_void_p] * 11 + [c_int, c_int, c_int, c_int, c_void_p]),
    "ggnn_weighted_segment_sum_f32": (c_int, [c_void_p] * 6 + [c_int, c_int, c_int, c_void_p]),
    "ggnn_range_sum_f32": (c_int, [c_void_p, POINTER(c_int64), c_int, c_void_p, c_void_p]),
    "ggnn_unsorted_segment_sum_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int, c_int, c_void_p]),
    "ggnn_gated_readout_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int,
                                       c_int, c_int, c_void_p]),
    "ggnn_readout_workspace_bytes": (c_size_t, [c_int, c_int, c_int]),
    "ggnn_readout_loss_fwd_f32": (c_int, [c_void_p] * 15 + [c_void_p, c_size_t, c_int, c_int, c_int, c_void_p]),
    "ggnn_readout_loss_bwd_f32": (c_int, [c_void_p] * 14 + [c_int] + [c_void_p] * 4 + [c_void_p, c_size_t, c_int, c_int, c_int, c_void_p]),
    "ggnn_gru_workspace_bytes": (c_size_t, [c_int, c_int]),
    "ggnn_gru_f32": (c_int, [POINTER(c_void_p), c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                             c_void_p, c_size_t, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p]),
    "ggnn_gather_segment_sum_attn_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int,
                                                 c_void_p, c_int, c_int, c_int, c_void_p]),
    "ggnn_rnn_f32": (c_int, [POINTER(c_void_p), c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p]),
    "ggnn_cudnn_gru_workspace_bytes": (c_size_t, [c_int, c_int]),
    "ggnn_cudnn_gru_f32": (c_int, [POINTER(c_void_p), c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                   c_void_p, c_void_p, c_void_p, c_size_t, c_int, c_int, c_void_p]),
    "ggnn_gru_is_fused": (c_int, [c_int]),
    "ggnn_gru_packed_bytes": (c_size_t, [c_int, c_int]),
    "ggnn_gru_pack_weights_f32": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p]),
    "ggnn_gru_packed_f32": (c_int, [POINTER(c_void_p), c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                    c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p]),
    "ggnn_edge_weights_pack_f32": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_void_p]),
    "ggnn_gru_packed_gather_f32": (c_int, [POINTER(c_void_p), c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                           c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p]),
    "ggnn_gru_packed_gather_train_f32": (c_int, [POINTER(c_void_p), c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                                 c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p,
                                                 c_int, c_int, c_int, c_int, c_void_p, c_void_p]),
    "ggnn_gru_gates_f32": (c_int, [POINTER(c_void_p), c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                   c_int, c_int, c_void_p]),
    "ggnn_gru_candidate_f32": (c_int, [POINTER(c_void_p), c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                       c_void_p, c_void_p, c_int, c_int, c_int, c_void_p]),
    "ggnn_sparse_propagate_workspace_bytes": (c_size_t, [c_int, c_int, c_int, c_int64]),
    "ggnn_sparse_propagate_f32": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, POINTER(c_int64),
                                          c_void_p, c_int, c_int, POINTER(c_int32), POINTER(c_int32), POINTER(c_int32),
                                          POINTER(c_void_p), POINTER(c_void_p), POINTER(c_void_p), POINTER(c_void_p),
                                          POINTER(c_void_p), POINTER(c_void_p), POINTER(c_void_p), POINTER(c_void_p), POINTER(c_int32),
                                          POINTER(c_int32), c_int, c_int, POINTER(c_void_p), c_void_p, c_size_t, c_void_p]),
    "ggnn_gru_bwd_stage1_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p,
                                        c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    "ggnn_gru_bwd_stage2_f32": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p]),
    "ggnn_dense_aggregate_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    "ggnn_gemm_f32": (c_int, [POINTER(c_void_p), c_int, c_int, c_int, c_void_p, c_int, c_void_p, c_int, c_int, c_int,
                              c_void_p]),
    "ggnn_xty_workspace_bytes": (c_size_t, [c_int, c_int, c_int, c_int]),
    "ggnn_xty_f32": (c_int, [POINTER(c_void_p), c_int, c_int, POINTER(c_int32), c_void_p, c_void_p, c_int, c_void_p, c_int, c_int,
                             c_int, POINTER(c_int32), c_int, c_void_p, c_size_t, c_void_p]),
    "ggnn_dense_propagate_supported": (c_int, [c_int, c_int, c_int]),
    "ggnn_dense_propagate_is_split": (c_int, [c_int, c_int, c_int]),
    "ggnn_dense_edge_packed_bytes": (c_size_t, [c_int, c_int]),
    "ggnn_dense_edge_pack_f32": (c_int, [c_void_p, c_int, c_int, c_void_p, c_void_p]),
    "ggnn_dense_gru_packed_bytes": (c_size_t, [c_int]),
    "ggnn_dense_gru_pack_f32": (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_void_p]),
    "ggnn_dense_propagate_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int,
                                         c_int, c_int, c_int, c_int, c_void_p]),
    "ggnn_assemble_batch_backward": (c_int, [POINTER(c_void_p), POINTER(c_void_p), c_int, c_int, POINTER(c_int64), POINTER(c_int64), c_void_p,
                                             c_void_p, c_int, c_int, c_int, c_int, c_int, POINTER(c_int64), POINTER(c_int64),
                                             POINTER(c_void_p), c_void_p]),
    "ggnn_assemble_batch": (c_int, [POINTER(c_void_p), c_int, c_int, POINTER(c_int64), POINTER(c_int64), c_void_p, c_int, c_int, c_int,
                                    c_int, c_int, POINTER(c_int64), POINTER(c_int64), POINTER(c_void_p), c_void_p]),
    "ggnn_xty_acc_f32": (c_int, [POINTER(c_void_p), c_int, c_int, POINTER(c_int32), c_void_p, c_void_p, c_int, c_void_p, c_void_p,
                                 c_int, c_int, c_int, c_int, POINTER(c_int32), c_int, c_void_p, c_size_t, c_void_p]),
    "ggnn_colsum_workspace_bytes": (c_size_t, [c_int]),
    "ggnn_colsum_f32": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_size_t, c_void_p]),
    "ggnn_gru_bwd_dx_cand_f32": (c_int, [c_void_p] * 7 + [c_int, c_int, c_int, c_void_p]),
    "ggnn_gru_bwd_dx_gates_f32": (c_int, [c_void_p] * 5 + [c_int, c_int, c_void_p, c_int, c_int, c_int, c_void_p]),
    "ggnn_gather_segment_sum_acc_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p]),
    "ggnn_bwd_dx_f32": (c_int, [c_void_p, c_int, c_int, c_void_p, c_int, c_void_p, c_int, c_int, c_void_p, c_void_p, c_int, c_int,
                                c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    "ggnn_act_bwd_f32": (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_int, c_int, c_void_p]),
    "ggnn_cudnn_gru_train_f32": (c_int, [POINTER(c_void_p), c_int] + [c_void_p] * 9 + [c_void_p, c_size_t, c_int, c_int, c_void_p]),
    "ggnn_cudnn_gru_bwd_stage_f32": (c_int, [c_void_p] * 10 + [c_int, c_int, c_void_p]),
    "ggnn_gru_bwd_is_fused": (c_int, [c_int]),
    "ggnn_gru_bwd_packed_bytes": (c_size_t, [c_int, c_int]),
    "ggnn_gru_bwd_fused_f32": (c_int, [c_void_p] * 12 + [POINTER(c_void_p), c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "ggnn_gru_bwd_fused_gather_f32": (c_int, [c_void_p] * 12 + [POINTER(c_void_p), c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "ggnn_optim_block_floats": (c_int, []),
    "ggnn_clip_adam_f32": (c_int, [c_void_p] * 9 + [c_int, c_float, c_float, c_float, c_float, c_float, c_void_p]),
    "ggnn_probe_mfma_workspace_bytes": (c_size_t, []),
    "ggnn_probe_mfma_rate": (c_int, [c_int, c_int, c_void_p, c_size_t, POINTER(ctypes.c_double), POINTER(ctypes.c_double), c_void_p]),
    "ggnn_gemm_tn_workspace_bytes": (c_size_t, [c_int, c_int, c_int]),
    "ggnn_gemm_tn_f32": (c_int, [c_void_p, c_int, c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_void_p, c_size_t, c_void_p]),
    "ggnn_pack_batch_tables": (c_int, [c_void_p, c_int, c_int, c_void_p, c_int, c_void_p, c_void_p, c_int, c_void_p, c_int, c_void_p,
                                       c_void_p, c_void_p, c_void_p]),
    "ggnn_sparse_train_prepare_f32": (c_int, [c_int, c_int, c_int, POINTER(c_int32), POINTER(c_void_p), c_float, POINTER(c_uint64),
                                              POINTER(c_void_p), POINTER(c_void_p), POINTER(c_int32), POINTER(c_void_p), POINTER(c_void_p),
                                              POINTER(c_void_p), POINTER(c_void_p), c_void_p]),
    "ggnn_sparse_train_workspace_bytes": (c_size_t, [c_int, c_int, c_int, c_int64, c_int]),
    "ggnn_sparse_train_forward_f32": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, POINTER(c_int64), c_void_p,
                                              c_int, c_int, POINTER(c_int32), POINTER(c_int32), POINTER(c_int32), POINTER(c_void_p),
                                              POINTER(c_void_p), POINTER(c_void_p), POINTER(c_void_p), POINTER(c_int32), c_int, c_void_p,
                                              c_size_t, POINTER(c_int64), c_void_p]),
    "ggnn_sparse_train_backward_f32": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, POINTER(c_int64), c_void_p, c_int, c_int,
                                               POINTER(c_int32), POINTER(c_int32), POINTER(c_int32), c_void_p, c_void_p, c_void_p,
                                               c_void_p, c_void_p, c_void_p, c_void_p, POINTER(c_void_p), POINTER(c_void_p), c_int,
                                               POINTER(c_void_p), POINTER(c_void_p), POINTER(c_void_p), POINTER(c_void_p),
                                               POINTER(c_void_p), c_void_p, POINTER(c_void_p), c_void_p, c_size_t, c_void_p, c_void_p]),
    "ggnn_dropout_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_uint64, c_float, c_int64, c_int, c_void_p]),
}

_lib = None


class GGNNError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__("libggnn_hip error %d: %s" % (code, msg))
        self.code = code


def load() -> ctypes.CDLL:
    """Load the shared library once and bind every declared symbol."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            "libggnn_hip.so not found at %s -- run `python __graft_entry__.py build` (hipcc, gfx950). "
            "There is no CPU fallback for the GGNN hot path." % LIB_PATH)
    lib = ctypes.CDLL(LIB_PATH)
    for name, (restype, argtypes) in SYMBOLS.items():
        fn = getattr(lib, name)        # AttributeError if the symbol is missing: loud by design
        fn.restype = restype
        fn.argtypes = argtypes
    ver = lib.ggnn_abi_version()
    if ver != ABI_VERSION:
        raise ImportError("libggnn_hip.so ABI version %d != expected %d; rebuild" % (ver, ABI_VERSION))
    _lib = lib
    return lib


def check(rc: int) -> None:
    if rc != 0:
        msg = load().ggnn_last_error()
        raise GGNNError(rc, msg.decode("utf-8", "replace") if msg else "")

"""CPU tests of the boundary and the host logic: the C-ABI library loads and exports every symbol the
header declares (no compute without a GPU), argument validation returns error codes instead of
faulting, the vectorised packer reproduces the reference packer's feed layout, and the op layer refuses
CPU tensors (there is no CPU product path)."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

from conftest import ROOT


def header_symbols():
    text = open(os.path.join(ROOT, "include", "ggnn_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(ggnn_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol(pkg):
    lib = pkg._lib.load()
    syms = header_symbols()
    assert len(syms) >= 10
    for s in syms:
        assert hasattr(lib, s), "libggnn_hip.so lacks %s" % s
    assert set(syms) == set(pkg._lib.SYMBOLS), "ctypes table and header disagree"
    assert lib.ggnn_abi_version() == pkg._lib.ABI_VERSION == 3


def test_argument_validation_without_gpu(pkg):
    lib = pkg._lib.load()
    # unsupported hidden size / null pointers -> error codes + message, never a crash
    assert lib.ggnn_msg_transform_f32(None, 100, None, None, 10, 100, 4, None) == -1
    assert b"null" in lib.ggnn_last_error()
    assert lib.ggnn_msg_transform_f32(None, 6, None, None, 10, 6, 4, None) == -1      # D % 4 != 0
    assert lib.ggnn_msg_transform_f32(None, 100, None, None, 0, 100, 4, None) == 0    # V == 0 is a no-op
    buf = (ctypes.c_float * 64)()
    p = ctypes.cast(buf, ctypes.c_void_p)
    aligned = ctypes.c_void_p((p.value + 15) // 16 * 16)
    assert lib.ggnn_msg_transform_f32(aligned, 12, aligned, aligned, 1, 12, 1, None) == -2   # D=12 unsupported
    assert b"unsupported" in lib.ggnn_last_error()
    off = (ctypes.c_int64 * 2)(0, 5)
    assert lib.ggnn_build_target_csr(None, off, 1, 10, 7, None, None, None, None, None, 0, None) == -1  # type_off[T] != M
    assert lib.ggnn_gru_workspace_bytes(1000, 100) >= 2 * 1000 * 100 * 4      # un-fused scratch (+ packed weight images)
    assert lib.ggnn_gru_workspace_bytes(1000, 200) == 2 * 1000 * 200 * 4      # no fused path at D=200: r*h and u only
    assert lib.ggnn_gru_is_fused(100) == 1 and lib.ggnn_gru_is_fused(256) == 2 and lib.ggnn_gru_is_fused(200) == 0
    # column-panel images of the three gates: f32 (4 bytes per weight) or, under the split matrix path, room for either operand format
    # of the GRU forward -- two f16 planes (4) or three bf16 planes (6) -- because the format is chosen per pack / launch (ABI 3)
    fmt = lib.ggnn_gru_forward_format()                      # (process default of the HOST policy: formats.py)
    assert fmt in ((2, 3) if lib.ggnn_matrix_path_is_split() else (0,))
    assert lib.ggnn_gru_packed_bytes(256, 1) == 3 * 2 * 256 * 256 * (6 if lib.ggnn_matrix_path_is_split() else 4)
    # an unknown operand format is an argument error, not a silent default
    assert lib.ggnn_gru_pack_weights_f32(aligned, aligned, 1, 100, 7, aligned, None) == -1 and b"gru_fmt" in lib.ggnn_last_error()
    assert lib.ggnn_msg_transform_compact_supported(256) == 1 and lib.ggnn_msg_transform_compact_supported(200) == 0
    assert lib.ggnn_csr_workspace_bytes(0, 10) > 0


def test_ops_refuse_cpu_tensors(pkg):
    h = torch.zeros(4, 100)
    W = torch.zeros(4, 100, 100)
    with pytest.raises(TypeError):
        pkg.ops.msg_transform(h, W)
    with pytest.raises(TypeError):
        pkg.ops.build_message_index([torch.zeros(0, 2, dtype=torch.int32)], 4)


def test_vectorised_packer_equals_reference_packer(pkg, oracle):
    ms = pkg.synthetic_qm9(120, mean_nodes=11, seed=4)
    raw = ms.to_json()
    for tie, T in ((True, 4), (False, 8)):
        ref = oracle.pack_batch(raw, T, 100, tie)
        mine = pkg.data.pack_batch(ms, np.arange(ms.num_graphs), T, 100, tie)
        for k in ("initial_node_representation", "num_incoming_edges_per_type", "graph_nodes_list", "target_values",
                  "target_mask"):
            assert np.array_equal(np.asarray(ref[k], np.float64), np.asarray(getattr(mine, k), np.float64)), k
        for a, b in zip(ref["adjacency_lists"], mine.adjacency_lists):
            assert a.dtype == b.dtype == np.int32 and np.array_equal(a, b)
    back = pkg.MoleculeSet.from_json(raw)
    assert np.array_equal(back.bonds, ms.bonds) and np.array_equal(back.node_ptr, ms.node_ptr)


def test_batch_boundaries_follow_strict_less_than(pkg):
    """chem_tensorflow_sparse.py:297: take graphs while node_offset + n < batch_size."""
    n = np.array([4, 4, 4, 4, 4, 4])
    assert pkg.data.batch_boundaries(n, 9) == [0, 2, 4, 6]       # 4+4 = 8 < 9, adding 4 more -> 12 (no)
    assert pkg.data.batch_boundaries(n, 8) == [0, 1, 2, 3, 4, 5, 6]   # 4+4 = 8 is NOT < 8
    assert pkg.data.batch_boundaries(n, 13) == [0, 3, 6]
    with pytest.raises(ValueError):
        pkg.data.batch_boundaries(np.array([10]), 10)
    ms = pkg.synthetic_qm9(300, mean_nodes=15, seed=0)
    p = pkg.SparseGGNNChemModel.default_params(); p["batch_size"] = 500
    batches = pkg.pack_batches(ms, p, 4)
    assert sum(b.num_graphs for b in batches) == 300
    assert all(b.num_nodes < 500 for b in batches)
    # data-parallel sharding with dp_balance_nodes off: same global batches, dealt round-robin, equal step counts, empty padding
    p["dp_balance_nodes"] = False
    shards = [pkg.pack_batches(ms, p, 4, rank=r, world_size=4) for r in range(4)]
    assert len({len(s) for s in shards}) == 1
    dealt = [b for i in range(len(shards[0])) for s in shards for b in [s[i]] if b.num_graphs]
    assert [b.num_nodes for b in dealt] == [b.num_nodes for b in batches]
    # default (balanced): the epoch re-cut into a multiple of 4 equal-node batches, still every graph once and below batch_size
    p["dp_balance_nodes"] = True
    shards = [pkg.pack_batches(ms, p, 4, rank=r, world_size=4) for r in range(4)]
    assert len({len(s) for s in shards}) == 1 and all(b.num_graphs > 0 and b.num_nodes < 500 for s in shards for b in s)
    assert sum(b.num_graphs for s in shards for b in s) == 300


def test_synthetic_molecules_are_qm9_shaped(pkg):
    ms = pkg.synthetic_qm9(5000, mean_nodes=18, seed=0)
    n = ms.nodes_per_graph()
    assert n.min() >= 3 and n.max() <= 29 and 17 < n.mean() < 19
    assert ms.num_fwd_edge_types == 4 and ms.annotation_size == 5
    assert np.all(ms.node_feat.sum(1) == 1)
    nb = np.diff(ms.bond_ptr)
    assert np.all(nb >= n - 1) and np.all(nb <= n + 1)            # spanning tree + 0..2 ring closures
    for g in (0, 1, 4999):                                         # bonds stay inside their molecule, no self loops
        b = ms.bonds[ms.bond_ptr[g]:ms.bond_ptr[g + 1]]
        assert b[:, [0, 2]].max() < n[g] and np.all(b[:, 0] != b[:, 2])

"""Device-side batch packer: the tensorised twin of data.pack_batch.

The reference packs every minibatch in pure Python on the host (chem_tensorflow_sparse.py:278-350); data.pack_batch
is its vectorised NumPy twin (~14 ms per 100k-node batch, plus the upload).  With the propagation at ~1.4 ms per
batch the packer -- the step immediately before the path -- would cap end-to-end throughput, so here the whole
dataset is uploaded ONCE in structure-of-arrays form and a batch is assembled on the GPU from the graph ids alone:
index arithmetic, one key sort, two bincounts.  Same outputs as data.pack_batch, bit for bit (tests/test_gpu_parity.py).

Only the batch boundaries (a greedy scan over at most G node counts, data.batch_boundaries) and the per-type message and
source-pair counts (sums over per-molecule tables, DeviceMoleculeSet.type_counts) are computed on the host; nothing is read
back from the device, so packing a batch never waits for the GPU.
"""
from __future__ import annotations

import ctypes
import os
from typing import Any, Dict, List, Optional, Sequence

import numpy as np
import torch

from . import _lib, formats, ops
from .data import MoleculeSet, batch_boundaries, epoch_boundaries

# Batches gathered from dataset-level tables (ggnn_assemble_batch) instead of being sorted and scanned one by one;
# GGNN_PACK_STATIC=0 keeps the general per-batch builders (same outputs, bit for bit: tests/test_gpu_parity.py).
USE_STATIC_TABLES = os.environ.get("GGNN_PACK_STATIC", "1") != "0"


class DeviceMoleculeSet:
    """data.MoleculeSet resident in HBM (QM9: 134k molecules = 2.4M atoms -> 60 MB)."""

    def __init__(self, ms: MoleculeSet, device, label_mask: Optional[np.ndarray] = None):
        t = lambda a, dt: torch.from_numpy(np.ascontiguousarray(a)).to(device=device, dtype=dt)
        self.host = ms
        self.device = torch.device(device)
        self.node_ptr = t(ms.node_ptr, torch.int64)
        self.node_feat = t(ms.node_feat, torch.float32)
        # max |h0| of any batch of this dataset (formats.py: operand bound of the two-piece f16 GRU format; NaN stays NaN)
        self.node_feat_absmax = float(np.abs(np.asarray(ms.node_feat, dtype=np.float32)).max()) if np.size(ms.node_feat) else 0.0
        self.bond_ptr = t(ms.bond_ptr, torch.int64)
        self.bonds = t(ms.bonds, torch.int64)                  # (src_local, bond_type 1..F, dst_local)
        self.targets = t(ms.targets, torch.float32)
        self.label_mask = None if label_mask is None else t(label_mask, torch.float32)
        self.nodes_per_graph = np.diff(ms.node_ptr)            # host copies of the sizes: no sync to size outputs
        self.bonds_per_graph = np.diff(ms.bond_ptr)
        self.max_bond_type = int(ms.bonds[:, 1].max()) if len(ms.bonds) else 0
        self.min_bond_type = int(ms.bonds[:, 1].min()) if len(ms.bonds) else 1
        self._identity_epoch = {}                              # batch_size -> (order, batch boundaries, order on the device) of the dataset order
        self._type_counts = {}                                 # (T, tie) -> per-graph message / source-pair counts per type
        self._static = {}                                      # (T, tie, compact) -> dataset-level tables of ggnn_assemble_batch
        # Bond endpoints come from the data file: check them ONCE against their graph's node count, so that the per-batch
        # index build can skip validation (pack_batch_device offsets them into the batch; a bad id would otherwise
        # become an out-of-bounds gather on the GPU).
        if len(ms.bonds):
            n_of_bond = np.repeat(self.nodes_per_graph, self.bonds_per_graph)
            ends = ms.bonds[:, [0, 2]].astype(np.int64)
            if (ends < 0).any() or (ends >= n_of_bond[:, None]).any():
                bad = int(np.nonzero(((ends < 0) | (ends >= n_of_bond[:, None])).any(axis=1))[0][0])
                g = int(np.searchsorted(ms.bond_ptr, bad, side='right') - 1)
                raise IndexError("bond %d of graph %d mentions node %s outside [0, %d)" % (
                    bad - int(ms.bond_ptr[g]), g, ends[bad].tolist(), int(self.nodes_per_graph[g])))


    def type_counts(self, num_edge_types: int, tie_fwd_bkwd: bool):
        """Host tables [G, T]: directed messages of type t in graph g (chem_tensorflow_sparse.py:259-263: every bond gives the
        forward edge of type t and the backward edge of type t, or t + T/2 when the directions are untied), and distinct
        (source node, type) pairs among them.  Both are properties of a molecule, so the per-type sizes of a BATCH -- the
        adjacency-list lengths and the row ranges of the compacted message transform, which the host needs as launch
        arguments -- are sums over its graph ids: no device->host read per batch."""
        key = (int(num_edge_types), bool(tie_fwd_bkwd))
        hit = self._type_counts.get(key)
        if hit is None:
            ms = self.host
            T = key[0]
            F = T if key[1] else T // 2
            G = ms.num_graphs
            B = len(ms.bonds)
            g_of_bond = np.repeat(np.arange(G, dtype=np.int64), self.bonds_per_graph)
            typ = ms.bonds[:, 1].astype(np.int64) - 1
            valid = (typ >= 0) & (typ < F)
            msgs = np.zeros((G, T), dtype=np.int64)
            pairs = np.zeros((G, T), dtype=np.int64)
            if B:
                node0 = np.asarray(ms.node_ptr, dtype=np.int64)[g_of_bond]
                src_f = node0 + ms.bonds[:, 0]; src_b = node0 + ms.bonds[:, 2]           # global node ids
                t_f = typ; t_b = typ if key[1] else typ + F
                for srcs, ts in ((src_f, t_f), (src_b, t_b)):
                    np.add.at(msgs, (g_of_bond[valid], ts[valid]), 1)
                NN = int(ms.node_ptr[-1]) + 1
                keys = np.unique(np.concatenate([(src_f * T + t_f)[valid], (src_b * T + t_b)[valid]]))      # (node, type) pairs
                pn, pt = keys // T, keys % T
                pg = np.searchsorted(np.asarray(ms.node_ptr, dtype=np.int64), pn, side='right') - 1
                np.add.at(pairs, (pg, pt), 1)
            hit = self._type_counts[key] = (msgs, pairs)
        return hit


    def upload_order(self, order: np.ndarray) -> torch.Tensor:
        """An epoch's graph order on the device (int64 [G]).  The identity order (validation, inference) is cached; a shuffled
        order goes through a pinned staging buffer on a copy stream of its own -- a pageable
        `torch.from_numpy(order).to(device)` on the packing stream is a staged synchronous copy behind everything queued there and
        held the launching thread for ~2 ms at every epoch start while the queues ran dry (tools/e2e_trace.py)."""
        G = int(order.shape[0])
        if G == self.host.num_graphs and G and order[0] == 0 and order[-1] == G - 1 and bool((np.diff(order) == 1).all()):
            ident = getattr(self, "_identity_order", None)
            if ident is None:
                ident = self._identity_order = self._complete(torch.arange(G, dtype=torch.int64, device=self.device))
            return ident
        if self.device.type != "cuda":
            return torch.from_numpy(order).to(self.device)
        stage = getattr(self, "_order_stage", None)
        if stage is None or stage[0].numel() < G:
            stage = self._order_stage = (torch.empty(max(G, 1), dtype=torch.int64, pin_memory=True), torch.cuda.Stream(self.device))
        pinned, up = stage
        up.synchronize()                                 # (the previous epoch's copy out of the staging buffer is long done; cheap)
        pinned[:G].copy_(torch.from_numpy(order))
        # On a stream of its own, so that waiting for the copy waits for nothing else; complete on return (any stream may read it).
        # The block is ALLOCATED under that stream too: the caching allocator then hands out a block of that stream's pool -- one
        # whose earlier uses were on `up` or were recorded -- never one that kernels still queued on the packing stream read
        # (round-3 advisor finding).  Readers on other streams record their use (pack_batches_device).
        with torch.cuda.stream(up):
            dev_order = torch.empty(G, dtype=torch.int64, device=self.device)
            dev_order.copy_(pinned[:G], non_blocking=True)
        up.synchronize()
        return dev_order

    def _complete(self, t: torch.Tensor) -> torch.Tensor:
        """A cached, never-freed device tensor that ANY stream may read without an event: the stream that fills it is drained once,
        when the tensor is created (the packing streams read these caches with no ordering behind the creating stream)."""
        if t.is_cuda:
            torch.cuda.current_stream(t.device).synchronize()
        return t

    def arange_i32(self, n: int) -> torch.Tensor:
        """arange(n) int32 on the device, cut from one cached ramp (the identity row list of the backward's compacted transform:
        one launch less per training batch)."""
        ramp = getattr(self, "_ramp", None)
        if ramp is None or ramp.numel() < n:
            ramp = self._ramp = self._complete(torch.arange(max(n, 1 << 18), dtype=torch.int32, device=self.device))
        return ramp[:n]

    def task_ids_dev(self, task_ids) -> torch.Tensor:
        key = tuple(int(t) for t in task_ids)
        if getattr(self, "_tids", (None, None))[0] != key:
            self._tids = (key, self._complete(torch.as_tensor(list(key), dtype=torch.int64, device=self.device)))
        return self._tids[1]

    def static_backward_tables(self, num_edge_types: int, tie_fwd_bkwd: bool):
        """The same for the backward pass's transpose structures (by-source CSR, ops.CompactBackward), built lazily on the first
        training batch."""
        tab = self.static_tables(num_edge_types, tie_fwd_bkwd, True)
        if tab is None or tab["slot_crow"] is None:
            return None
        if "bwd_ptrs" not in tab:
            idx, comp = tab["_index"], tab["_comp"]
            bwd = ops.compact_backward(idx, comp)
            src = idx._source_index
            keep = [src.row_ptr, src.gather_row, src.msg_perm, bwd.rows_index.row_ptr, bwd.rows_index.gather_row, bwd.rows_index.msg,
                    bwd.node_index.row_ptr, bwd.node_index.gather_row]
            tab["bwd_keep"] = keep
            tab["bwd_ptrs"] = (ctypes.c_void_p * 8)(*[t.data_ptr() for t in keep])
        return tab

    def static_tables(self, num_edge_types: int, tie_fwd_bkwd: bool, compact: bool):
        """Dataset-level tables of ggnn_assemble_batch: the general builders run ONCE over all graphs taken as one batch (in
        dataset order); every later batch is gathered from their outputs.  None when the whole dataset does not fit the
        builders' 32-bit indices (then every batch is built on its own, as before)."""
        key = (int(num_edge_types), bool(tie_fwd_bkwd), bool(compact))
        if key in self._static:
            return self._static[key]
        T = key[0]
        ms = self.host
        Nd = int(ms.node_ptr[-1])
        msgs_gt, pairs_gt = self.type_counts(T, key[1])
        Md = int(msgs_gt.sum())
        tab = None
        if ms.num_graphs and Md and Nd * T < 2 ** 31 - 1 and Md < 2 ** 31 - 1 and T <= 16:
            A = self.node_feat.shape[1]
            full = pack_batch_device(self, np.arange(ms.num_graphs, dtype=np.int64), T, max(A, 4 * ((A + 3) // 4)), key[1], (0,),
                                     compact=False, training=False, static=False)
            idx = full['message_index']
            dev = self.device
            i32 = lambda a: torch.from_numpy(np.ascontiguousarray(a).astype(np.int32)).to(dev)
            excl = lambda c: np.cumsum(c, axis=0) - c
            tab = {"A": A, "T": T, "node_ptr": i32(ms.node_ptr), "feat": self.node_feat.contiguous(),
                   "nin": full['num_incoming_edges_per_type'].contiguous(), "row_ptr": idx.row_ptr, "adj": idx.adj,
                   "slot_gather": idx.gather_row, "slot_msg": idx.msg_perm, "slot_crow": None, "pair_node": None,
                   "e_off": i32(excl(msgs_gt)), "p_off": None, "type_off": [int(x) for x in idx.type_off], "type_row_off": [0] * (T + 1),
                   "msgs_gt": msgs_gt, "pairs_gt": pairs_gt, "tie": key[1],
                   # [Gd, 2 + 2T + 1] per graph: nodes, message slots, messages per type, compact rows per type, compact rows
                   "counts_dev_t": i32(np.concatenate([self.nodes_per_graph[:, None], msgs_gt.sum(axis=1, keepdims=True), msgs_gt,
                                                       pairs_gt if compact else np.zeros_like(pairs_gt),
                                                       (pairs_gt if compact else np.zeros_like(pairs_gt)).sum(axis=1, keepdims=True)], axis=1).T)}
            if compact:
                comp = ops.build_compact_sources(idx)
                tab.update(slot_crow=comp.gather_row, pair_node=comp.pair_node, p_off=i32(excl(pairs_gt)),
                           type_row_off=[int(x) for x in comp.type_row_off], _index=idx, _comp=comp)
                want = [0] + [int(x) for x in np.cumsum(pairs_gt.sum(axis=0))]
                if tab["type_row_off"] != want:
                    raise AssertionError("per-molecule pair counts %s disagree with the device's compaction %s" % (want, tab["type_row_off"]))
            if tab["type_off"] != [0] + [int(x) for x in np.cumsum(msgs_gt.sum(axis=0))]:
                raise AssertionError("per-molecule message counts disagree with the packed dataset")
            ptrs = [tab[k] for k in ("node_ptr", "feat", "nin", "row_ptr", "adj", "slot_gather", "slot_msg", "slot_crow", "pair_node", "e_off", "p_off")]
            tab["ptrs"] = (ctypes.c_void_p * 11)(*[None if t is None else t.data_ptr() for t in ptrs])
            tab["c_type_off"] = (ctypes.c_int64 * (T + 1))(*tab["type_off"])
            tab["c_type_row_off"] = (ctypes.c_int64 * (T + 1))(*tab["type_row_off"])
        self._static[key] = tab
        return tab


def _assemble_from_tables(dms: DeviceMoleculeSet, tab: dict, gids_h: np.ndarray, hidden_size: int, training: bool = False,
                          gids_dev: Optional[torch.Tensor] = None, task_ids: Sequence[int] = (0,)):
    """h0, graph_nodes_list, graph_ptr, nin and the batch's MessageIndex (+ compacted sources, slot heads) gathered from the
    dataset-level tables: the (graph, type) prefix sums come from per-molecule count tables, and ggnn_assemble_batch does the rest
    in ONE launch.  With the graph ids on the device (an epoch's order is uploaded once) the prefix sums and the batch's labels are
    formed there too (ggnn_pack_batch_tables, one launch): a batch is two dependent launches (+ one for the backward structures).
    Returns (h0, gnl, graph_ptr, nin, index, type_off, labels) with labels = (target_values, target_mask) or None (host path)."""
    lib = _lib.load()
    dev = dms.device
    T, A = tab["T"], tab["A"]
    compact = tab["slot_crow"] is not None
    G = len(gids_h)
    n = dms.nodes_per_graph[gids_h].astype(np.int64)
    mg = tab["msgs_gt"][gids_h].reshape(G, T)
    pg = tab["pairs_gt"][gids_h].reshape(G, T) if compact else np.zeros((G, T), np.int64)
    if gids_dev is not None:
        V, M = int(n.sum()), int(mg.sum())
        tsum, psum = mg.sum(axis=0), pg.sum(axis=0)
        R = int(psum.sum())
        type_off = [0] + [int(x) for x in np.cumsum(tsum)]
        type_row_off = [0] + [int(x) for x in np.cumsum(psum)]
    else:
        incl = lambda c: np.concatenate([np.zeros((1,) + c.shape[1:], np.int64), np.cumsum(c, axis=0)])
        node_off, slot_off, msg_off, pair_off = incl(n), incl(mg.sum(axis=1)), incl(mg), incl(pg)
        V, M, R = int(node_off[-1]), int(slot_off[-1]), int(pair_off[-1].sum())
        type_off = [0] + [int(x) for x in np.cumsum(msg_off[-1])]
        type_row_off = [0] + [int(x) for x in np.cumsum(pair_off[-1])]
    if V * T >= 2 ** 31 - 1 or V * hidden_size >= 2 ** 31 - 1:
        raise ValueError("batch too large for 32-bit indices")
    st = torch.cuda.current_stream().cuda_stream
    labels = None
    if gids_dev is not None:
        # the graph ids are on the device already (pack_batches_device uploads an epoch's order once): the prefix sums are formed
        # there too -- a host->device copy per batch would make the host wait for everything queued on the stream, i.e. for
        # the previous batch's whole forward pass (tools/h2d_probe.py)
        ct = tab["counts_dev_t"]                                                # [2 + 2T + 1, Gd]
        rows = ct.shape[0]
        bt = torch.empty(G + rows * (G + 1), dtype=torch.int32, device=dev)
        tids = dms.task_ids_dev(task_ids)
        K = int(tids.numel())
        tv = torch.empty((K, G), dtype=torch.float32, device=dev)
        tm = torch.empty((K, G), dtype=torch.float32, device=dev)
        _lib.check(lib.ggnn_pack_batch_tables(ct.data_ptr(), ct.shape[1], rows, gids_dev.data_ptr(), G, dms.targets.data_ptr(),
                                              None if dms.label_mask is None else dms.label_mask.data_ptr(), dms.targets.shape[1],
                                              tids.data_ptr(), K, bt.data_ptr(), tv.data_ptr(), tm.data_ptr(), st))
        labels = (tv, tm)
    else:
        batch_tab = np.concatenate([gids_h, node_off, slot_off, msg_off.T.ravel(), pair_off.T.ravel(), pair_off.sum(axis=1)]).astype(np.int32)
        bt = torch.from_numpy(batch_tab).to(dev)
    i32 = lambda *shape: torch.empty(shape, dtype=torch.int32, device=dev)
    h0 = torch.empty((V, hidden_size), dtype=torch.float32, device=dev)
    gnl, graph_ptr, nin = i32(V), i32(G + 1), torch.empty((V, T), dtype=torch.float32, device=dev)
    adj, row_ptr, gather_row, msg_perm = i32(M, 2), i32(V + 1), i32(M), i32(M)
    pair_node, gather_c = (i32(max(R, 1)), i32(M)) if compact else (None, None)
    heads = i32(V, 4) if (ops.USE_SLOT_HEADS and V and M) else None
    outs = [h0, gnl, graph_ptr, nin, adj, row_ptr, gather_row, msg_perm, pair_node, gather_c, heads]
    c_out = (ctypes.c_void_p * 11)(*[None if t is None or t.numel() == 0 else t.data_ptr() for t in outs])
    c_to = (ctypes.c_int64 * (T + 1))(*type_off)
    c_tro = (ctypes.c_int64 * (T + 1))(*type_row_off)
    _lib.check(lib.ggnn_assemble_batch(tab["ptrs"], A, T, tab["c_type_off"], tab["c_type_row_off"], bt.data_ptr(), G, V, M, R,
                                       hidden_size, c_to, c_tro, c_out, st))
    index = ops.MessageIndex(adj, type_off, row_ptr, gather_row, msg_perm, V, T)
    if heads is not None and not compact:
        index._slot_heads = (gather_row, heads)
    if compact and M:
        comp = index._compact = ops.CompactSources(pair_node[:max(R, 1)], type_row_off, gather_c)
        if heads is not None:
            comp._slot_heads = (gather_c, heads)
        btab = dms.static_backward_tables(T, tab["tie"]) if training and R else None
        if btab is not None:
            # the backward's transpose structures, gathered as well (ggnn_assemble_batch_backward: one more launch)
            src_rp, src_g, src_m = i32(V * T + 1), i32(M), i32(M)
            rows_rp, rows_g, rows_m, node_rp, node_order = i32(R + 1), i32(M), i32(M), i32(V + 1), i32(R)
            c_bout = (ctypes.c_void_p * 8)(*[t.data_ptr() for t in (src_rp, src_g, src_m, rows_rp, rows_g, rows_m, node_rp, node_order)])
            _lib.check(lib.ggnn_assemble_batch_backward(tab["ptrs"], btab["bwd_ptrs"], A, T, tab["c_type_off"], tab["c_type_row_off"],
                                                        bt.data_ptr(), gnl.data_ptr(), G, V, M, R, hidden_size, c_to, c_tro, c_bout, st))
            index._source_index = ops.MessageIndex(adj, type_off, src_rp, src_g, src_m, V * T, T)
            bwd = object.__new__(ops.CompactBackward)
            bwd.rows_index = ops.SegmentIndex(rows_rp, rows_g, R, rows_m)
            bwd.source_node_index = ops.SegmentIndex(src_rp[::T].contiguous(), src_g, V, src_m)
            bwd.node_index = ops.SegmentIndex(node_rp, node_order, V)
            bwd.identity = ops.CompactSources(dms.arange_i32(max(R, 1)), type_row_off, gather_c)
            comp._bwd = bwd
    return h0, gnl, graph_ptr, nin, index, type_off, labels


def _ranges(starts: torch.Tensor, lengths: torch.Tensor, total: int) -> torch.Tensor:
    """cat_i arange(starts[i], starts[i] + lengths[i]) on the device; `total` = sum(lengths), known on the host."""
    if total == 0:
        return torch.zeros(0, dtype=torch.int64, device=starts.device)
    ends = torch.cumsum(lengths, 0)
    base = torch.repeat_interleave(starts - (ends - lengths), lengths, output_size=total)
    return base + torch.arange(total, dtype=torch.int64, device=starts.device)


def pack_batch_device(dms: DeviceMoleculeSet, graph_ids: np.ndarray, num_edge_types: int, hidden_size: int,
                      tie_fwd_bkwd: bool = True, task_ids: Sequence[int] = (0,), compact: bool = True,
                      training: bool = False, static: Optional[bool] = None,
                      graph_ids_dev: Optional[torch.Tensor] = None) -> Dict[str, Any]:
    """One batch from graphs `graph_ids` (in this order), assembled on the GPU: the feed dict of
    SparseGGNNChemModel.to_device_batch (chem_tensorflow_sparse.py:254-276, 298-348), message index included."""
    dev = dms.device
    gids_h = np.asarray(graph_ids, dtype=np.int64)
    G = len(gids_h)
    T = int(num_edge_types)
    V = int(dms.nodes_per_graph[gids_h].sum()) if G else 0
    B = int(dms.bonds_per_graph[gids_h].sum()) if G else 0
    A = dms.node_feat.shape[1]
    if A > hidden_size:
        raise ValueError("annotation_size %d exceeds hidden_size %d" % (A, hidden_size))
    F = T if tie_fwd_bkwd else T // 2
    if B and (dms.min_bond_type < 1 or dms.max_bond_type > F):
        raise IndexError("edge type outside [0, num_edge_types)")
    gids = torch.from_numpy(gids_h).to(dev) if graph_ids_dev is None else graph_ids_dev      # (int64, the batch's graphs on the device)
    tids = dms.task_ids_dev(task_ids)

    def labels_torch():
        tv = dms.targets[gids][:, tids].t().contiguous()                            # :335
        tm = torch.ones_like(tv) if dms.label_mask is None else dms.label_mask[gids][:, tids].t().contiguous()
        return tv * tm, tm                                                          # masked labels feed 0. (:319-321)

    want_compact = bool(compact and ops.compact_supported(hidden_size))
    tab = dms.static_tables(T, tie_fwd_bkwd, want_compact) if (USE_STATIC_TABLES if static is None else static) else None
    if tab is not None:
        # gathered from the dataset-level tables: no sort, no scan, two launches (ggnn_pack_batch_tables, ggnn_assemble_batch)
        h0, gnl, graph_ptr, nin, index, type_off, labels = _assemble_from_tables(dms, tab, gids_h, hidden_size, training, graph_ids_dev,
                                                                                 task_ids)
        tv, tm = labels if labels is not None else labels_torch()
        adjacency = [index.adj[type_off[t]:type_off[t + 1]] for t in range(T)]
        return formats.declare_h0_absmax({
            'initial_node_representation': h0, 'adjacency_lists': adjacency, 'num_incoming_edges_per_type': nin,
            'graph_nodes_list': gnl, 'graph_ptr': graph_ptr, 'target_values': tv, 'target_mask': tm, 'num_graphs': G,
            'message_index': ops.prepare_message_index(index, hidden_size, compact, training), 'graph_nodes_sorted': True,
            'graph_ids': gids,
        }, dms.node_feat_absmax)
    tv, tm = labels_torch()
    n = dms.node_ptr[gids + 1] - dms.node_ptr[gids]
    offs = torch.cumsum(n, 0) - n                                                   # node offset of each graph (:297)
    nsel = _ranges(dms.node_ptr[gids], n, V)
    h0 = torch.zeros((V, hidden_size), dtype=torch.float32, device=dev)             # :300-302 zero-pad to D
    if V:
        h0[:, :A] = dms.node_feat[nsel]
    gnl = torch.repeat_interleave(torch.arange(G, dtype=torch.int32, device=dev), n, output_size=V)   # :304
    nb = dms.bond_ptr[gids + 1] - dms.bond_ptr[gids]
    bsel = _ranges(dms.bond_ptr[gids], nb, B)
    bonds = dms.bonds[bsel]
    boff = torch.repeat_interleave(offs, nb, output_size=B)
    src = bonds[:, 0] + boff                                                        # :307 (+ node_offset)
    dst = bonds[:, 2] + boff
    typ = bonds[:, 1] - 1                                                           # :258
    s_all = torch.cat([src, dst]); d_all = torch.cat([dst, src])                    # :259-263 both directions
    t_all = torch.cat([typ, typ]) if tie_fwd_bkwd else torch.cat([typ, typ + F])
    # :265 sorted((src,dst)) per type, graphs in order == one sort on (type, src, dst) of the offset ids
    Vk = max(V, 1)
    key, _ = torch.sort((t_all * Vk + s_all) * Vk + d_all)
    d_sorted = key % Vk
    ts = key // Vk
    s_sorted = ts % Vk
    t_sorted = ts // Vk
    # :310-313 incoming-edge counts per (node, type).  (scatter_add, not bincount: bincount reads max(input) back to the host)
    nin = torch.zeros(V * T, dtype=torch.float32, device=dev).scatter_add_(
        0, d_sorted * T + t_sorted, torch.ones(1, dtype=torch.float32, device=dev).expand(d_sorted.shape[0])).view(V, T)
    adj = torch.stack([s_sorted, d_sorted], dim=1).to(torch.int32)
    msgs_gt, pairs_gt = dms.type_counts(T, tie_fwd_bkwd)
    counts: List[int] = [int(c) for c in msgs_gt[gids_h].sum(axis=0)] if G else [0] * T      # per-type sizes from host tables
    pair_counts = [int(c) for c in pairs_gt[gids_h].sum(axis=0)] if G else [0] * T
    type_row_off = [0]
    for c in pair_counts:
        type_row_off.append(type_row_off[-1] + c)
    adjacency, o = [], 0
    for t in range(T):
        adjacency.append(adj[o:o + counts[t]])                                      # :343-348 (empty types: [0,2])
        o += counts[t]
    return formats.declare_h0_absmax({
        'initial_node_representation': h0,
        'adjacency_lists': adjacency,
        'num_incoming_edges_per_type': nin,
        'graph_nodes_list': gnl,
        'graph_ptr': torch.cat([offs, offs.new_full((1,), V)]).to(torch.int32),       # first node of every graph (+ V)
        'target_values': tv,
        'target_mask': tm,
        'num_graphs': G,
        # (ids are offsets we just built from bond endpoints DeviceMoleculeSet checked: no per-batch validation)
        'message_index': ops.prepare_message_index(ops.build_message_index(adjacency, V, validate=False), hidden_size, compact, training,
                                                   type_row_off=type_row_off),
        'graph_nodes_sorted': True,
        'graph_ids': gids,
    }, dms.node_feat_absmax)


def pack_batches_device(dms: DeviceMoleculeSet, params: dict, num_edge_types: int, order: Optional[np.ndarray] = None,
                        rank: int = 0, world_size: int = 1, compact: bool = True, training: bool = False):
    """Generator over one epoch's batches for graph order `order` -- data.pack_batches on the device (same batch
    boundaries, same rank assignment, same empty padding batches)."""
    ms = dms.host
    G = ms.num_graphs
    if order is None:
        # dataset order (validation, inference): order, batch boundaries and the device copy of the order are the same every epoch
        balance = bool(params.get("dp_balance_nodes", True))
        key = (int(params["batch_size"]), int(world_size), balance)
        cached = dms._identity_epoch.get(key)
        if cached is None:
            ident = np.arange(G, dtype=np.int64)
            cached = dms._identity_epoch[key] = (ident, epoch_boundaries(dms.nodes_per_graph, params["batch_size"], world_size, balance),
                                                 dms.upload_order(ident))
        order, bounds, order_dev = cached
    else:
        order = np.asarray(order, np.int64)
        bounds = epoch_boundaries(dms.nodes_per_graph[order], params["batch_size"], world_size, bool(params.get("dp_balance_nodes", True)))
        order_dev = dms.upload_order(order)                     # ONE upload per epoch: the batches slice it on the device
    nb = len(bounds) - 1
    steps = (nb + world_size - 1) // world_size
    for s in range(steps):
        i = s * world_size + rank
        ids = order[bounds[i]:bounds[i + 1]] if i < nb else np.zeros(0, np.int64)
        ids_dev = order_dev[bounds[i]:bounds[i + 1]] if i < nb else order_dev[:0]
        if order_dev.is_cuda:
            order_dev.record_stream(torch.cuda.current_stream(order_dev.device))    # (allocated on the upload stream, read on this one)
        yield pack_batch_device(dms, ids, num_edge_types, params["hidden_size"], params.get("tie_fwd_bkwd", True),
                                params.get("task_ids", [0]), compact, training, graph_ids_dev=ids_dev)

"""Molecule-graph data handling for the sparse/dense GGNN hot path.

Two jobs, both on the host, both off the timed path:

1. `MoleculeSet` -- a tensorised (structure-of-arrays) container for a list of graphs in the
   reference's on-disk schema ``{'targets': [[y],..], 'graph': [[src, bond, dst],..],
   'node_features': [[one-hot],..]}`` (reference get_data.py:82-86), plus a vectorised generator of
   synthetic QM9-shaped molecules (there is no network / no RDKit / no QM9 here).

2. `pack_batches` -- the vectorised twin of the reference's pure-Python packer
   (chem_tensorflow_sparse.py:254-350): graphs are concatenated into one disconnected super-graph
   while ``node_offset + n < batch_size`` (strict, :297); with `tie_fwd_bkwd` every bond yields both
   directions under the same type (:259-263); per type the list is sorted by (src,dst) per graph and
   graphs are concatenated in order (:265, :307, :345) -- which is a global lexsort on the offset
   node ids; `num_incoming_edges_per_type` counts incoming edges per (node,type) (:260-263, :310-313);
   node features are zero-padded to `hidden_size` (:300-302).  Output arrays are float32/int32
   directly (the reference feeds float64/int64 and lets TF cast).
"""
from __future__ import annotations

import json
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence

import numpy as np


@dataclass
class MoleculeSet:
    """Structure-of-arrays view of a list of molecule graphs.

    node_ptr   int64 [G+1]   node range of graph g is node_ptr[g]:node_ptr[g+1]
    node_feat  float32 [N,A] per-node annotation (A = annotation_size, 5 for QM9: H,C,N,O,F)
    bond_ptr   int64 [G+1]   bond range of graph g
    bonds      int32 [B,3]   (src_local, bond_type in 1..F, dst_local)   (get_data.py:70-71)
    targets    float32 [G,K] targets[g,k] = d['targets'][k][0]
    """
    node_ptr: np.ndarray
    node_feat: np.ndarray
    bond_ptr: np.ndarray
    bonds: np.ndarray
    targets: np.ndarray

    @property
    def num_graphs(self) -> int:
        return len(self.node_ptr) - 1

    @property
    def annotation_size(self) -> int:
        return self.node_feat.shape[1]

    @property
    def num_fwd_edge_types(self) -> int:
        """max bond id seen (chem_tensorflow.py:116-119)."""
        return int(self.bonds[:, 1].max()) if len(self.bonds) else 0

    def nodes_per_graph(self) -> np.ndarray:
        return np.diff(self.node_ptr)

    # ---- reference JSON schema <-> arrays -------------------------------------------------
    @classmethod
    def from_json(cls, raw: Sequence[dict]) -> "MoleculeSet":
        n = np.array([len(d["node_features"]) for d in raw], dtype=np.int64)
        b = np.array([len(d["graph"]) for d in raw], dtype=np.int64)
        node_ptr = np.concatenate([[0], np.cumsum(n)])
        bond_ptr = np.concatenate([[0], np.cumsum(b)])
        A = len(raw[0]["node_features"][0]) if len(raw) else 0   # chem_tensorflow.py:121
        feat = np.zeros((int(node_ptr[-1]), A), np.float32)
        bonds = np.zeros((int(bond_ptr[-1]), 3), np.int32)
        K = len(raw[0]["targets"]) if len(raw) else 0
        targets = np.zeros((len(raw), K), np.float32)
        for g, d in enumerate(raw):
            feat[node_ptr[g]:node_ptr[g + 1]] = np.asarray(d["node_features"], np.float32)
            if b[g]:
                bonds[bond_ptr[g]:bond_ptr[g + 1]] = np.asarray(d["graph"], np.int32)
            targets[g] = [t[0] for t in d["targets"]]
        return cls(node_ptr, feat, bond_ptr, bonds, targets)

    @classmethod
    def load(cls, path: str) -> "MoleculeSet":
        with open(path, "r") as f:
            return cls.from_json(json.load(f))

    def to_json(self) -> List[dict]:
        out = []
        for g in range(self.num_graphs):
            out.append({
                "targets": [[float(t)] for t in self.targets[g]],
                "graph": [[int(s), int(e), int(d)] for s, e, d in self.bonds[self.bond_ptr[g]:self.bond_ptr[g + 1]]],
                "node_features": self.node_feat[self.node_ptr[g]:self.node_ptr[g + 1]].astype(int).tolist(),
            })
        return out

    def subset(self, idx: np.ndarray) -> "MoleculeSet":
        idx = np.asarray(idx, dtype=np.int64)
        n = np.diff(self.node_ptr)[idx]
        b = np.diff(self.bond_ptr)[idx]
        node_ptr = np.concatenate([[0], np.cumsum(n)])
        bond_ptr = np.concatenate([[0], np.cumsum(b)])
        nsel = _ranges(self.node_ptr[idx], n)
        bsel = _ranges(self.bond_ptr[idx], b)
        return MoleculeSet(node_ptr, self.node_feat[nsel], bond_ptr, self.bonds[bsel], self.targets[idx])


def _ranges(starts: np.ndarray, lengths: np.ndarray) -> np.ndarray:
    """Concatenation of arange(starts[i], starts[i]+lengths[i]) without a Python loop."""
    total = int(lengths.sum())
    if total == 0:
        return np.zeros(0, np.int64)
    ends = np.cumsum(lengths)
    base = np.repeat(starts - (ends - lengths), lengths)
    return base + np.arange(total, dtype=np.int64)


def synthetic_large_graph(num_nodes: int = 100000, num_edges: int = 1000000, num_edge_types: int = 4, seed: int = 0,
                          power_law: bool = False):
    """BASELINE.json configs[4]: ONE large graph -- `num_edges` directed messages split uniformly
    over `num_edge_types`, sources and targets uniform over `num_nodes` (or Zipf-distributed TARGETS with
    power_law=True: a few hub nodes collect thousands of messages), not symmetric.  Returns the sparse model's feed
    pieces as NumPy arrays: (adjacency_lists: T x int32 [E_t,2] sorted by (src,dst) like the reference's packer
    (chem_tensorflow_sparse.py:265), num_incoming_edges_per_type: float32 [V,T])."""
    rng = np.random.default_rng(seed)
    V, M, T = int(num_nodes), int(num_edges), int(num_edge_types)
    types = rng.integers(0, T, M)
    src = rng.integers(0, V, M).astype(np.int32)
    if power_law:
        dst = (np.minimum(rng.zipf(1.6, M), V) - 1).astype(np.int32)
        dst = rng.permutation(V).astype(np.int32)[dst]            # hubs are not the low node ids
    else:
        dst = rng.integers(0, V, M).astype(np.int32)
    adjacency_lists = []
    nin = np.zeros((V, T), np.float32)
    for t in range(T):
        sel = types == t
        a = np.stack([src[sel], dst[sel]], axis=1).astype(np.int32).reshape(-1, 2)
        a = a[np.lexsort((a[:, 1], a[:, 0]))]
        nin[:, t] = np.bincount(a[:, 1], minlength=V).astype(np.float32)
        adjacency_lists.append(np.ascontiguousarray(a))
    return adjacency_lists, nin


def synthetic_qm9(num_graphs: int, mean_nodes: float = 18.0, seed: int = 0, num_bond_types: int = 4,
                  annotation_size: int = 5, num_tasks: int = 1, max_degree: Optional[int] = 4) -> MoleculeSet:
    """Synthetic QM9-shaped molecules.

    n ~ clip(round(N(mean_nodes,3)), 3, 29) atoms (29 = the dense model's largest bucket,
    chem_tensorflow_dense.py:134); a random spanning tree plus 0-2 ring-closure bonds, no atom with more than
    `max_degree` bonds (4 = the largest valence in QM9: C; None = unconstrained recursive tree); bond type ~
    Categorical(0.85, 0.07, 0.03, 0.05) over {1..4} (get_data.py:62); one-hot atom annotation
    uniform over `annotation_size`; z-scored scalar targets.  mean_nodes = 18 is QM9 with hydrogens
    (get_data.py:66 AddHs); BASELINE.json's "~9 nodes" is the heavy-atom count.
    All four bond types are guaranteed to appear (num_edge_types is data-derived,
    chem_tensorflow.py:116-120).
    """
    rng = np.random.default_rng(seed)
    n = np.clip(np.rint(rng.normal(mean_nodes, 3.0, num_graphs)), 3, 29).astype(np.int64)
    node_ptr = np.concatenate([[0], np.cumsum(n)])
    N = int(node_ptr[-1])
    local = np.arange(N, dtype=np.int64) - np.repeat(node_ptr[:-1], n)
    graph_of = np.repeat(np.arange(num_graphs, dtype=np.int64), n)
    nonroot = local > 0
    if max_degree is None:
        # spanning tree: node i>=1 bonds to a uniformly random earlier node of its graph (unbounded degrees)
        parent = np.floor(rng.random(N) * np.maximum(local, 1)).astype(np.int64)
    else:
        # valence-bounded tree: node i bonds to one of the max_degree-1 nodes before it, so a node has at most
        # max_degree-1 children + 1 parent (QM9: no atom has more than 4 bonds)
        if max_degree < 2:
            raise ValueError("max_degree must be >= 2")
        back = 1 + np.floor(rng.random(N) * np.minimum(np.maximum(local, 1), max_degree - 1)).astype(np.int64)
        parent = np.maximum(local - back, 0)
    t_src = parent[nonroot]
    t_dst = local[nonroot]
    t_g = graph_of[nonroot]
    # 0-2 ring closures per graph between distinct nodes that are not already bonded
    k = rng.integers(0, 3, num_graphs)
    r_g = np.repeat(np.arange(num_graphs, dtype=np.int64), k)
    a = np.floor(rng.random(len(r_g)) * n[r_g]).astype(np.int64)
    b = np.floor(rng.random(len(r_g)) * n[r_g]).astype(np.int64)
    lo, hi = np.minimum(a, b), np.maximum(a, b)
    ok = (lo != hi) & (parent[node_ptr[r_g] + hi] != lo)
    r_g, lo, hi = r_g[ok], lo[ok], hi[ok]
    key = (r_g * 32 + lo) * 32 + hi
    _, first = np.unique(key, return_index=True)
    r_g, lo, hi = r_g[first], lo[first], hi[first]
    if max_degree is not None and len(r_g):
        # keep a ring closure only while both atoms still have a free valence (first, then second closure of a graph)
        deg = np.bincount(np.concatenate([node_ptr[t_g] + t_src, node_ptr[t_g] + t_dst]), minlength=N)
        nth = np.arange(len(r_g)) - np.searchsorted(r_g, r_g, side="left")      # 0 / 1 within the graph (r_g is sorted)
        keep = np.zeros(len(r_g), dtype=bool)
        for j in range(int(nth.max()) + 1):
            sel = np.nonzero(nth == j)[0]
            ga, gb = node_ptr[r_g[sel]] + lo[sel], node_ptr[r_g[sel]] + hi[sel]
            good = (deg[ga] < max_degree) & (deg[gb] < max_degree)
            keep[sel[good]] = True
            np.add.at(deg, ga[good], 1)
            np.add.at(deg, gb[good], 1)
        r_g, lo, hi = r_g[keep], lo[keep], hi[keep]
    g_all = np.concatenate([t_g, r_g])
    s_all = np.concatenate([t_src, lo])
    d_all = np.concatenate([t_dst, hi])
    order = np.argsort(g_all, kind="stable")
    g_all, s_all, d_all = g_all[order], s_all[order], d_all[order]
    probs = np.array([0.85, 0.07, 0.03, 0.05][:num_bond_types], dtype=np.float64)
    probs = probs / probs.sum()
    btype = rng.choice(np.arange(1, num_bond_types + 1), size=len(g_all), p=probs).astype(np.int64)
    if len(btype) >= num_bond_types:
        btype[:num_bond_types] = np.arange(1, num_bond_types + 1)
    bonds = np.stack([s_all, btype, d_all], axis=1).astype(np.int32)
    bond_ptr = np.concatenate([[0], np.cumsum(np.bincount(g_all, minlength=num_graphs))]).astype(np.int64)
    feat = np.zeros((N, annotation_size), np.float32)
    feat[np.arange(N), rng.integers(0, annotation_size, N)] = 1.0
    targets = rng.normal(0, 1, (num_graphs, num_tasks)).astype(np.float32)
    return MoleculeSet(node_ptr, feat, bond_ptr, bonds, targets)


@dataclass
class SparseBatch:
    """One minibatch in the reference feed layout (chem_tensorflow_sparse.py:331-350), as NumPy
    float32/int32 arrays.  Keys mirror the reference placeholder names."""
    node_features: np.ndarray                    # [V,A] f32 (un-padded annotations)
    hidden_size: int
    adjacency_lists: List[np.ndarray]            # T x [E_t,2] i32 (src,dst)
    num_incoming_edges_per_type: np.ndarray      # [V,T] f32
    graph_nodes_list: np.ndarray                 # [V] i32
    target_values: np.ndarray                    # [tasks,G] f32
    target_mask: np.ndarray                      # [tasks,G] f32
    num_graphs: int
    extras: Dict[str, object] = field(default_factory=dict)

    @property
    def initial_node_representation(self) -> np.ndarray:
        """[V,D] f32: annotations zero-padded to hidden_size (chem_tensorflow_sparse.py:300-302).  Built
        on demand; the device upload pads on the GPU instead of touching 4*V*D host bytes per batch."""
        V, A = self.node_features.shape
        h0 = np.zeros((V, self.hidden_size), np.float32)
        h0[:, :A] = self.node_features
        return h0

    @property
    def num_nodes(self) -> int:
        return self.node_features.shape[0]

    @property
    def num_messages(self) -> int:
        return int(sum(len(a) for a in self.adjacency_lists))


def batch_boundaries(nodes_per_graph: np.ndarray, batch_size: int) -> List[int]:
    """Greedy packing of chem_tensorflow_sparse.py:287-297: a batch keeps taking the next graph
    while node_offset + n < batch_size (strict).  Returns graph-index boundaries [0, ..., G].
    A graph with n >= batch_size would make the reference loop forever (an empty batch is yielded
    and num_graphs never advances); that is reported as an error here."""
    if (nodes_per_graph >= batch_size).any():
        raise ValueError("a graph has >= batch_size nodes; the reference packer cannot place it")
    csum = np.concatenate([[0], np.cumsum(nodes_per_graph)])
    bounds = [0]
    G = len(nodes_per_graph)
    while bounds[-1] < G:
        s = bounds[-1]
        # largest e with csum[e] - csum[s] < batch_size
        e = int(np.searchsorted(csum, csum[s] + batch_size, side="left")) - 1
        bounds.append(min(max(e, s + 1), G))
    return bounds


def epoch_boundaries(nodes_per_graph: np.ndarray, batch_size: int, world_size: int = 1, balance: bool = True) -> List[int]:
    """Batch boundaries of one epoch for `world_size` data-parallel ranks (batch i goes to rank i % world_size).

    One rank, or an epoch whose greedy batch count B (batch_boundaries: the reference's packing) already divides by the rank
    count: the reference's batches.  Otherwise sharding whole greedy batches pads the last step with EMPTY batches -- full
    QM9 (~25 batches of < 100,000 nodes) on 8 ranks runs 4 steps of which the last is 1/8 occupied: 25/32 = 0.78 of the
    ranks' time does work.  With `balance` the epoch is re-cut into ceil(B / N) * N batches of EQUAL node count (each still
    below batch_size, the reference's bound, chem_tensorflow_sparse.py:297): every rank carries the same work in every step
    and no step is padded.  A step of the data-parallel job is the union of its N batches under ONE global loss
    normalisation (parallel.py), so re-cutting changes which graphs share a step -- like any other batch_size -- not what a
    step computes.  Falls back to the greedy boundaries when an equal cut would overflow a batch or leave one empty."""
    bounds = batch_boundaries(nodes_per_graph, batch_size)
    nb = len(bounds) - 1
    if world_size <= 1 or not balance or nb == 0 or nb % world_size == 0:
        return bounds
    target = (nb + world_size - 1) // world_size * world_size
    G = len(nodes_per_graph)
    if target > G:
        return bounds
    csum = np.concatenate([[0], np.cumsum(nodes_per_graph)]).astype(np.int64)
    total = int(csum[-1])
    cuts = [0]
    for k in range(1, target):
        want = total * k / target
        e = int(np.searchsorted(csum, want, side="left"))                 # first boundary at or past the target
        if e > 0 and abs(csum[e - 1] - want) <= abs(csum[min(e, G)] - want):
            e -= 1                                                        # ... or the one before it, whichever is nearer
        e = min(max(e, cuts[-1] + 1), G - (target - k))                   # at least one graph per batch, also for those to come
        while e > cuts[-1] + 1 and csum[e] - csum[cuts[-1]] >= batch_size:
            e -= 1
        cuts.append(e)
    cuts.append(G)
    sizes = np.diff(csum[cuts])
    if (sizes >= batch_size).any() or (np.diff(cuts) <= 0).any():
        return bounds
    return [int(c) for c in cuts]


def pack_batch(ms: MoleculeSet, graph_ids: np.ndarray, num_edge_types: int, hidden_size: int,
               tie_fwd_bkwd: bool = True, task_ids: Sequence[int] = (0,),
               label_mask: Optional[np.ndarray] = None) -> SparseBatch:
    """Build ONE batch from graphs `graph_ids` (in this order).  Vectorised restatement of
    chem_tensorflow_sparse.py:254-276 + :298-348.

    For tie_fwd_bkwd=False the reference code is broken; the intended semantics are
    implemented: forward types 0..F-1, backward types F..2F-1 with F = num_edge_types//2, the
    reversed edge (dst,src) counted as incoming at src.
    """
    graph_ids = np.asarray(graph_ids, dtype=np.int64)
    n = np.diff(ms.node_ptr)[graph_ids]
    G = len(graph_ids)
    offs = np.concatenate([[0], np.cumsum(n)])
    V = int(offs[-1])
    nsel = _ranges(ms.node_ptr[graph_ids], n)
    if ms.annotation_size > hidden_size:
        raise ValueError("annotation_size %d exceeds hidden_size %d" % (ms.annotation_size, hidden_size))
    feats = ms.node_feat[nsel]                                                   # padded to D lazily (:300-302)
    gnl = np.repeat(np.arange(G, dtype=np.int32), n)                             # :304
    nb = np.diff(ms.bond_ptr)[graph_ids]
    bsel = _ranges(ms.bond_ptr[graph_ids], nb)
    bonds = ms.bonds[bsel].astype(np.int64)
    boff = np.repeat(offs[:-1], nb)
    src = bonds[:, 0] + boff                                                     # :307 (+ node_offset)
    dst = bonds[:, 2] + boff
    typ = bonds[:, 1] - 1                                                        # :258
    if tie_fwd_bkwd:
        s_all = np.concatenate([src, dst]); d_all = np.concatenate([dst, src])   # :259-263
        t_all = np.concatenate([typ, typ])
    else:
        F = num_edge_types // 2
        s_all = np.concatenate([src, dst]); d_all = np.concatenate([dst, src])
        t_all = np.concatenate([typ, typ + F])
    if len(t_all) and (t_all.min() < 0 or t_all.max() >= num_edge_types):
        raise IndexError("edge type outside [0, num_edge_types)")
    order = np.lexsort((d_all, s_all, t_all))                                    # :265 sorted((src,dst)) per type
    s_all, d_all, t_all = s_all[order], d_all[order], t_all[order]
    cnt = np.bincount(t_all, minlength=num_edge_types)
    tptr = np.concatenate([[0], np.cumsum(cnt)])
    adjacency = [np.stack([s_all[tptr[t]:tptr[t + 1]], d_all[tptr[t]:tptr[t + 1]]], axis=1).astype(np.int32)
                 if cnt[t] else np.zeros((0, 2), np.int32) for t in range(num_edge_types)]   # :343-348
    nin = np.bincount(d_all * num_edge_types + t_all, minlength=V * num_edge_types)
    nin = nin.reshape(V, num_edge_types).astype(np.float32)                      # :310-313
    task_ids = list(task_ids)
    tv = ms.targets[graph_ids][:, task_ids].T.astype(np.float32).copy()          # :335
    tm = np.ones_like(tv) if label_mask is None else label_mask[graph_ids][:, task_ids].T.astype(np.float32)
    tv = tv * tm                                                                 # masked labels feed 0. (:319-321)
    return SparseBatch(feats, hidden_size, adjacency, nin, gnl, tv, tm, G, extras={"graph_ids": graph_ids})


def pack_batches(ms: MoleculeSet, params: dict, num_edge_types: int, order: Optional[np.ndarray] = None,
                 label_mask: Optional[np.ndarray] = None, rank: int = 0, world_size: int = 1) -> List[SparseBatch]:
    """All minibatches of one epoch (chem_tensorflow_sparse.py:278-350) for graph order `order`.

    Data parallel (world_size > 1): batch i goes to rank i % world_size and every rank gets the same number of batches.
    The epoch is cut by epoch_boundaries: into equal-node batches whose count divides by the rank count
    (params['dp_balance_nodes'], default on), or -- switched off -- exactly as on one device, the tail padded with EMPTY
    batches (no graphs) so that the per-step gradient all-reduce stays matched."""
    G = ms.num_graphs
    order = np.arange(G, dtype=np.int64) if order is None else np.asarray(order, np.int64)
    bounds = epoch_boundaries(np.diff(ms.node_ptr)[order], params["batch_size"], world_size, bool(params.get("dp_balance_nodes", True)))
    nb = len(bounds) - 1
    steps = (nb + world_size - 1) // world_size
    out = []
    for s in range(steps):
        i = s * world_size + rank
        ids = order[bounds[i]:bounds[i + 1]] if i < nb else np.zeros(0, np.int64)
        out.append(pack_batch(ms, ids, num_edge_types, params["hidden_size"], params.get("tie_fwd_bkwd", True),
                              params.get("task_ids", [0]), label_mask))
    return out


# ---- dense layout (chem_tensorflow_dense.py:30-36, 132-228) ----------------------------------
@dataclass
class DenseBatch:
    initial_node_representation: np.ndarray      # [b,v,D] f32
    adjacency_matrix: np.ndarray                 # [b,e,v,v] f32, A[b,e,dst,src]
    node_mask: np.ndarray                        # [b,v] f32
    num_vertices: int
    target_values: np.ndarray                    # [tasks,b]
    target_mask: np.ndarray                      # [tasks,b]
    num_graphs: int


DENSE_BUCKET_SIZES = np.array(list(range(4, 28, 2)) + [29])   # chem_tensorflow_dense.py:134


def pack_dense_batch(ms: MoleculeSet, graph_ids: np.ndarray, num_vertices: int, num_edge_types: int,
                     hidden_size: int, tie_fwd_bkwd: bool = True, task_ids: Sequence[int] = (0,),
                     label_mask: Optional[np.ndarray] = None) -> DenseBatch:
    """chem_tensorflow_dense.py:30-36 (graph_to_adj_mat), :143-153 (pad_annotations + mask), :175-193 (make_batch:
    a label masked by task_sample_ratios feeds value 0 / mask 0).  label_mask: [num_graphs, len(task_ids)] or None."""
    graph_ids = np.asarray(graph_ids, np.int64)
    b = len(graph_ids)
    v = num_vertices
    n = np.diff(ms.node_ptr)[graph_ids]
    if (n > v).any():
        raise ValueError("graph larger than bucket size")
    h0 = np.zeros((b, v, hidden_size), np.float32)
    mask = np.zeros((b, v), np.float32)
    A = np.zeros((b, num_edge_types, v, v), np.float32)
    gi = np.repeat(np.arange(b), n)
    li = _ranges(np.zeros(b, np.int64), n)
    nsel = _ranges(ms.node_ptr[graph_ids], n)
    h0[gi, li, :ms.annotation_size] = ms.node_feat[nsel]
    mask[gi, li] = 1.0
    nb = np.diff(ms.bond_ptr)[graph_ids]
    bsel = _ranges(ms.bond_ptr[graph_ids], nb)
    bonds = ms.bonds[bsel].astype(np.int64)
    bg = np.repeat(np.arange(b), nb)
    bwd = 0 if tie_fwd_bkwd else num_edge_types // 2
    A[bg, bonds[:, 1] - 1, bonds[:, 2], bonds[:, 0]] = 1.0            # amat[e-1, dest, src] = 1
    A[bg, bonds[:, 1] - 1 + bwd, bonds[:, 0], bonds[:, 2]] = 1.0      # amat[e-1+off, src, dest] = 1
    tv = ms.targets[graph_ids][:, list(task_ids)].T.astype(np.float32).copy()
    tm = np.ones_like(tv) if label_mask is None else label_mask[graph_ids].T.astype(np.float32).copy()
    return DenseBatch(h0, A, mask, v, tv * tm, tm, b)

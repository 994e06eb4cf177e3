"""DenseGGNNChemModel -- host-side mirror of chem_tensorflow_dense.py:51-265 on PyTorch-ROCm tensors.

Per timestep (chem_tensorflow_dense.py:100-115):
    m_e  = h W_e (+ b_e)          for all e: ONE FP32-MFMA GEMM [b*v,D]x[D,e*D]  (ggnn_msg_transform_f32)
    acts = sum_e A_e m_e          batched [v,v]x[v,D] from LDS                   (ggnn_dense_aggregate_f32)
    h    = GRU(acts, h)           one GRU shared by all timesteps (:101-102)     (ggnn_gru_f32)
Forward (inference / validation) path; the dense training path is not built (the sparse model is the
north-star path).
"""
from __future__ import annotations

from collections import defaultdict
from typing import Any, Dict, Sequence

import numpy as np
import torch

from . import ops
from .chem_model import ChemModel
from .data import DENSE_BUCKET_SIZES, MoleculeSet, pack_dense_batch
from .sparse_model import GRUCellWeights
from .utils import glorot_init, tf_dropout, tf_glorot_uniform

import os

# Inference with all timesteps in one graph-resident launch (GGNN_DENSE_GRAPH_KERNEL=0: three launches per timestep)
USE_GRAPH_RESIDENT_KERNEL = os.environ.get("GGNN_DENSE_GRAPH_KERNEL", "1") != "0"


class DenseGGNNChemModel(ChemModel):
    def __init__(self, args):
        super().__init__(args)

    @classmethod
    def default_params(cls):
        # chem_tensorflow_dense.py:57-66
        params = dict(super().default_params())
        params.update({
            'batch_size': 256,
            'graph_state_dropout_keep_prob': 1.,
            'task_sample_ratios': {},
            'use_edge_bias': True,
            'edge_weight_dropout_keep_prob': 1
        })
        return params

    def prepare_specific_graph_model(self) -> None:
        """chem_tensorflow_dense.py:68-91."""
        h_dim = self.params['hidden_size']
        for name in ('initial_node_representation', 'node_mask', 'num_vertices', 'adjacency_matrix'):
            self.placeholders[name] = None
        self.placeholders['graph_state_keep_prob'] = 1.0
        self.placeholders['edge_weight_dropout_keep_prob'] = 1.0
        dev = self.device
        # :84 glorot over the last two dims of [e,h,h]
        self.weights['edge_weights'] = torch.from_numpy(glorot_init([self.num_edge_types, h_dim, h_dim])).to(dev)
        if self.params['use_edge_bias']:
            self.weights['edge_biases'] = torch.zeros([self.num_edge_types, 1, h_dim], dtype=torch.float32, device=dev)
        # :87-90 tf.contrib.rnn.GRUCell(h_dim) (tanh), gate bias 1, candidate bias 0
        self.weights['node_gru'] = GRUCellWeights(tf_glorot_uniform([2 * h_dim, 2 * h_dim], self.tf_generator).to(dev),
                                                  torch.ones(2 * h_dim, dtype=torch.float32, device=dev),
                                                  tf_glorot_uniform([2 * h_dim, h_dim], self.tf_generator).to(dev),
                                                  torch.zeros(h_dim, dtype=torch.float32, device=dev))

    def graph_model_variables(self) -> Dict[str, torch.Tensor]:
        out = {"graph_model/Variable:0": self.weights['edge_weights']}
        if self.params['use_edge_bias']:
            out["graph_model/Variable_1:0"] = self.weights['edge_biases']
        cell = self.weights['node_gru']
        base = "graph_model/gru_scope/gru_cell"
        out[base + "/gates/kernel:0"] = cell.gates_kernel; out[base + "/gates/bias:0"] = cell.gates_bias
        out[base + "/candidate/kernel:0"] = cell.candidate_kernel; out[base + "/candidate/bias:0"] = cell.candidate_bias
        return out

    def set_graph_weights(self, edge_weights, edge_biases, gru: dict) -> None:
        with torch.no_grad():
            as_t = lambda a: torch.as_tensor(np.asarray(a, dtype=np.float32)).to(self.device)
            self.weights['edge_weights'].copy_(as_t(edge_weights))
            if self.params['use_edge_bias']:
                self.weights['edge_biases'].copy_(as_t(edge_biases).reshape(self.weights['edge_biases'].shape))
            c = self.weights['node_gru']
            c.gates_kernel.copy_(as_t(gru['Wg'])); c.gates_bias.copy_(as_t(gru['bg']))
            c.candidate_kernel.copy_(as_t(gru['Wc'])); c.candidate_bias.copy_(as_t(gru['bc']))

    def propagate_format(self, v: int) -> int:
        """Operand format of the graph-resident dense forward for the fed batch (formats.py: the two-piece f16 format only where its
        range is PROVEN, the exact bf16x3 split otherwise).  The dense cell is the tanh GRU (chem_tensorflow_dense.py:88), so every
        state stays <= S = max(1, max|h0|); a vertex sums over at most v * E (source, type) pairs: |acts| <= v E (D max|W_e| S +
        max|b_e|) (:103-112); the weights must lie within the x 2^8 packing's range.  Kept in self.last_format / last_format_bounds."""
        from . import formats
        pol = formats.policy()
        if not formats.split_path() or pol == "exact":
            fmt = formats.BF16X3
        elif pol == "force2":
            fmt = formats.F16X2
        else:
            cell = self.weights['node_gru']
            ts = [self.weights['edge_weights'], cell.gates_kernel, cell.candidate_kernel]
            if self.params['use_edge_bias']:
                ts.append(self.weights['edge_biases'])
            mx = formats.weight_absmax(ts)
            S = formats.state_bound(formats.h0_absmax(self.placeholders), 'tanh')
            # (|A| <= 1 for the reference's 0 / 1 adjacency; a weighted or multi-edge feed scales the sum: measured, cached per tensor)
            a_max = formats.adjacency_absmax(self.placeholders['adjacency_matrix'])
            acts = max(1.0, a_max) * v * self.num_edge_types * (self.params['hidden_size'] * mx[0] * S + (mx[3] if len(mx) > 3 else 0.0))
            if a_max != a_max:
                acts = float("nan")
            fmt = formats.layer_format(S, acts, formats.nanmax(mx[0], mx[1], mx[2]))
            self.last_format_bounds = {"proven": fmt == formats.F16X2, "state_bound": S, "acts_bound": acts, "adjacency_absmax": a_max,
                                       "weight_absmax": formats.nanmax(mx[0], mx[1], mx[2])}
        self.last_format = fmt
        return fmt

    def compute_final_node_representations(self) -> torch.Tensor:
        """chem_tensorflow_dense.py:93-117."""
        if self.training and torch.is_grad_enabled():
            return self._compute_for_training()
        ph = self.placeholders
        v = ph['num_vertices']
        h_dim = self.params['hidden_size']
        h = ph['initial_node_representation']                          # [b, v, h]
        b = h.shape[0]
        h = h.reshape(-1, h_dim).contiguous()                          # :97
        A = ph['adjacency_matrix']                                     # [b, e, v, v]; the :80 transpose is an indexing choice
        keep_w = float(ph.get('edge_weight_dropout_keep_prob', 1.0))
        keep_s = float(ph.get('graph_state_keep_prob', 1.0))
        bias = self.weights['edge_biases'].reshape(self.num_edge_types, h_dim) if self.params['use_edge_bias'] else None
        cell = self.weights['node_gru']
        # the GRU's LDS weight images are packed once per weight version (the one cell is shared by all timesteps, :101-102)
        from .autograd import _PACKED
        if USE_GRAPH_RESIDENT_KERNEL and keep_w >= 1.0 and keep_s >= 1.0 and h.is_cuda \
                and ops.dense_propagate_supported(int(v), self.num_edge_types, h_dim):
            # all timesteps in ONE launch: a graph's states stay on its CU (ggnn_dense_propagate_f32)
            W = self.weights['edge_weights'].contiguous()
            return ops.dense_propagate(h.reshape(b, int(v), h_dim), A.contiguous(), _PACKED.dense_edge(W),
                                       _PACKED.dense_gru(cell.gates_kernel, cell.candidate_kernel, h_dim), bias,
                                       cell.gates_bias, cell.candidate_bias, self.params['num_timesteps'],
                                       fmt=self.propagate_format(int(v)))
        packed = _PACKED.gru(cell.gates_kernel, cell.candidate_kernel, 1, h_dim) if ops.gru_is_fused(h_dim) else None
        for i in range(self.params['num_timesteps']):                  # :100
            # :104 a fresh weight-dropout mask per (timestep, edge type)
            W = tf_dropout(self.weights['edge_weights'], keep_w, self.dropout_seed('edge_weights', i)).contiguous()
            Hm = ops.msg_transform(h, W)                               # :104-106 for all edge types
            acts = ops.dense_aggregate(A, Hm, bias)                    # :107-112
            if packed is not None:
                h = ops.gru_packed([acts], h, packed, cell.gates_bias, cell.candidate_bias, "tanh")      # :115
            else:
                h = ops.gru([acts], h, cell.gates_kernel, cell.gates_bias, cell.candidate_kernel, cell.candidate_bias,
                            "tanh")                                    # :115
            h = tf_dropout(h, keep_s, self.dropout_seed('state', i))
        return h.reshape(b, v, h_dim)                                  # :116

    def _compute_for_training(self) -> torch.Tensor:
        """Training form of chem_tensorflow_dense.py:93-117.  A dense step IS a sparse step on the b*v padded nodes:
        acts[d] = sum_e sum_s A_e[d,s] (h_s W_e + b_e) = segment_sum of the transformed source rows + nin @ b_e, with
        sum aggregation and the single shared GRU (the identity tests/test_oracle.py pins).  So the 0/1 adjacency
        tensor is turned into per-type (src, dst) lists once per batch and every timestep runs through the same
        differentiable step as the sparse model (backward.PropagationStepFn: hand-written backward kernels).  Padded
        vertices are isolated nodes there -- their state still evolves through the GRU biases, as in the reference,
        and is masked only at the readout (:126)."""
        from .autograd import propagation_step
        ph = self.placeholders
        v = int(ph['num_vertices'])
        h_dim, T = self.params['hidden_size'], self.num_edge_types
        h0 = ph['initial_node_representation']
        b = h0.shape[0]
        A = ph['adjacency_matrix']                                      # [b, e, dst, src]
        sparse_form = ph.get('_sparse_form')
        if sparse_form is None or sparse_form[0] is not A:
            nz = A.nonzero()                                            # rows (b, e, dst, src), lexicographic
            base = nz[:, 0] * v
            pairs = torch.stack([base + nz[:, 3], base + nz[:, 2]], dim=1).to(torch.int32)
            etype = nz[:, 1]
            adjacency_lists = [pairs[etype == t].contiguous() for t in range(T)]
            nin = A.sum(dim=3).permute(0, 2, 1).reshape(b * v, T).to(torch.float32).contiguous()
            sparse_form = ph['_sparse_form'] = (A, ops.build_message_index(adjacency_lists, b * v), nin)
        _, index, nin = sparse_form
        keep_w = float(ph.get('edge_weight_dropout_keep_prob', 1.0))
        keep_s = float(ph.get('graph_state_keep_prob', 1.0))
        bias = self.weights['edge_biases'].reshape(T, h_dim) if self.params['use_edge_bias'] else None
        cell = self.weights['node_gru']
        h = h0.reshape(-1, h_dim).contiguous()
        for i in range(self.params['num_timesteps']):
            # :104 a fresh mask per (timestep, edge type): the [e, h, h] tensor's rows are keyed (e, row), one seed per timestep
            ew_mask = (keep_w, self.dropout_seed('edge_weights', i)) if keep_w < 1.0 else None
            h = propagation_step(h, index, nin, self.weights['edge_weights'], bias, False, [], cell, "tanh", need_grad=True,
                                 ew_mask=ew_mask)
            h = tf_dropout(h, keep_s, self.dropout_seed('state', i))
        return h.reshape(b, v, h_dim)

    def gated_regression_with_loss(self, last_h, regression_gate, regression_transform, target_values, target_mask):
        """chem_tensorflow_dense.py:119-129 + chem_tensorflow.py:161-166 on the fused readout kernels: graph g owns the
        consecutive rows g*v .. g*v+v-1 of the flattened [b*v, h] states, padding vertices are switched off by node_mask."""
        from .autograd import readout_loss
        ph = self.placeholders
        h_dim = self.params['hidden_size']
        g, t = regression_gate.params, regression_transform.params
        if not last_h.is_cuda or h_dim > 256 or len(g["weights"]) != 1 or len(t["weights"]) != 1:
            return None
        b, v = last_h.shape[0], int(ph['num_vertices'])
        key = (b, v, str(last_h.device))
        cached = getattr(self, '_readout_index', None)
        if cached is None or cached[0] != key:
            gnl = torch.arange(b, dtype=torch.int32, device=last_h.device).repeat_interleave(v).contiguous()
            gptr = (torch.arange(b + 1, dtype=torch.int32, device=last_h.device) * v).contiguous()
            self._readout_index = cached = (key, gnl, gptr)
        keep = float(ph.get('out_layer_dropout_keep_prob', 1.0))
        out, num, ab, ms = readout_loss(last_h.reshape(-1, h_dim), ph['initial_node_representation'].reshape(-1, h_dim).contiguous(),
                                        cached[1], cached[2], ph['node_mask'].reshape(-1).contiguous(), b,
                                        regression_gate.dropped_weight(0), g["biases"][0], regression_transform.dropped_weight(0),
                                        t["biases"][0], target_values.contiguous(), target_mask.contiguous())
        self.output = out
        return out, num, ab, ms

    def gated_regression(self, last_h, regression_gate, regression_transform):
        """chem_tensorflow_dense.py:119-129."""
        ph = self.placeholders
        h_dim = self.params['hidden_size']
        gate_input = torch.cat([last_h, ph['initial_node_representation']], dim=2).reshape(-1, 2 * h_dim)
        last = last_h.reshape(-1, h_dim)
        gated_outputs = torch.sigmoid(regression_gate(gate_input)) * regression_transform(last)     # [b*v, 1]
        gated_outputs = gated_outputs.reshape(-1, ph['num_vertices'])                                # [b, v]
        masked_gated_outputs = gated_outputs * ph['node_mask']
        output = masked_gated_outputs.sum(dim=1)                                                     # [b]
        self.output = output
        return output

    # ----- Data preprocessing and chunking into minibatches:
    def process_raw_graphs(self, raw_data, is_training_data: bool, bucket_sizes=None) -> Any:
        """chem_tensorflow_dense.py:132-164: bucket graphs by padded size; a bucket holds graph ids."""
        ms = raw_data if isinstance(raw_data, MoleculeSet) else MoleculeSet.from_json(raw_data)
        if bucket_sizes is None:
            bucket_sizes = DENSE_BUCKET_SIZES
        n = ms.nodes_per_graph()
        # :138 argmax(bucket_sizes > max vertex index mentioned by an edge); graphs without bonds (the
        # reference would fail on max([])) fall back to their node count
        max_idx = n - 1
        nb = np.diff(ms.bond_ptr)
        has = nb > 0
        if has.any():
            ends = np.maximum(ms.bonds[:, 0], ms.bonds[:, 2]).astype(np.int64)
            max_idx = max_idx.copy()
            max_idx[has] = np.maximum.reduceat(ends, ms.bond_ptr[:-1][has])
        if (max_idx >= n).any():
            raise IndexError("a bond mentions a vertex outside its graph")
        chosen = np.searchsorted(bucket_sizes, np.maximum(max_idx, n - 1), side='right')
        if (chosen >= len(bucket_sizes)).any():
            raise ValueError("graph larger than the largest bucket size")
        bucketed = defaultdict(list)
        for g, bi in enumerate(chosen):
            bucketed[int(bi)].append(g)
        # :153-158 per bucket: shuffle, then labels of the examples beyond task_sample_ratios are masked.  The mask
        # belongs to the graph (it follows it through the later per-epoch shuffles).  The reference's `labels[task_id] = None`
        # indexes the per-task label list (len(task_ids) entries) by TASK ID: the right label only for task_ids == [0..k), an
        # IndexError or a neighbour's label otherwise.  Like the sparse model here, the label of THAT task is masked.
        label_mask = np.ones((ms.num_graphs, len(self.params['task_ids'])), dtype=np.float32)
        if is_training_data:
            for bucket_list in bucketed.values():
                np.random.shuffle(bucket_list)
                for internal_id, task_id in enumerate(self.params['task_ids']):
                    task_sample_ratio = self.params['task_sample_ratios'].get(str(task_id))
                    if task_sample_ratio is not None:
                        ex_to_sample = int(len(bucket_list) * task_sample_ratio)
                        if bucket_list[ex_to_sample:]:
                            label_mask[np.asarray(bucket_list[ex_to_sample:]), internal_id] = 0.0
        # :160-162 one entry per full batch of a bucket (remainder graphs are dropped)
        bucket_at_step = [[bucket_idx for _ in range(len(bucket_data) // self.params['batch_size'])]
                          for bucket_idx, bucket_data in bucketed.items()]
        bucket_at_step = [x for y in bucket_at_step for x in y]
        return {"molecules": ms, "bucketed": dict(bucketed), "bucket_sizes": np.asarray(bucket_sizes),
                "bucket_at_step": bucket_at_step, "device_batches": {}, "label_mask": label_mask}

    def to_device_batch(self, db) -> Dict[str, Any]:
        dev = self.device
        t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
        return {'initial_node_representation': t(db.initial_node_representation), 'adjacency_matrix': t(db.adjacency_matrix),
                'node_mask': t(db.node_mask), 'num_vertices': db.num_vertices, 'target_values': t(db.target_values),
                'target_mask': t(db.target_mask), 'num_graphs': db.num_graphs}

    def make_minibatch_iterator(self, data, is_training: bool):
        """chem_tensorflow_dense.py:195-228."""
        ms: MoleculeSet = data["molecules"]
        bucketed, bucket_sizes, bucket_at_step = data["bucketed"], data["bucket_sizes"], data["bucket_at_step"]
        if is_training:                                   # :197-200 both shuffles are in place (orders compose over epochs)
            np.random.shuffle(bucket_at_step)
            for bucket in bucketed.values():
                np.random.shuffle(bucket)
        bucket_counters = defaultdict(int)
        dropout_keep_prob = self.params['graph_state_dropout_keep_prob'] if is_training else 1.
        bs = self.params['batch_size']
        for step in range(len(bucket_at_step)):
            bucket = bucket_at_step[step]
            start_idx = bucket_counters[bucket] * bs
            ids = np.asarray(bucketed[bucket][start_idx:start_idx + bs])
            key = (bucket, bucket_counters[bucket])
            if is_training or key not in data["device_batches"]:
                db = pack_dense_batch(ms, ids, int(bucket_sizes[bucket]), self.num_edge_types,
                                      self.params['hidden_size'], self.params['tie_fwd_bkwd'], self.params['task_ids'],
                                      label_mask=data.get("label_mask"))
                feed = self.to_device_batch(db)
                if not is_training:
                    data["device_batches"][key] = feed
            else:
                feed = data["device_batches"][key]
            feed = dict(feed)
            # :222-223 the dense model feeds graph_state_dropout_keep_prob into BOTH keep-prob placeholders
            feed['graph_state_keep_prob'] = dropout_keep_prob
            feed['edge_weight_dropout_keep_prob'] = dropout_keep_prob
            bucket_counters[bucket] += 1
            yield feed

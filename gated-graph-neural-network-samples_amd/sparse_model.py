"""SparseGGNNChemModel -- host-side mirror of chem_tensorflow_sparse.py:36-376 on PyTorch-ROCm tensors,
with compute_final_node_representations() running on the hand-written gfx950 kernels of
libggnn_hip.so (msg_transform -> gather_segment_sum -> gru per timestep).

Per timestep the reference executes (chem_tensorflow_sparse.py:153-216)
    T gathers + T [E_t,D]x[D,D] matmuls + concat + unsorted_segment_sum (+ bias) (/ degree) + concat + GRUCell;
here it is three launches (one FP32-MFMA GEMM [V,D]x[D,T*D], one gather/segment-sum with the
bias/mean epilogue, one fused GRU = two MFMA GEMMs with sigmoid/tanh/blend epilogues) and no [M,D]
or concat temporaries.
"""
from __future__ import annotations

from collections import namedtuple
from typing import Any, Dict, List, Optional, Sequence

import numpy as np
import torch

from . import ops
from .chem_model import ChemModel
from .data import MoleculeSet, SparseBatch, pack_batches
from .data_device import DeviceMoleculeSet, pack_batches_device
from .utils import glorot_init, SMALL_NUMBER, tf_dropout, tf_glorot_uniform

GGNNWeights = namedtuple('GGNNWeights', ['edge_weights',
                                         'edge_biases',
                                         'edge_type_attention_weights',
                                         'rnn_cells', ])

GRUCellWeights = namedtuple('GRUCellWeights', ['gates_kernel', 'gates_bias', 'candidate_kernel', 'candidate_bias'])
RNNCellWeights = namedtuple('RNNCellWeights', ['kernel', 'bias'])
CudnnGRUCellWeights = namedtuple('CudnnGRUCellWeights', ['gates_kernel', 'gates_bias', 'input_kernel', 'input_bias',
                                                         'hidden_kernel', 'hidden_bias'])

# graph_rnn_cell (lower-cased, chem_tensorflow_sparse.py:102-112) -> (weights tuple, per field:
#   (oracle key, TF-1.3 variable suffix under <layer scope>/timestep_0/, shape(in_dim, D), initial value))
CELL_SPECS = {
    'gru': (GRUCellWeights, [
        ('Wg', 'gru_cell/gates/kernel:0', lambda i, d: (i + d, 2 * d), 'glorot'),
        ('bg', 'gru_cell/gates/bias:0', lambda i, d: (2 * d,), 1.0),
        ('Wc', 'gru_cell/candidate/kernel:0', lambda i, d: (i + d, d), 'glorot'),
        ('bc', 'gru_cell/candidate/bias:0', lambda i, d: (d,), 0.0)]),
    'rnn': (RNNCellWeights, [
        ('W', 'basic_rnn_cell/kernel:0', lambda i, d: (i + d, d), 'glorot'),
        ('b', 'basic_rnn_cell/bias:0', lambda i, d: (d,), 0.0)]),
    'cudnncompatiblegrucell': (CudnnGRUCellWeights, [
        ('Wg', 'cudnn_compatible_gru_cell/gates/kernel:0', lambda i, d: (i + d, 2 * d), 'glorot'),
        ('bg', 'cudnn_compatible_gru_cell/gates/bias:0', lambda i, d: (2 * d,), 1.0),
        ('Wcx', 'cudnn_compatible_gru_cell/candidate/input_projection/kernel:0', lambda i, d: (i, d), 'glorot'),
        ('bcx', 'cudnn_compatible_gru_cell/candidate/input_projection/bias:0', lambda i, d: (d,), 0.0),
        ('Wch', 'cudnn_compatible_gru_cell/candidate/hidden_projection/kernel:0', lambda i, d: (d, d), 'glorot'),
        ('bch', 'cudnn_compatible_gru_cell/candidate/hidden_projection/bias:0', lambda i, d: (d,), 0.0)]),
}


class SparseGGNNChemModel(ChemModel):
    def __init__(self, args):
        super().__init__(args)

    @classmethod
    def default_params(cls):
        # chem_tensorflow_sparse.py:40-61
        params = dict(super().default_params())
        params.update({
            'batch_size': 100000,
            'use_edge_bias': False,
            'use_propagation_attention': False,
            'use_edge_msg_avg_aggregation': True,
            'residual_connections': {  # For layer i, specify list of layers whose output is added as an input
                                     "2": [0],
                                     "4": [0, 2]
                                    },

            'layer_timesteps': [2, 2, 1, 2, 1],  # number of layers & propagation steps per layer

            'graph_rnn_cell': 'GRU',  # GRU, CudnnCompatibleGRUCell, or RNN
            'graph_rnn_activation': 'tanh',  # tanh, ReLU
            'graph_state_dropout_keep_prob': 1.,
            'task_sample_ratios': {},
            'edge_weight_dropout_keep_prob': .8
        })
        return params

    # ---- weights ------------------------------------------------------------------------------------
    def prepare_specific_graph_model(self) -> None:
        """chem_tensorflow_sparse.py:63-115.  Placeholders become dict slots filled by feed(); the
        variables are created here with the reference's shapes and initialisers."""
        h_dim = self.params['hidden_size']
        # The reference accepts any hidden size and any number of residual inputs per layer (chem_tensorflow_sparse.py:46-50,
        # 139-145, 211-212).  Hidden sizes the kernels do not take as they are run zero-padded to ops.kernel_width(h_dim) (states
        # and weight blocks; the variables keep the reference's shapes); layers with more than 2 residual inputs run the generic
        # two-launch GRU instead of the single-launch kernels (up to 6 residual inputs: 8 K segments with the messages and h).
        if h_dim <= 0:
            raise ValueError("hidden_size %r must be positive" % (h_dim,))
        self._kw = ops.kernel_width(h_dim)
        self._padded_cache = {}
        if self.annotation_size > h_dim:
            raise ValueError("annotation_size %d exceeds hidden_size %d" % (self.annotation_size, h_dim))
        for layer_idx in range(len(self.params['layer_timesteps'])):
            res = self.params['residual_connections'].get(str(layer_idx)) or []
            if len(res) + 1 > ops.GRU_MAX_INPUTS:
                raise ValueError("layer %d has %d residual inputs; the GRU kernels take at most %d (plus the aggregated "
                                 "messages)" % (layer_idx, len(res), ops.GRU_MAX_INPUTS - 1))
            for r in res:
                if not 0 <= int(r) <= layer_idx:
                    raise ValueError("layer %d: residual connection %r refers to a layer that is not computed yet" % (layer_idx, r))
        for name in ('initial_node_representation', 'adjacency_lists', 'num_incoming_edges_per_type',
                     'graph_nodes_list', 'message_index'):
            self.placeholders[name] = None
        self.placeholders['graph_state_keep_prob'] = 1.0
        self.placeholders['edge_weight_dropout_keep_prob'] = 1.0

        activation_name = self.params['graph_rnn_activation'].lower()
        if activation_name not in ('tanh', 'relu'):
            raise Exception("Unknown activation function type '%s'." % activation_name)
        cell_type = self.params['graph_rnn_cell'].lower()
        if cell_type not in CELL_SPECS:
            raise Exception("Unknown RNN cell type '%s'." % cell_type)
        if cell_type == 'cudnncompatiblegrucell':
            assert (activation_name == 'tanh')                                  # :106
        self.cell_type = cell_type

        # Generate per-layer values for edge weights, biases and gated units:
        self.gnn_weights = GGNNWeights([], [], [], [])
        self._edge_weight_vars: List[torch.Tensor] = []
        dev = self.device
        for layer_idx in range(len(self.params['layer_timesteps'])):
            # :88-90 glorot over [T*D, D], then viewed as [T,D,D]
            ew = torch.from_numpy(glorot_init([self.num_edge_types * h_dim, h_dim])).to(dev)
            self._edge_weight_vars.append(ew)
            self.gnn_weights.edge_weights.append(ew.view(self.num_edge_types, h_dim, h_dim))
            if self.params['use_propagation_attention']:                        # :94-96
                self.gnn_weights.edge_type_attention_weights.append(
                    torch.ones([self.num_edge_types], dtype=torch.float32, device=dev))
            if self.params['use_edge_bias']:
                self.gnn_weights.edge_biases.append(torch.zeros([self.num_edge_types, h_dim], dtype=torch.float32, device=dev))
            res = self.params['residual_connections'].get(str(layer_idx)) or []
            in_dim = h_dim * (len(res) + 1)
            # TF-1.3 cells: kernels glorot_uniform (the variable scope's default initializer); GRU gate bias 1.0,
            # every other bias 0 (shapes: CELL_SPECS)
            cls, fields = CELL_SPECS[cell_type]
            tensors = []
            for (_, _, shape_fn, init) in fields:
                shape = shape_fn(in_dim, h_dim)
                if init == 'glorot':
                    tensors.append(tf_glorot_uniform(list(shape), self.tf_generator).to(dev))
                else:
                    tensors.append(torch.full(shape, float(init), dtype=torch.float32, device=dev))
            self.gnn_weights.rnn_cells.append(cls(*tensors))

    def graph_model_variables(self) -> Dict[str, torch.Tensor]:
        """graph_model/* variables under the names TF-1.3 gives them (used by the pickle checkpoints,
        chem_tensorflow.py:309-359)."""
        out = {}
        for l in range(len(self.params['layer_timesteps'])):
            scope = "graph_model/gnn_layer_%i" % l
            out["%s/gnn_edge_weights_%i:0" % (scope, l)] = self._edge_weight_vars[l]
            if self.params['use_edge_bias']:
                out["%s/gnn_edge_biases_%i:0" % (scope, l)] = self.gnn_weights.edge_biases[l]
            if self.params['use_propagation_attention']:
                out["%s/edge_type_attention_weights_%i:0" % (scope, l)] = self.gnn_weights.edge_type_attention_weights[l]
            cell = self.gnn_weights.rnn_cells[l]
            for (_, suffix, _, _), t in zip(CELL_SPECS[self.cell_type][1], cell):
                out["%s/timestep_0/%s" % (scope, suffix)] = t
        return out

    def set_graph_weights(self, layers: Sequence[dict]) -> None:
        """Inject explicit weights (oracle layout: 'edge_weights' [T,D,D], optional 'edge_biases' [T,D],
        'Wg','bg','Wc','bc') -- parity tests need this because TF's RNG is not reproducible."""
        with torch.no_grad():
            for l, L in enumerate(layers):
                as_t = lambda a: torch.as_tensor(np.asarray(a, dtype=np.float32)).to(self.device)
                self._edge_weight_vars[l].copy_(as_t(L['edge_weights']).reshape(self._edge_weight_vars[l].shape))
                if self.params['use_edge_bias']:
                    self.gnn_weights.edge_biases[l].copy_(as_t(L['edge_biases']))
                if self.params['use_propagation_attention']:
                    self.gnn_weights.edge_type_attention_weights[l].copy_(as_t(L['edge_type_attention_weights']))
                cell = self.gnn_weights.rnn_cells[l]
                for (key, _, _, _), t in zip(CELL_SPECS[self.cell_type][1], cell):
                    t.copy_(as_t(L[key]))

    # ---- parameters at the kernel width -----------------------------------------------------------------------
    @staticmethod
    def _pad_blocks(W: torch.Tensor, D: int, Dk: int) -> torch.Tensor:
        """[rb*D, cb*D] (or [cb*D]) -> [rb*Dk, cb*Dk] ([cb*Dk]): every D x D block (D-vector) zero-padded to Dk.  Differentiable
        (the gradient of a padded tensor is sliced back to the variable's shape)."""
        import torch.nn.functional as F
        if W.dim() == 1:
            cb = W.shape[0] // D
            return F.pad(W.view(cb, D), (0, Dk - D)).reshape(cb * Dk)
        rb, cb = W.shape[0] // D, W.shape[1] // D
        return F.pad(W.view(rb, D, cb, D), (0, Dk - D, 0, 0, 0, Dk - D)).reshape(rb * Dk, cb * Dk)

    def _kernel_layer(self, layer_idx: int, need_grad: bool):
        """(edge weight variable viewed [T, Dk, Dk], edge biases [T, Dk] | None, attention weights | None, cell tensors) of one
        layer at the kernel width Dk = ops.kernel_width(hidden_size).  Dk == hidden_size: the variables themselves.  Otherwise
        zero-padded copies: rebuilt with autograd when training, cached per weight version otherwise (so that the packed LDS
        images derived from them are reused from batch to batch)."""
        D, Dk, T = self.params['hidden_size'], self._kw, self.num_edge_types
        ew_var = self._edge_weight_vars[layer_idx]
        eb = self.gnn_weights.edge_biases[layer_idx] if self.params['use_edge_bias'] else None
        attn = self.gnn_weights.edge_type_attention_weights[layer_idx] if self.params['use_propagation_attention'] else None
        cell = self.gnn_weights.rnn_cells[layer_idx]
        if Dk == D:
            return ew_var.view(T, D, D), eb, attn, cell
        src = [ew_var] + ([eb] if eb is not None else []) + list(cell)
        versions = tuple(t._version for t in src)
        if not need_grad:
            hit = self._padded_cache.get(layer_idx)
            if hit is not None and hit[0] == versions:
                return hit[1]
        import torch.nn.functional as F
        ew = self._pad_blocks(ew_var, D, Dk).view(T, Dk, Dk)
        ebp = None if eb is None else F.pad(eb, (0, Dk - D))
        out = (ew, ebp, attn, type(cell)(*[self._pad_blocks(t, D, Dk) for t in cell]))
        if not need_grad:
            self._padded_cache[layer_idx] = (versions, out)
        return out

    # ---- operand format of the fused GRU forward, per layer (formats.py) ------------------------------------------------
    def gru_formats(self, h0: torch.Tensor, ew_keep: float = 1.0, st_keep: float = 1.0, training: bool = False) -> List[int]:
        """formats.F16X2 for the layers whose GRU operands are PROVABLY inside the two-piece f16 format's range for this batch and
        these weights, formats.BF16X3 (exact, every input) for the others -- the reference multiplies in plain f32
        (chem_tensorflow_sparse.py:215-216).  See formats.py for the bounds.  The same proof decides the operand format of each
        layer's compacted message transform (:160-164; its operands are the states and the edge weights, whatever the
        aggregation): self.last_edge_formats.  The decision of the last call is kept in self.last_gru_formats /
        self.last_edge_formats / self.last_gru_format_bounds (bench.py and the tests report it)."""
        from . import formats
        p = self.params
        L = len(p['layer_timesteps'])
        pol = formats.policy()
        plain_gru = self.cell_type == 'gru' and not p['use_propagation_attention']
        if not formats.split_path() or pol == "exact":
            fm, em = [formats.BF16X3] * L, [formats.BF16X3] * L
        elif pol == "force2":
            fm, em = [formats.F16X2] * L, [formats.F16X2] * L
        elif p['graph_rnn_activation'].lower() != 'tanh' or not plain_gru:
            fm, em = [formats.BF16X3] * L, [formats.BF16X3] * L     # no bound on the states (ReLU), or not the plain GRU path
            self.last_gru_format_bounds = {"proven": False, "why": "relu cell / variant cell: no bound on the states"}
        else:
            per_layer = []
            for l in range(L):
                cell = self.gnn_weights.rnn_cells[l]
                ts = [self._edge_weight_vars[l], cell.gates_kernel, cell.candidate_kernel]
                if p['use_edge_bias']:
                    ts.append(self.gnn_weights.edge_biases[l])
                per_layer.append(ts)
            flat = [t for ts in per_layer for t in ts]
            if training and getattr(self, "optimizer", None) is not None and getattr(self.optimizer, "fused", False):
                if getattr(self, "_train_weight_bounds", None) is None:
                    self._train_weight_bounds = formats.TrainingWeightBounds()
                maxima = self._train_weight_bounds.get(flat, self.optimizer)
            else:
                maxima = formats.weight_absmax(flat)
            h0_max = formats.h0_absmax(self.placeholders)            # (of the FED tensor; `h0` may be its zero-padded copy)
            S = formats.state_bound(h0_max, 'tanh', int(sum(p['layer_timesteps'])), st_keep)
            # (the mean's divisor bounds the sum only if the fed in-degree table covers the messages: declared by the packers, else checked)
            mi = self.placeholders.get('message_index')
            use_avg = bool(p['use_edge_msg_avg_aggregation']) and formats.nin_consistent(self.placeholders, getattr(mi, 'row_ptr', None))
            fm, em, i, inc_max, w_max = [], [], 0, 0.0, 0.0
            for l in range(L):
                ew, wg, wc = maxima[i], maxima[i + 1], maxima[i + 2]
                eb = maxima[i + 3] if p['use_edge_bias'] else 0.0
                i += len(per_layer[l])
                inc = formats.incoming_bound(S, p['hidden_size'], ew, eb, use_avg, ew_keep)      # (sum aggregation: inf)
                fm.append(formats.layer_format(S, inc, formats.nanmax(wg, wc)))
                # the transform's operands: the states (<= S) and the (weight-dropout-masked: / keep) edge weights
                em.append(formats.layer_format(S, 0.0, ew / min(max(ew_keep, 1e-30), 1.0)))
                inc_max, w_max = formats.nanmax(inc_max, inc), formats.nanmax(w_max, wg, wc)
            self.last_gru_format_bounds = {"proven": all(f == formats.F16X2 for f in fm), "h0_absmax": h0_max, "state_bound": S,
                                           "incoming_bound": inc_max, "gru_weight_absmax": w_max,
                                           "limits": {"activation": formats.MAX_ACTIVATION, "weight": formats.MAX_WEIGHT}}
        self.last_gru_formats, self.last_edge_formats = fm, em
        return fm

    # ---- the hot path -----------------------------------------------------------------------------------
    def compute_final_node_representations(self) -> torch.Tensor:
        """chem_tensorflow_sparse.py:117-218."""
        from .autograd import propagation_step
        ph = self.placeholders
        h0 = ph['initial_node_representation']
        h_dim, Dk = self.params['hidden_size'], self._kw
        if h0.shape[1] != Dk:                                                     # (see ops.kernel_width: zero columns stay zero)
            import torch.nn.functional as F
            h0 = F.pad(h0, (0, Dk - h0.shape[1])).contiguous()
        node_states_per_layer = [h0]                                              # :118-119
        index = ph.get('message_index')
        if index is None:                                                         # :120-129
            index = ops.build_message_index(ph['adjacency_lists'], h0.shape[0])
            ph['message_index'] = index
        nin = ph['num_incoming_edges_per_type']
        use_avg = bool(self.params['use_edge_msg_avg_aggregation'])
        act = self.params['graph_rnn_activation']
        ew_keep = float(ph.get('edge_weight_dropout_keep_prob', 1.0))
        st_keep = float(ph.get('graph_state_keep_prob', 1.0))
        need_grad = self.training and torch.is_grad_enabled()
        variant = self.params['use_propagation_attention'] or self.cell_type != 'gru'
        gru_fmts = self.gru_formats(h0, ew_keep, st_keep, need_grad)

        if not variant and not need_grad and ew_keep >= 1.0 and st_keep >= 1.0 and ops._timing is None:
            # inference: the whole layer/timestep loop below runs inside ONE native call
            final = self._propagate_native(h0, index, nin, use_avg, act, gru_fmts)
            return final if Dk == h_dim else final[:, :h_dim].contiguous()

        for (layer_idx, num_timesteps) in enumerate(self.params['layer_timesteps']):   # :131
            layer_residual_connections = self.params['residual_connections'].get(str(layer_idx))   # :140
            if layer_residual_connections is None:
                layer_residual_states = []
            else:
                layer_residual_states = [node_states_per_layer[residual_layer_idx]
                                         for residual_layer_idx in layer_residual_connections]
            ew_var, edge_biases, attn_w, cell = self._kernel_layer(layer_idx, need_grad)
            # :91 one weight-dropout mask per layer per run, shared by the layer's timesteps
            ew_mask = None
            if ew_keep < 1.0:
                ew_mask = (ew_keep, self.dropout_seed('edge_weights', layer_idx))
            plain_step = not variant
            if ew_mask is None:
                edge_weights = ew_var
            elif Dk != h_dim:
                # padded width: the mask is drawn on the VARIABLE (reference shape), the masked weights are padded
                edge_weights = self._pad_blocks(tf_dropout(self._edge_weight_vars[layer_idx], ew_mask[0], ew_mask[1]), h_dim,
                                                Dk).view(self.num_edge_types, Dk, Dk)
                ew_mask = None
            elif plain_step and need_grad:
                # (training on the default cell: the propagation step gets the variable AND the mask, so that its weight
                # gradient can be accumulated unmasked and masked once per layer -- backward.PropagationStepFn)
                edge_weights = ew_var
            else:
                edge_weights = tf_dropout(ew_var, ew_mask[0], ew_mask[1])
            cur = node_states_per_layer[-1]                                        # :152
            for step in range(num_timesteps):                                      # :153
                if variant and need_grad:
                    # non-default switches, training: HIP forward, autograd-derived backward (variants.py)
                    from .variants import variant_step
                    cur = variant_step(cur, index, nin, edge_weights, edge_biases, attn_w, use_avg, layer_residual_states,
                                       self.cell_type, tuple(cell), act)
                elif variant:
                    cur = self._variant_step(cur, index, nin, edge_weights.contiguous(), edge_biases, attn_w, use_avg,
                                             layer_residual_states, cell, act)
                else:
                    cur = propagation_step(cur, index, nin, edge_weights, edge_biases, use_avg,
                                           layer_residual_states, cell, act, need_grad,
                                           ew_mask if (need_grad and plain_step) else None, gru_fmt=gru_fmts[layer_idx],
                                           edge_fmt=ops.GRU_FMT_EXACT if need_grad else self.last_edge_formats[layer_idx])
                if st_keep < 1.0:                                                  # :113-114 DropoutWrapper(state)
                    cur = tf_dropout(cur, st_keep, self.dropout_seed('state', layer_idx, step), self._node_uid())
            node_states_per_layer.append(cur)
        final = node_states_per_layer[-1]                                          # :218
        return final if Dk == h_dim else final[:, :h_dim].contiguous()

    def _node_uid(self) -> Optional[torch.Tensor]:
        """int64 [V]: (dataset graph id << 20) + node index within its graph -- the row keys of the state-dropout mask, so
        that a node's mask does not depend on the batch (or the data-parallel shard) the node was packed into.  Needs the
        packers' 'graph_ids'; a foreign feed without them gets row-index keys (None)."""
        ph = self.placeholders
        uid = ph.get('node_uid')
        if uid is None and ph.get('graph_ids') is not None and ph.get('graph_ptr') is not None:
            gnl = ph['graph_nodes_list'].long()
            V = gnl.shape[0]
            first = ph['graph_ptr'].long()[gnl]
            uid = (ph['graph_ids'].long()[gnl] << 20) + (torch.arange(V, device=gnl.device) - first)
            ph['node_uid'] = uid = uid.contiguous()
        return uid

    def _variant_step(self, h, index, nin, edge_weights, edge_biases, attn_w, use_avg, residual_states, cell, act):
        """One timestep with the non-default switches of chem_tensorflow_sparse.py: propagation attention
        (:147-149, 170-196) and/or the BasicRNNCell / CudnnCompatibleGRUCell cells (:105-110).  Dense transform."""
        H = ops.msg_transform(h, edge_weights)
        if self.params['use_propagation_attention']:
            incoming = ops.gather_segment_sum_attn(H, h, index, attn_w, nin, edge_biases, use_avg)
        else:
            incoming = ops.gather_segment_sum(H, index, nin, edge_biases, use_avg)
        xs = list(residual_states) + [incoming]
        if self.cell_type == 'gru':
            return ops.gru(xs, h, cell.gates_kernel, cell.gates_bias, cell.candidate_kernel, cell.candidate_bias, act)
        if self.cell_type == 'rnn':
            return ops.rnn(xs, h, cell.kernel, cell.bias, act)
        return ops.cudnn_gru(xs, h, *cell)

    def _propagate_native(self, h0, index, nin, use_avg, act, gru_fmts) -> torch.Tensor:
        """compute_final_node_representations through ggnn_sparse_propagate_f32 (the loop of :131-218 in C):
        source-compacted transform and pre-packed weight images where the hidden size supports them."""
        from .autograd import USE_COMPACT_TRANSFORM, _PACKED
        D, T = self._kw, self.num_edge_types
        L = len(self.params['layer_timesteps'])
        comp = None
        if USE_COMPACT_TRANSFORM and ops.compact_supported(D):
            comp = getattr(index, "_compact", None)
            if comp is None:
                comp = index._compact = ops.build_compact_sources(index)
        layers = [self._kernel_layer(l, False) for l in range(L)]
        edge_w = [lay[0].contiguous() for lay in layers]
        edge_fmts = list(self.last_edge_formats)
        edge_packed = [_PACKED.edge(w, edge_fmts[l]) for l, w in enumerate(edge_w)] if comp is not None else None
        edge_bias = [lay[1] for lay in layers] if self.params['use_edge_bias'] else None
        cells = [lay[3] for lay in layers]
        residuals = [self.params['residual_connections'].get(str(l)) or [] for l in range(L)]
        gru_packed = None
        if ops.gru_is_fused(D):
            # (layers with more inputs than the single-launch kernels take run the generic GRU on the raw weights)
            gru_packed = [_PACKED.gru(c.gates_kernel, c.candidate_kernel, len(residuals[l]) + 1, D, gru_fmts[l])
                          if len(residuals[l]) + 1 <= ops.GRU_FUSED_MAX_INPUTS else None for l, c in enumerate(cells)]
        outs = ops.sparse_propagate(h0, index, comp, nin, use_avg, self.params['layer_timesteps'], residuals,
                                    edge_w, edge_packed, edge_bias,
                                    [c.gates_kernel for c in cells], [c.gates_bias for c in cells],
                                    [c.candidate_kernel for c in cells], [c.candidate_bias for c in cells],
                                    gru_packed, act, gru_fmt=gru_fmts, edge_fmt=edge_fmts)
        return outs[-1]

    def _graph_nodes_sorted(self) -> bool:
        """The fused readout sums a graph's nodes as one segment: graph_nodes_list must be non-decreasing.  Batches of this
        package's packers say so ('graph_nodes_sorted'); a foreign feed is checked once per list object (one sync)."""
        ph = self.placeholders
        if ph.get('graph_nodes_sorted') is not None:
            return bool(ph['graph_nodes_sorted'])
        gnl = ph['graph_nodes_list']
        cached = getattr(self, '_gnl_checked', None)
        if cached is None or cached[0] is not gnl:
            ok = bool(gnl.numel() < 2 or bool((gnl[1:] >= gnl[:-1]).all().item()))
            self._gnl_checked = cached = (gnl, ok)
        return cached[1]

    def gated_regression_with_loss(self, last_h, regression_gate, regression_transform, target_values, target_mask):
        """gated_regression (chem_tensorflow_sparse.py:220-231) and the masked sums of chem_tensorflow.py:161-166 in one
        fused, differentiable, deterministic unit (autograd.readout_loss).  None when the fused kernels do not apply."""
        from .autograd import readout_loss
        D, Dk = self.params['hidden_size'], self._kw
        if not last_h.is_cuda or Dk > 256 or not self._graph_nodes_sorted():
            return None
        g, t = regression_gate.params, regression_transform.params
        if len(g["weights"]) != 1 or len(t["weights"]) != 1:
            return None
        ph = self.placeholders
        h0 = ph['initial_node_representation']
        gW, tW = regression_gate.dropped_weight(0), regression_transform.dropped_weight(0)       # utils.py:68 dropout on W
        if Dk != D:
            # hidden sizes the kernels run zero-padded (ops.kernel_width): states and the [hT | h0] / hT weight vectors at width Dk
            import torch.nn.functional as F
            last_h = F.pad(last_h, (0, Dk - D))
            h0 = h0 if h0.shape[1] == Dk else F.pad(h0, (0, Dk - h0.shape[1]))
            gW, tW = self._pad_blocks(gW.reshape(-1), D, Dk), self._pad_blocks(tW.reshape(-1), D, Dk)
        out, num, ab, ms = readout_loss(last_h, h0.contiguous(), ph['graph_nodes_list'], ph.get('graph_ptr'), None,
                                        ph['num_graphs'], gW, g["biases"][0], tW, t["biases"][0],
                                        target_values.contiguous(), target_mask.contiguous())
        self.output = out
        return out, num, ab, ms

    def gated_regression(self, last_h, regression_gate, regression_transform):
        """chem_tensorflow_sparse.py:220-231."""
        from .autograd import segment_sum_rows
        keep = float(self.placeholders.get('out_layer_dropout_keep_prob', 1.0))
        D, Dk = self.params['hidden_size'], self._kw
        h0 = self.placeholders['initial_node_representation']
        if not (self.training and torch.is_grad_enabled()) and keep >= 1.0 and last_h.is_cuda:
            # inference: one fused HIP pass (no [V,2D] concat, no per-node intermediates); deterministic segmented sum for
            # batcher output, the atomic form only for an unsorted graph_nodes_list
            g, t = regression_gate.params, regression_transform.params
            ph = self.placeholders
            gW, tW = g["weights"][0].reshape(-1), t["weights"][0].reshape(-1)
            hT = last_h.contiguous()
            if Dk != D:                                                             # zero-padded width, see ops.kernel_width
                import torch.nn.functional as F
                hT = F.pad(hT, (0, Dk - D))
                h0 = (h0 if h0.shape[1] == Dk else F.pad(h0, (0, Dk - h0.shape[1]))).contiguous()
                gW, tW = self._pad_blocks(gW, D, Dk), self._pad_blocks(tW, D, Dk)
            if self._graph_nodes_sorted() and Dk <= 256:
                output = ops.readout_loss_fwd(hT, h0, ph['graph_nodes_list'], ph.get('graph_ptr'), None, ph['num_graphs'], gW,
                                              g["biases"][0], tW, t["biases"][0], None, None)[0]
            else:
                output = ops.gated_readout(hT, h0, ph['graph_nodes_list'], ph['num_graphs'], gW.reshape(-1, 1), g["biases"][0],
                                           tW.reshape(-1, 1), t["biases"][0])
            self.output = output
            return output
        gate_input = torch.cat([last_h, h0[:, :D]], dim=-1)                                          # [v x 2h]
        gated_outputs = torch.sigmoid(regression_gate(gate_input)) * regression_transform(last_h)    # [v x 1]
        # Sum up all nodes per-graph
        graph_representations = segment_sum_rows(gated_outputs, self.placeholders['graph_nodes_list'],
                                                 self.placeholders['num_graphs'])                    # [g x 1]
        output = graph_representations.squeeze(-1)                                                   # [g]
        self.output = output
        return output

    # ---- data preprocessing and chunking into minibatches ------------------------------------------------
    def process_raw_graphs(self, raw_data, is_training_data: bool) -> Any:
        """chem_tensorflow_sparse.py:234-252.  Keeps the data tensorised (MoleculeSet) instead of a list
        of per-graph dicts; the adjacency lists / in-degree tables of :254-276 are built per batch by
        data.pack_batch.  Training data are shuffled once here (:243-244) and labels beyond
        task_sample_ratios are masked (:245-250)."""
        ms = raw_data if isinstance(raw_data, MoleculeSet) else MoleculeSet.from_json(raw_data)
        label_mask = np.ones((ms.num_graphs, ms.targets.shape[1]), dtype=np.float32)
        if is_training_data:
            perm = np.random.permutation(ms.num_graphs)
            ms = ms.subset(perm)
            for task_id in self.params['task_ids']:
                task_sample_ratio = self.params['task_sample_ratios'].get(str(task_id))
                if task_sample_ratio is not None:
                    ex_to_sample = int(ms.num_graphs * task_sample_ratio)
                    label_mask[ex_to_sample:, task_id] = 0.0
        return {"molecules": ms, "label_mask": label_mask, "device_batches": None}

    def to_device_batch(self, b: SparseBatch, compact: bool = True) -> Dict[str, Any]:
        """Upload one packed batch and build its message index (once; reused every epoch)."""
        dev = self.device
        t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
        adjacency = [t(a) for a in b.adjacency_lists]
        V, A = b.node_features.shape
        h0 = torch.zeros((V, b.hidden_size), dtype=torch.float32, device=dev)     # :300-302 zero-pad to D
        if V:
            h0[:, :A] = t(b.node_features)
        from . import formats
        return formats.declare_h0_absmax({
            'initial_node_representation': h0,
            'adjacency_lists': adjacency,
            'num_incoming_edges_per_type': t(b.num_incoming_edges_per_type),
            'graph_nodes_list': t(b.graph_nodes_list),
            'graph_ptr': t(np.concatenate([[0], np.cumsum(np.bincount(b.graph_nodes_list, minlength=b.num_graphs))]).astype(np.int32)),
            'target_values': t(b.target_values),
            'target_mask': t(b.target_mask),
            'num_graphs': b.num_graphs,
            'message_index': ops.prepare_message_index(ops.build_message_index(adjacency, V), b.hidden_size, compact),
            'graph_nodes_sorted': True,
            'graph_ids': None if b.extras.get("graph_ids") is None else t(np.asarray(b.extras["graph_ids"], dtype=np.int64)),
        }, float(np.abs(b.node_features).max()) if b.node_features.size else 0.0)

    def prepare_resident_data(self, data: Any, is_training: bool) -> None:
        """Upload the dataset and build the dataset-level tables of the device packer on the CURRENT stream (run_epoch calls this
        before it hands the epoch to the producer thread, whose stream is ordered behind it)."""
        if not bool(self.params.get('pack_on_device', True)) or data is None:
            return
        if data.get("molecules_dev") is None:
            data["molecules_dev"] = DeviceMoleculeSet(data["molecules"], self.device, data["label_mask"])
        from . import backward, data_device
        if data_device.USE_STATIC_TABLES:
            compact = ((not is_training) or backward.USE_COMPACT_TRANSFORM) and ops.compact_supported(self.params['hidden_size'])
            tie = self.params.get("tie_fwd_bkwd", True)
            data["molecules_dev"].static_tables(self.num_edge_types, tie, compact)
            if is_training and compact:
                data["molecules_dev"].static_backward_tables(self.num_edge_types, tie)

    def make_minibatch_iterator(self, data: Any, is_training: bool):
        """chem_tensorflow_sparse.py:278-350: minibatches as one disconnected super-graph each.
        Batches are assembled on the GPU from the resident dataset (data_device.py; params['pack_on_device']=False
        selects the NumPy packer data.pack_batches + upload, same results); validation batches are packed once and
        stay resident in HBM; training batches are re-packed per epoch after the shuffle (:281-282)."""
        ms: MoleculeSet = data["molecules"]
        state_dropout_keep_prob = self.params['graph_state_dropout_keep_prob'] if is_training else 1.
        edge_weights_dropout_keep_prob = self.params['edge_weight_dropout_keep_prob'] if is_training else 1.
        rank = self.dist.rank if self.dist is not None else 0
        world = self.dist.world_size if self.dist is not None else 1
        on_device = bool(self.params.get('pack_on_device', True))
        if on_device and data.get("molecules_dev") is None:
            # the dataset goes to HBM once; batches are then assembled on the GPU from graph ids (data_device.py)
            data["molecules_dev"] = DeviceMoleculeSet(ms, self.device, data["label_mask"])

        # the compacted-transform pair list is part of the batch unless this epoch trains on the dense transform
        from . import backward
        compact = (not is_training) or backward.USE_COMPACT_TRANSFORM

        # (a hidden size the kernels run zero-padded: the annotations are zero-padded anyway, :300-302, so the batches are packed at
        # the kernel width right away -- 'initial_node_representation' is then [V, ops.kernel_width(hidden_size)])
        pack_params = self.params if self._kw == self.params['hidden_size'] else dict(self.params, hidden_size=self._kw)

        def epoch_batches(order):
            if on_device:
                return pack_batches_device(data["molecules_dev"], pack_params, self.num_edge_types, order, rank, world, compact,
                                           training=is_training and compact)
            return (self.to_device_batch(b, compact) for b in
                    pack_batches(ms, pack_params, self.num_edge_types, order, data["label_mask"], rank, world))

        if is_training:
            # :281-282 np.random.shuffle(data) shuffles the reference's graph list IN PLACE, so the orders of successive
            # epochs compose; permutation(n) draws the same swaps as shuffle(list of n).  Same seed on every rank.
            perm = np.random.permutation(ms.num_graphs)
            prev = data.get("epoch_order")
            order = perm if prev is None else prev[perm]
            data["epoch_order"] = order
            device_batches = epoch_batches(order)
        else:
            if data["device_batches"] is None:
                data["device_batches"] = list(epoch_batches(None))
            device_batches = data["device_batches"]
        for db in device_batches:
            feed = dict(db)
            feed['graph_state_keep_prob'] = state_dropout_keep_prob
            feed['edge_weight_dropout_keep_prob'] = edge_weights_dropout_keep_prob
            yield feed

    def forward_dataset(self, data: Any, num_streams: int = 2, feed_hook=None, consumer_streams=None):
        """Inference over a whole dataset with every batch assembled fresh on the GPU, pipelined: batch i+1.. are packed on
        high-priority side streams under batch i's forward, the forwards alternate over `num_streams` compute streams
        (utils.StreamPrefetcher; the reference overlaps its host-side packer with sess.run through a producer thread,
        chem_tensorflow.py:219).  Yields (feed, final node representations, stream): the states are valid on `stream` -- work
        queued there sees them; to read them from the host, synchronise it.  Streams are kept for later calls.
        feed_hook(feed): called on the consumer stream before the forward (bench.py swaps in dense random initial states).
        consumer_streams: compute streams to use instead of the model's own.  (A process should stay frugal with streams: the
        runtime multiplexes them onto a handful of hardware queues -- GPU_MAX_HW_QUEUES, 4 by default; with 1 or 2 every
        multi-stream number of bench.py falls to its one-stream value -- and which streams end up sharing a queue is not under
        the program's control: with bench.py's two headline streams still alive, two MORE compute streams plus two packing
        streams ran this pipeline at exactly the one-stream rate; re-using the two existing ones runs it at full rate.)"""
        from .utils import StreamPrefetcher
        self.prepare_resident_data(data, False)
        pipe = getattr(self, '_pipeline_streams', None)
        if pipe is None or (consumer_streams is None and len(pipe[0]) != num_streams):
            pipe = self._pipeline_streams = ([] if consumer_streams is not None else [torch.cuda.Stream(self.device) for _ in range(num_streams)],
                                             [torch.cuda.Stream(self.device, priority=-1) for _ in range(2)])
        if consumer_streams is not None:
            pipe = (list(consumer_streams), pipe[1])
        pack_params = self.params if self._kw == self.params['hidden_size'] else dict(self.params, hidden_size=self._kw)
        gen = pack_batches_device(data["molecules_dev"], pack_params, self.num_edge_types, None, 0, 1, True)
        cur = torch.cuda.current_stream(self.device)
        for st in pipe[0]:
            st.wait_stream(cur)
        # (the generator computes inside no_grad / the consumer stream's context but YIELDS outside them: a suspended generator must not
        # leave its caller with grad mode off and another stream current -- round-3 advisor finding)
        for feed, st in StreamPrefetcher(gen, self.device, consumer_streams=pipe[0], pack_streams=pipe[1]):
            with torch.no_grad(), torch.cuda.stream(st):
                if feed_hook is not None:
                    feed_hook(feed)
                self.feed(feed)
                states = self.compute_final_node_representations()
            yield feed, states, st

    def evaluate_one_batch(self, data):
        """chem_tensorflow_sparse.py:352-362."""
        outs = []
        for item in self.make_minibatch_iterator(data, is_training=False):
            item['graph_state_keep_prob'] = 1.0
            item['edge_weight_dropout_keep_prob'] = 1.0
            item['out_layer_dropout_keep_prob'] = 1.0
            with torch.no_grad():
                self.forward_batch(item)
            outs.append(self.output.detach().cpu().numpy())
            print(outs[-1])
        return outs

    def example_evaluation(self, molecules_file: str = 'molecules_valid.json'):
        """chem_tensorflow_sparse.py:364-376: first 10 validation molecules, predictions next to targets."""
        import json
        n_example_molecules = 10
        with open(molecules_file, 'r') as valid_file:
            example_molecules = json.load(valid_file)[:n_example_molecules]
        for mol in example_molecules:
            print(mol['targets'])
        example_molecules = self.process_raw_graphs(example_molecules, is_training_data=False)
        return self.evaluate_one_batch(example_molecules)

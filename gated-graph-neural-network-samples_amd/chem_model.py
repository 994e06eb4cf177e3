"""ChemModel -- host-side mirror of the reference's plugin base class (chem_tensorflow.py:16-359).

Same surface (default_params, load_data, process_raw_graphs, make_model, make_train_step,
gated_regression, prepare_specific_graph_model, compute_final_node_representations,
make_minibatch_iterator, run_epoch, train, save_progress, restore_progress), same parameter keys and
defaults, same loss / metric / clipping / Adam arithmetic -- but eager on PyTorch-ROCm tensors with the
hot path running in libggnn_hip.so instead of a TF-1.x session.  `self.placeholders` is a dict
name -> tensor that `feed()` fills from a minibatch (the reference's feed_dict); `self.ops` holds the
values of the last executed batch under the reference's op names.

Differences that are deliberate:
  * args is a plain dict with the reference's docopt keys ('--config', '--config-file',
    '--data_dir', '--log_dir', '--restore', '--freeze-graph-model', '--evaluate'); extra keys:
    'train_data' / 'valid_data' (in-memory MoleculeSet or raw JSON list instead of files),
    '--device' (default 'cuda:0'), '--quiet' (no log files), 'dist' (a DataParallelContext).
  * no TensorBoard summaries (chem_tensorflow.py:195-200).
"""
from __future__ import annotations

import json
import os
import pickle
import random
import time
from typing import Any, Dict, List, Optional, Sequence

import numpy as np
import torch

from .data import MoleculeSet
from .utils import MLP, SMALL_NUMBER


class ChemModel(object):
    @classmethod
    def default_params(cls):
        # chem_tensorflow.py:18-37
        return {
            'num_epochs': 3000,
            'patience': 25,
            'learning_rate': 0.001,
            'clamp_gradient_norm': 1.0,
            'out_layer_dropout_keep_prob': 1.0,

            'hidden_size': 100,
            'num_timesteps': 4,
            'use_graph': True,

            'tie_fwd_bkwd': True,
            'task_ids': [0],

            'random_seed': 0,

            'train_file': 'molecules_train.json',
            'valid_file': 'molecules_valid.json'
        }

    def __init__(self, args):
        self.args = args
        self.quiet = bool(args.get('--quiet'))
        self.device = torch.device(args.get('--device') or 'cuda:0')
        self.dist = args.get('dist')

        # Collect argument things (chem_tensorflow.py:42-54):
        data_dir = ''
        if '--data_dir' in args and args['--data_dir'] is not None:
            data_dir = args['--data_dir']
        self.data_dir = data_dir
        self.run_id = "_".join([time.strftime("%Y-%m-%d-%H-%M-%S"), str(os.getpid())])
        log_dir = args.get('--log_dir') or '.'
        self.log_file = os.path.join(log_dir, "%s_log.json" % self.run_id)
        self.best_model_file = os.path.join(log_dir, "%s_model_best.pickle" % self.run_id)

        # Collect parameters (chem_tensorflow.py:56-68): defaults < --config-file < --config
        params = self.default_params()
        config_file = args.get('--config-file')
        if config_file is not None:
            with open(config_file, 'r') as f:
                params.update(json.load(f))
        config = args.get('--config')
        if config is not None:
            params.update(json.loads(config) if isinstance(config, str) else config)
        self.params = params
        if not self.quiet:
            os.makedirs(log_dir, exist_ok=True)
            with open(os.path.join(log_dir, "%s_params.json" % self.run_id), "w") as f:
                json.dump(params, f)
            print("Run %s starting with following parameters:\n%s" % (self.run_id, json.dumps(self.params)))
        random.seed(params['random_seed'])
        np.random.seed(params['random_seed'])
        torch.manual_seed(params['random_seed'])
        self.tf_generator = torch.Generator().manual_seed(params['random_seed'])   # stands in for tf.set_random_seed (:85)

        # Load data (chem_tensorflow.py:72-77):
        self.max_num_vertices = 0
        self.num_edge_types = 0
        self.annotation_size = 0
        self.train_data = self.load_data(args.get('train_data', params['train_file']), is_training_data=True)
        self.valid_data = self.load_data(args.get('valid_data', params['valid_file']), is_training_data=False)

        # Build the actual model (chem_tensorflow.py:79-91)
        self.placeholders: Dict[str, Any] = {}
        self.weights: Dict[str, Any] = {}
        self.ops: Dict[str, Any] = {}
        self.training = False
        self.dropout_step = 0                    # optimisation steps taken: part of every dropout mask's key (dropout_seed)
        self.make_model()
        self.make_train_step()

        # Restore/initialize variables (chem_tensorflow.py:93-100):
        restore_file = args.get('--restore')
        if restore_file is not None:
            self.train_step_id, self.valid_step_id = self.restore_progress(restore_file)
        else:
            self.train_step_id = 0
            self.valid_step_id = 0

    # ---- data ---------------------------------------------------------------------------------
    def load_data(self, source, is_training_data: bool):
        """chem_tensorflow.py:104-123.  `source` is a file name under data_dir, a raw JSON list or a
        MoleculeSet.  Derives num_edge_types (max bond id x (1 tied | 2 untied), :116-120) and
        annotation_size (:121) from the data, like the reference."""
        if source is None:
            return None
        if isinstance(source, MoleculeSet):
            ms = source
        elif isinstance(source, (list, tuple)):
            ms = MoleculeSet.from_json(source)
        else:
            full_path = os.path.join(self.data_dir, source)
            if not self.quiet:
                print("Loading data from %s" % full_path)
            ms = MoleculeSet.load(full_path)
        restrict = self.args.get("--restrict_data")
        if restrict is not None and restrict > 0:
            ms = ms.subset(np.arange(min(restrict, ms.num_graphs)))
        if ms.num_graphs:
            self.max_num_vertices = max(self.max_num_vertices, int(ms.nodes_per_graph().max()) - 1)
            self.num_edge_types = max(self.num_edge_types,
                                      ms.num_fwd_edge_types * (1 if self.params['tie_fwd_bkwd'] else 2))
            self.annotation_size = max(self.annotation_size, ms.annotation_size)
        return self.process_raw_graphs(ms, is_training_data)

    def process_raw_graphs(self, raw_data, is_training_data: bool) -> Any:
        raise Exception("Models have to implement process_raw_graphs!")

    # ---- model ----------------------------------------------------------------------------------
    def make_model(self):
        """chem_tensorflow.py:133-170 (weight creation part; the per-batch arithmetic is in
        run_batch)."""
        self.placeholders['target_values'] = None
        self.placeholders['target_mask'] = None
        self.placeholders['num_graphs'] = None
        self.placeholders['out_layer_dropout_keep_prob'] = 1.0
        self.prepare_specific_graph_model()
        for task_id in self.params['task_ids']:
            keep = lambda: self.placeholders['out_layer_dropout_keep_prob']
            self.weights['regression_gate_task%i' % task_id] = MLP(2 * self.params['hidden_size'], 1, [], keep,
                                                                   device=self.device)
            self.weights['regression_transform_task%i' % task_id] = MLP(self.params['hidden_size'], 1, [], keep,
                                                                        device=self.device)
            for kind in ('regression_gate', 'regression_transform'):
                self.weights['%s_task%i' % (kind, task_id)].dropout_seed = \
                    lambda layer, kind=kind, task_id=task_id: self.dropout_seed(kind, task_id, layer)

    def dropout_seed(self, *site) -> int:
        """Key of the dropout mask drawn at `site` in the current optimisation step: a hash of (random_seed, step, site), hence
        identical on every rank of a data-parallel job whatever else a rank draws (the reference's masks come from TF's
        graph-seeded stream, chem_tensorflow.py:85; under DP every rank must hold the same weight mask)."""
        from .utils import dropout_seed
        return dropout_seed(self.params['random_seed'], self.dropout_step, *site)

    def named_variables(self) -> Dict[str, torch.Tensor]:
        """All trainable tensors under TF-style variable names (chem_tensorflow.py:311-313 naming)."""
        out = dict(self.graph_model_variables())
        for task_id in self.params['task_ids']:
            for scope, key in (("regression_gate", 'regression_gate_task%i'), ("regression", 'regression_transform_task%i')):
                mlp = self.weights[key % task_id]
                for i, (W, b) in enumerate(zip(mlp.params["weights"], mlp.params["biases"])):
                    out["out_layer_task%i/%s/MLP_W_layer%i:0" % (task_id, scope, i)] = W
                    out["out_layer_task%i/%s/MLP_b_layer%i:0" % (task_id, scope, i)] = b
        return out

    def graph_model_variables(self) -> Dict[str, torch.Tensor]:
        return {}

    def forward_batch(self, batch_data: Dict[str, Any]):
        """The per-batch part of make_model (chem_tensorflow.py:141-170): final node representations,
        per-task gated regression, masked loss and MAE."""
        self.feed(batch_data)
        if self.params['use_graph']:
            final = self.compute_final_node_representations()
        else:
            final = torch.zeros_like(self.placeholders['initial_node_representation'])   # :147
        self.ops['final_node_representations'] = final
        self.ops['losses'] = []
        fused_readout = getattr(self, 'gated_regression_with_loss', None)
        for (internal_id, task_id) in enumerate(self.params['task_ids']):
            gate_mlp = self.weights['regression_gate_task%i' % task_id]
            transform_mlp = self.weights['regression_transform_task%i' % task_id]
            task_target_values = self.placeholders['target_values'][internal_id, :]
            task_target_mask = self.placeholders['target_mask'][internal_id, :]
            # models with a fused readout + loss kernel (one forward and one backward launch group on the GPU) return the
            # prediction together with the three masked sums of :161-166; None -> the op-by-op form below
            fused = fused_readout(final, gate_mlp, transform_mlp, task_target_values, task_target_mask) if fused_readout else None
            if fused is not None:
                computed_values, loss_num, abs_sum, mask_sum = fused
            else:
                computed_values = self.gated_regression(final, gate_mlp, transform_mlp)
                diff = computed_values - task_target_values                                   # :161
                diff = diff * task_target_mask                                                # :164
                loss_num, abs_sum, mask_sum = (0.5 * diff * diff).sum(), diff.abs().sum(), task_target_mask.sum()
            task_target_num = mask_sum + SMALL_NUMBER                                         # :163
            self.ops['accuracy_task%i' % task_id] = abs_sum / task_target_num                 # :165
            task_loss = loss_num / task_target_num                                            # :166
            # :168 looks the ratio up with an int key although configs carry str keys -> never applied
            task_loss = task_loss * (1.0 / (self.params['task_sample_ratios'].get(task_id) or 1.0))
            self.ops['losses'].append(task_loss)
            self.ops['loss_numerator_task%i' % task_id] = loss_num
            self.ops['abs_error_sum_task%i' % task_id] = abs_sum
            self.ops['loss_denominator_task%i' % task_id] = mask_sum
        self.ops['loss'] = torch.stack(self.ops['losses']).sum()                              # :170
        return self.ops['loss']

    def feed(self, batch_data: Dict[str, Any]) -> None:
        """The reference's feed_dict: every placeholder the batch carries is replaced.  Values DERIVED from a fed
        placeholder and cached next to it (the message index built from 'adjacency_lists', the dense model's sparse
        form of 'adjacency_matrix') are dropped unless the batch brings its own, so a reference-style feed dict can
        never run on the previous batch's index."""
        for src, derived in self.DERIVED_PLACEHOLDERS.items():
            if src in batch_data and batch_data[src] is not self.placeholders.get(src):   # (same object: cache stays valid)
                for d in derived:
                    if d not in batch_data:
                        self.placeholders[d] = None
        self.placeholders.update(batch_data)

    DERIVED_PLACEHOLDERS = {'adjacency_lists': ('message_index',), 'adjacency_matrix': ('_sparse_form',),
                            'initial_node_representation': ('h0_absmax', '_h0_absmax_of'),      # (formats.h0_absmax: measured once per fed h0)
                            'graph_nodes_list': ('graph_ptr', 'graph_nodes_sorted', 'graph_ids', 'node_uid')}

    def make_train_step(self):
        """chem_tensorflow.py:172-193: Adam(lr) on all trainable variables (minus graph_model/* when
        --freeze-graph-model), per-variable clip_by_norm."""
        from .train import TFAdam
        variables = self.named_variables()
        if self.args.get('--freeze-graph-model'):
            graph_vars = set(self.graph_model_variables().keys())
            for name in graph_vars:
                if not self.quiet:
                    print("Freezing weights of variable %s." % name)
            variables = {k: v for k, v in variables.items() if k not in graph_vars}
        self.trainable_variables = variables
        self.optimizer = TFAdam(list(variables.values()), lr=self.params['learning_rate'])

    def gated_regression(self, last_h, regression_gate, regression_transform):
        raise Exception("Models have to implement gated_regression!")

    def prepare_specific_graph_model(self) -> None:
        raise Exception("Models have to implement prepare_specific_graph_model!")

    def compute_final_node_representations(self) -> torch.Tensor:
        raise Exception("Models have to implement compute_final_node_representations!")

    def make_minibatch_iterator(self, data: Any, is_training: bool):
        raise Exception("Models have to implement make_minibatch_iterator!")

    # ---- training loop ----------------------------------------------------------------------------
    def train_batch(self, batch_data: Dict[str, Any]):
        """One optimisation step (the fetch of ops['train_step'], chem_tensorflow.py:231,183-191)."""
        from .train import train_step
        try:
            return train_step(self, batch_data)
        finally:
            self.dropout_step += 1

    def run_epoch(self, epoch_name: str, data, is_training: bool, start_step: int = 0):
        """chem_tensorflow.py:214-253."""
        chemical_accuracies = np.array([0.066513725, 0.012235489, 0.071939046, 0.033730778, 0.033486113, 0.004278493,
                                        0.001330901, 0.004165489, 0.004128926, 0.00409976, 0.004527465, 0.012292586,
                                        0.037467458])
        loss = 0
        accuracies = []
        start_time = time.time()
        processed_graphs = 0
        steps = 0
        sharded = self.dist is not None and self.dist.active
        shard_stats, shard_graphs = [], []
        pending = None
        batch_iterator = self.make_minibatch_iterator(data, is_training)
        threaded = self.params.get('threaded_batches', 'auto')
        if threaded == 'auto':
            # The producer thread pays when the launching thread is the bottleneck (the autograd path: ~4 ms of Python per step);
            # with the native step the host has slack, packing a batch inline costs 0.3 ms of it, and a second thread only takes
            # the interpreter lock away from the launches (5.8 vs 6.05 ms per step, tools/bench_extra.py epoch).
            from . import train_native
            threaded = not train_native.model_eligible(self)
        if is_training and threaded:
            # chem_tensorflow.py:219 ThreadedIterator(..., max_queue_size=5): the next batches are packed while this one trains
            # (validation batches are packed once and stay resident: nothing to prefetch).  Two ahead is enough here.
            from .utils import ThreadedIterator
            prepare = getattr(self, 'prepare_resident_data', None)
            if prepare is not None:
                prepare(data, is_training)               # resident dataset + tables on THIS stream, before the producer starts
            dev = getattr(self, "device", None)
            if dev is not None and torch.device(dev).type == "cuda" and torch.cuda.is_available() and getattr(self, '_packer_stream', None) is None:
                self._packer_stream = torch.cuda.Stream(dev)   # one packer stream for all epochs of this model
            batch_iterator = ThreadedIterator(batch_iterator, max_queue_size=2, device=dev, stream=getattr(self, '_packer_stream', None))
        for step, batch_data in enumerate(batch_iterator):
            num_graphs = batch_data['num_graphs']
            processed_graphs += num_graphs
            if is_training:
                batch_data['out_layer_dropout_keep_prob'] = self.params['out_layer_dropout_keep_prob']
                batch_loss = self.train_batch(batch_data)
            else:
                batch_data['out_layer_dropout_keep_prob'] = 1.0
                with torch.no_grad():
                    batch_loss = self.forward_batch(batch_data)
            if sharded:
                # Data parallel: this rank saw one shard of the step's global batch (the union of the ranks' batches).
                # Keep the loss / MAE numerators and the mask counts on the device; they are summed over the ranks once
                # per epoch (below), so that EVERY rank derives the same epoch statistics -- and hence the same
                # best-epoch / patience decisions: ranks that disagreed would leave the others hanging in the next
                # gradient all-reduce.
                shard_stats.append(torch.stack(
                    [self.ops[k % t].detach().to(torch.float64).reshape(())
                     for k in ('loss_numerator_task%i', 'loss_denominator_task%i', 'abs_error_sum_task%i')
                     for t in self.params['task_ids']]))
                shard_graphs.append(float(num_graphs))
                steps += 1
                continue
            # The step's loss and per-task MAE stay on the device until the NEXT step has been queued: reading them now would
            # stall the launching thread until the GPU has finished this step, and the GPU would then idle while the next
            # step's launches are issued ("Loss so far" therefore trails by one batch; the epoch result does not).
            stats = torch.stack([batch_loss.detach().reshape(()).to(torch.float64)] +
                                [self.ops['accuracy_task%i' % t].detach().reshape(()).to(torch.float64) for t in self.params['task_ids']])
            stats = self._readback(stats)
            if pending is not None:
                loss, processed_seen = self._absorb_step_stats(pending, loss, accuracies, epoch_name, step - 1)
            pending = (stats, num_graphs, processed_graphs)
            steps += 1
        if pending is not None:
            loss, _ = self._absorb_step_stats(pending, loss, accuracies, epoch_name, steps - 1)
        if sharded:
            loss, accuracies, processed_graphs = self._reduce_epoch_stats(shard_stats, shard_graphs)
        else:
            accuracies = np.sum(accuracies, axis=0) / processed_graphs
            loss = loss / processed_graphs
        error_ratios = accuracies / chemical_accuracies[self.params["task_ids"]]
        instance_per_sec = processed_graphs / (time.time() - start_time)
        return loss, accuracies, error_ratios, instance_per_sec, steps

    def _readback(self, stats: torch.Tensor):
        """Start the device->host copy of a step's statistics on a SIDE stream, behind an event recorded where they were computed.
        `tensor.cpu()` would put the copy on the training stream, i.e. behind everything queued there: with the next step
        already enqueued the host then waits for that whole step, the launch queue runs dry once per step and the GPU idles
        while the following step is being enqueued (1.1 ms of a 7.2 ms fresh-batch step).  Returns (host tensor, event)."""
        if not stats.is_cuda:
            return (stats, None)
        from .backward import side_stream
        rb = side_stream(stats.device)          # (shared with the weight-gradient products: they are done before the step's end)
        ready = torch.cuda.Event()
        ready.record()
        host = torch.empty(stats.shape, dtype=stats.dtype, device='cpu', pin_memory=True)
        with torch.cuda.stream(rb):
            rb.wait_event(ready)
            host.copy_(stats, non_blocking=True)
            done = torch.cuda.Event()
            done.record()
        stats.record_stream(rb)
        return (host, done)

    def _absorb_step_stats(self, pending, loss, accuracies, epoch_name, step):
        """chem_tensorflow.py:237-246: loss += batch_loss * num_graphs, accuracies likewise, progress line."""
        (host, done), num_graphs, processed = pending
        if done is not None:
            done.synchronize()
        vals = host.numpy()
        loss += float(vals[0]) * num_graphs
        accuracies.append(vals[1:] * num_graphs)
        if not self.quiet:
            print("Running %s, batch %i (has %i graphs). Loss so far: %.4f" % (epoch_name, step, num_graphs, loss / processed), end='\r')
        return loss, processed

    def _reduce_epoch_stats(self, shard_stats, shard_graphs):
        """Epoch loss / per-task MAE under data parallelism: per step the global batch's loss is
        sum_tasks (sum_ranks numerator) / (sum_ranks mask count + 1e-7) (chem_tensorflow.py:161-169 on the union of the
        shards), weighted by the global graph count like the reference's `loss += batch_loss * num_graphs` (:237)."""
        K = len(self.params['task_ids'])
        if not shard_stats:
            return 0.0, np.zeros(K), 0
        packed = torch.cat([torch.stack(shard_stats),
                            torch.tensor(shard_graphs, dtype=torch.float64, device=shard_stats[0].device)[:, None]], dim=1)
        self.dist.all_reduce_sum_(packed)
        packed = packed.cpu().numpy()
        num, den, abs_sum, graphs = packed[:, :K], packed[:, K:2 * K], packed[:, 2 * K:3 * K], packed[:, 3 * K]
        ratios = np.array([1.0 / (self.params['task_sample_ratios'].get(t) or 1.0) for t in self.params['task_ids']])
        step_loss = (num / (den + SMALL_NUMBER) * ratios).sum(axis=1)
        step_acc = abs_sum / (den + SMALL_NUMBER)
        total = graphs.sum()
        return float((step_loss * graphs).sum() / total), (step_acc * graphs[:, None]).sum(axis=0) / total, int(total)

    def train(self):
        """chem_tensorflow.py:255-307."""
        log_to_save = []
        total_time_start = time.time()
        # Move everything alive so far (torch, the datasets, the model) out of the cyclic collector's reach: a full
        # collection over them stalls the launching thread for tens of ms, during which the GPU queue runs dry.
        import gc
        gc.collect()
        gc.freeze()
        if self.args.get('--restore') is not None:
            _, valid_accs, _, _, steps = self.run_epoch("Resumed (validation)", self.valid_data, False)
            best_val_acc = np.sum(valid_accs)
            best_val_acc_epoch = 0
            self.valid_step_id += steps
            print("\r\x1b[KResumed operation, initial cum. val. acc: %.5f" % best_val_acc)
        else:
            (best_val_acc, best_val_acc_epoch) = (float("+inf"), 0)
        for epoch in range(1, self.params['num_epochs'] + 1):
            print("== Epoch %i" % epoch)
            train_loss, train_accs, train_errs, train_speed, train_steps = self.run_epoch(
                "epoch %i (training)" % epoch, self.train_data, True, self.train_step_id)
            self.train_step_id += train_steps
            accs_str = " ".join(["%i:%.5f" % (id, acc) for (id, acc) in zip(self.params['task_ids'], train_accs)])
            errs_str = " ".join(["%i:%.5f" % (id, err) for (id, err) in zip(self.params['task_ids'], train_errs)])
            print("\r\x1b[K Train: loss: %.5f | acc: %s | error_ratio: %s | instances/sec: %.2f" % (
                train_loss, accs_str, errs_str, train_speed))
            valid_loss, valid_accs, valid_errs, valid_speed, valid_steps = self.run_epoch(
                "epoch %i (validation)" % epoch, self.valid_data, False, self.valid_step_id)
            self.valid_step_id += valid_steps
            accs_str = " ".join(["%i:%.5f" % (id, acc) for (id, acc) in zip(self.params['task_ids'], valid_accs)])
            errs_str = " ".join(["%i:%.5f" % (id, err) for (id, err) in zip(self.params['task_ids'], valid_errs)])
            print("\r\x1b[K Valid: loss: %.5f | acc: %s | error_ratio: %s | instances/sec: %.2f" % (
                valid_loss, accs_str, errs_str, valid_speed))
            epoch_time = time.time() - total_time_start
            log_entry = {
                'epoch': epoch,
                'time': epoch_time,
                'train_results': (train_loss, train_accs.tolist(), train_errs.tolist(), train_speed),
                'valid_results': (valid_loss, valid_accs.tolist(), valid_errs.tolist(), valid_speed),
            }
            log_to_save.append(log_entry)
            if not self.quiet:
                with open(self.log_file, 'w') as f:
                    json.dump(log_to_save, f, indent=4)
            val_acc = np.sum(valid_accs)  # type: float
            if val_acc < best_val_acc:
                if not self.quiet and (self.dist is None or self.dist.rank == 0):      # one writer under data parallelism
                    self.save_progress(self.best_model_file, self.train_step_id, self.valid_step_id)
                    print("  (Best epoch so far, cum. val. acc decreased to %.5f from %.5f. Saving to '%s')" % (
                        val_acc, best_val_acc, self.best_model_file))
                best_val_acc = val_acc
                best_val_acc_epoch = epoch
            elif epoch - best_val_acc_epoch >= self.params['patience']:
                print("Stopping training after %i epochs without improvement on validation accuracy." %
                      self.params['patience'])
                break
        return log_to_save

    # ---- checkpoints (chem_tensorflow.py:309-359; same pickle schema, TF variable names) ----------
    def save_progress(self, model_path: str, train_step: int, valid_step: int) -> None:
        weights_to_save = {}
        for name, t in self.named_variables().items():
            assert name not in weights_to_save
            weights_to_save[name] = t.detach().cpu().numpy()
        for name, t in self.optimizer.state_variables(self.trainable_variables).items():
            weights_to_save[name] = t
        data_to_save = {"params": self.params, "weights": weights_to_save,
                        "train_step": train_step, "valid_step": valid_step}
        with open(model_path, 'wb') as out_file:
            pickle.dump(data_to_save, out_file, pickle.HIGHEST_PROTOCOL)

    def restore_progress(self, model_path: str) -> (int, int):
        if not self.quiet:
            print("Restoring weights from file %s." % model_path)
        with open(model_path, 'rb') as in_file:
            data_to_load = pickle.load(in_file)
        # Assert that we got the same model configuration (chem_tensorflow.py:336-340)
        assert len(self.params) == len(data_to_load['params'])
        for (par, par_value) in self.params.items():
            # Fine to have different task_ids, number of epochs:
            if par not in ['task_ids', 'num_epochs']:
                assert par_value == data_to_load['params'][par]
        used_vars = set()
        for name, t in self.named_variables().items():
            used_vars.add(name)
            if name in data_to_load['weights']:
                with torch.no_grad():
                    t.copy_(torch.from_numpy(np.asarray(data_to_load['weights'][name])).to(t.device).reshape(t.shape))
            else:
                print('Freshly initializing %s since no saved value was found.' % name)
        used_vars |= self.optimizer.load_state_variables(self.trainable_variables, data_to_load['weights'])
        for var_name in data_to_load['weights']:
            if var_name not in used_vars:
                print('Saved weights for %s not used by model.' % var_name)
        return data_to_load['train_step'], data_to_load['valid_step']

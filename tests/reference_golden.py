"""Loader for tests/golden/reference_*.npz -- the vectors produced by running the reference's own Python source
(tests/golden/make_reference_golden.py; needs /root/reference, so it is run in the build container and the
fixtures are committed).  Nothing here touches /root/reference."""
import glob
import importlib.util
import json
import os
import pickle

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
GOLDEN = os.path.join(HERE, "golden")
_ALL = sorted(os.path.basename(p)[len("reference_"):-len(".npz")] for p in glob.glob(os.path.join(GOLDEN, "reference_*.npz")))
CASES = [c for c in _ALL if not c.startswith("loop_")]
LOOP_CASES = [c for c in _ALL if c.startswith("loop_")]          # the reference's whole train() loop
SPARSE_CASES = [c for c in CASES if c.startswith("sparse")]
DENSE_CASES = [c for c in CASES if c.startswith("dense")]

_spec = importlib.util.spec_from_file_location("make_reference_golden", os.path.join(GOLDEN, "make_reference_golden.py"))
_gen = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(_gen)
golden_weights = _gen.golden_weights
stats = _gen.stats


class Golden:
    def __init__(self, case):
        self.case = case
        z = np.load(os.path.join(GOLDEN, "reference_%s.npz" % case), allow_pickle=False)
        self.z = z
        self.kind = str(z["kind"])
        self.params = json.loads(str(z["params"]))
        self.num_edge_types = int(z["num_edge_types"])
        self.names = [str(n) for n in z["trainable_names"]]
        self.shapes = [tuple(json.loads(str(s))) for s in z["trainable_shapes"]]
        self.global_names = [str(n) for n in z["global_names"]]
        self.train_molecules = json.loads(str(z["train_molecules"]))
        self.valid_molecules = json.loads(str(z["valid_molecules"]))
        self.weights = {n: golden_weights(n, s, int(z["weight_seed"])) for n, s in zip(self.names, self.shapes)}
        self.num_valid_batches = int(z["num_valid_batches"])
        self.train_losses = z["train_losses"] if "train_losses" in z.files else np.zeros(0)

    def feed(self, prefix):
        """The reference's feed_dict of one batch, keyed by placeholder key (adjacency lists as a list)."""
        pre = prefix + "_feed_"
        out, adj = {}, []
        for k in self.z.files:
            if not k.startswith(pre):
                continue
            key = k[len(pre):]
            if key.startswith("adjacency_") and key != "adjacency_matrix":
                adj.append((int(key[len("adjacency_"):]), self.z[k]))
            else:
                out[key] = self.z[k]
        if adj:
            out["adjacency_lists"] = [a for _, a in sorted(adj, key=lambda t: t[0])]
        return out

    def result(self, prefix, what):
        return self.z["%s_%s" % (prefix, what)]

    # ---- weights in the oracle's layout -------------------------------------------------------------
    def sparse_layers(self):
        D, T = self.params["hidden_size"], self.num_edge_types
        cell = self.params["graph_rnn_cell"].lower()
        scope = {"gru": "gru_cell", "rnn": "basic_rnn_cell", "cudnncompatiblegrucell": "cudnn_compatible_gru_cell"}[cell]
        layers = []
        for l in range(len(self.params["layer_timesteps"])):
            w = self.weights
            base = "graph_model/gnn_layer_%d/" % l
            c = base + "timestep_0/%s/" % scope
            L = {"edge_weights": w[base + "gnn_edge_weights_%d:0" % l].reshape(T, D, D)}          # sparse:88-90
            if base + "gnn_edge_biases_%d:0" % l in w:
                L["edge_biases"] = w[base + "gnn_edge_biases_%d:0" % l]
            if base + "edge_type_attention_weights_%d:0" % l in w:
                L["edge_type_attention_weights"] = w[base + "edge_type_attention_weights_%d:0" % l]
            if cell == "gru":
                L.update(Wg=w[c + "gates/kernel:0"], bg=w[c + "gates/bias:0"], Wc=w[c + "candidate/kernel:0"],
                         bc=w[c + "candidate/bias:0"])
            elif cell == "rnn":
                L.update(W=w[c + "kernel:0"], b=w[c + "bias:0"])
            else:
                L.update(Wg=w[c + "gates/kernel:0"], bg=w[c + "gates/bias:0"],
                         Wcx=w[c + "candidate/input_projection/kernel:0"], bcx=w[c + "candidate/input_projection/bias:0"],
                         Wch=w[c + "candidate/hidden_projection/kernel:0"], bch=w[c + "candidate/hidden_projection/bias:0"])
            layers.append(L)
        return layers

    def dense_weights(self):
        w = self.weights
        c = "graph_model/gru_scope/gru_cell/"
        gru = dict(Wg=w[c + "gates/kernel:0"], bg=w[c + "gates/bias:0"], Wc=w[c + "candidate/kernel:0"], bc=w[c + "candidate/bias:0"])
        return w["graph_model/Variable:0"], w.get("graph_model/Variable_1:0"), gru

    def readout(self, task_id=0):
        w, p = self.weights, "out_layer_task%d/" % task_id
        return (w[p + "regression_gate/MLP_W_layer0:0"], w[p + "regression_gate/MLP_b_layer0:0"],
                w[p + "regression/MLP_W_layer0:0"], w[p + "regression/MLP_b_layer0:0"])

    def write_checkpoint(self, path):
        """A checkpoint in the reference's pickle schema (chem_tensorflow.py:309-323) holding the golden weights: what a
        reference user would hand to ``--restore``."""
        with open(path, "wb") as f:
            pickle.dump({"params": self.params, "weights": dict(self.weights), "train_step": 0, "valid_step": 0}, f)
        return path

    def model_args(self, device, **extra):
        args = {"--quiet": True, "--device": device, "--config": json.dumps(self.params),
                "train_data": self.train_molecules, "valid_data": self.valid_molecules}
        args.update(extra)
        return args


class GoldenLoop:
    """Per-epoch log and best-model checkpoint statistics of the reference's own train() run."""

    def __init__(self, case):
        z = np.load(os.path.join(GOLDEN, "reference_%s.npz" % case), allow_pickle=False)
        self.z, self.case, self.kind = z, case, str(z["kind"])
        self.params = json.loads(str(z["params"]))
        self.train_molecules = json.loads(str(z["train_molecules"]))
        self.valid_molecules = json.loads(str(z["valid_molecules"]))
        self.best_names = [str(n) for n in z["best_names"]]

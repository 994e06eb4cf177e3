"""MI355X-native GGNN propagation engine (drop-in for the hot path of
microsoft/gated-graph-neural-network-samples' chem_tensorflow_sparse.py / chem_tensorflow_dense.py).

The directory name carries a hyphenated repo name, so import it with
    importlib.import_module("gated-graph-neural-network-samples_amd")
or through the alias module `ggnn_amd` at the repo root.  Sub-modules are imported eagerly and
registered under both names.
"""
import importlib as _importlib
import sys as _sys

_ALIAS = "ggnn_amd"
_SUBMODULES = ["_lib", "data", "utils", "ops", "formats", "data_device", "autograd", "backward", "train", "train_native", "chem_model", "sparse_model",
               "dense_model", "parallel", "build"]

_sys.modules.setdefault(_ALIAS, _sys.modules[__name__])
for _m in _SUBMODULES:
    _mod = _importlib.import_module("." + _m, __name__)
    _sys.modules[_ALIAS + "." + _m] = _mod

from .chem_model import ChemModel                                   # noqa: E402
from .sparse_model import SparseGGNNChemModel, GGNNWeights          # noqa: E402
from .dense_model import DenseGGNNChemModel                         # noqa: E402
from .data import MoleculeSet, synthetic_qm9, synthetic_large_graph, pack_batches          # noqa: E402

__all__ = ["ChemModel", "SparseGGNNChemModel", "DenseGGNNChemModel", "GGNNWeights", "MoleculeSet",
           "synthetic_qm9", "synthetic_large_graph", "pack_batches"]

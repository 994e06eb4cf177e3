"""DenseGGNNChemModel placeholder (chem_tensorflow_dense.py) -- built in the dense milestone."""
from __future__ import annotations

from .chem_model import ChemModel


class DenseGGNNChemModel(ChemModel):
    def __init__(self, args):
        raise NotImplementedError("dense path not built yet")

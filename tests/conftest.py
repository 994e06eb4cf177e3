import importlib
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)

PKG = "gated-graph-neural-network-samples_amd"


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def pkg():
    return importlib.import_module(PKG)


@pytest.fixture(scope="session")
def oracle():
    import ggnn_oracle
    return ggnn_oracle


@pytest.fixture(scope="session")
def oracle_torch():
    import ggnn_oracle_torch
    return ggnn_oracle_torch


@pytest.fixture(scope="session")
def cuda():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")


def random_graph_batch(rng, V, M, T, D, sorted_src=False):
    """Random multigraph feed in the reference layout: T adjacency lists [E_t,2], nin [V,T], h [V,D]."""
    types = rng.integers(0, T, M)
    src = rng.integers(0, V, M).astype(np.int32)
    dst = rng.integers(0, V, M).astype(np.int32)
    adj = []
    for t in range(T):
        a = np.stack([src[types == t], dst[types == t]], axis=1).astype(np.int32).reshape(-1, 2)
        if sorted_src and len(a):
            a = a[np.lexsort((a[:, 1], a[:, 0]))]
        adj.append(a)
    nin = np.zeros((V, T), np.float32)
    for t in range(T):
        np.add.at(nin[:, t], adj[t][:, 1], 1.0)
    h = rng.uniform(-1, 1, (V, D)).astype(np.float32)
    return h, adj, nin

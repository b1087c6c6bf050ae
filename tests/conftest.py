import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs an MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    import torch
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def golden():
    def load(name):
        return np.load(os.path.join(GOLDEN, name + ".npz"))
    return load


def latents(N, d, seed, scale=1.0):
    """Same generator as oracle/gen_golden.py (fixture inputs are regenerated, not stored)."""
    return (np.random.RandomState(seed).randn(N, d) * scale).astype(np.float32)


def clustered_latents(N, d, seed):
    r = np.random.RandomState(seed)
    centres = r.randn(6, d).astype(np.float32) * 3
    z = (centres[r.randint(0, 6, size=N)] + 0.05 * r.randn(N, d)).astype(np.float32)
    z[5] = z[3]
    z[N - 1] = z[N // 2]
    z[17] = z[16] = z[15]
    return z


def swiss_roll_latents(N, d, seed):
    """Noisy 2-D swiss roll in d dims (SURVEY 8(d) second distribution; same generator as bench.swiss_roll)."""
    r = np.random.RandomState(seed)
    t = 1.5 * np.pi * (1.0 + 2.0 * r.rand(N))
    h = 21.0 * r.rand(N)
    x = np.zeros((N, d), dtype=np.float64)
    x[:, 0], x[:, 1], x[:, 2] = t * np.cos(t), h, t * np.sin(t)
    x = x / 7.0 + 0.02 * r.randn(N, d)
    return x.astype(np.float32)


def csr_from_golden(g, tag, n, with_data=True):
    from scipy import sparse
    ip, ix = g[f"{tag}/indptr"], g[f"{tag}/indices"]
    data = g[f"{tag}/data"] if with_data and f"{tag}/data" in g.files else np.ones(len(ix), np.float32)
    return sparse.csr_matrix((data, ix, ip), shape=(n, n))

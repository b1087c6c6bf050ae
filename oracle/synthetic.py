"""Seeded synthetic inputs shared by the fixture generators and the tests.  TEST INFRASTRUCTURE ONLY.

Everything here is integer / IEEE-exact arithmetic, so the same arrays come out on any machine: fixtures
store the reference's OUTPUTS and these functions regenerate the inputs.
"""
import hashlib

import numpy as np


def gauss_latents(N: int, d: int, seed: int) -> np.ndarray:
    """z = RandomState(seed).randn(N, d).astype(float32)."""
    return np.random.RandomState(seed).randn(N, d).astype(np.float32)


def latents_as_images(z_flat: np.ndarray, H: int = 4, W: int = 4) -> np.ndarray:
    """(N*H*W, C) rows in (n, h, w) order -> the (N, C, H, W) tensor whose flattening
    (build_codebook.py:35) gives those rows back."""
    n, C = z_flat.shape
    assert n % (H * W) == 0
    return np.ascontiguousarray(np.transpose(z_flat.reshape(n // (H * W), H, W, C), (0, 3, 1, 2)))


def formula_weights(rows: np.ndarray, cols: np.ndarray) -> np.ndarray:
    """Bit-reproducible positive f32 weight of the undirected edge {rows, cols}: an integer hash of
    (min, max) mapped to [0.5, 1.5).  One correctly rounded fp64 division and one f64->f32 rounding."""
    a = np.minimum(rows, cols).astype(np.uint64)
    b = np.maximum(rows, cols).astype(np.uint64)
    h = (a * np.uint64(2654435761) + b * np.uint64(40503) + (a ^ b) * np.uint64(97)) % np.uint64(1000003)
    return (0.5 + h.astype(np.float64) / 1000003.0).astype(np.float32)


def digest(a: np.ndarray) -> np.ndarray:
    """sha256 of the array's bytes as 32 uint8 (structure check without shipping the structure)."""
    return np.frombuffer(hashlib.sha256(np.ascontiguousarray(a).tobytes()).digest(), dtype=np.uint8).copy()


def zero_clusters(sizes, w: float = 1.0):
    """Groups of coincident nodes (explicit zero-weight cliques) chained by weight-w edges between their first
    nodes: every k-means++ draw inside an exhausted neighbourhood has degenerate weights (sum == 0), so a chain
    takes the reference's uniform fallback (kmeans_optimized.py:62-69) several times."""
    from scipy import sparse
    rows, cols, data, reps, off = [], [], [], [], 0
    for s in sizes:
        for i in range(s):
            for j in range(s):
                if i != j:
                    rows.append(off + i), cols.append(off + j), data.append(0.0)
        reps.append(off)
        off += s
    for a in range(len(reps) - 1):
        rows += [reps[a], reps[a + 1]]
        cols += [reps[a + 1], reps[a]]
        data += [w, w]
    return sparse.csr_matrix((np.array(data, np.float32), (rows, cols)), shape=(off, off))


ZERO_CASES = (((4, 3), 5, 0), ((4, 3), 6, 42), ((6,), 4, 1), ((3, 3, 3), 8, 7), ((5, 1, 4), 7, 3))


def seeded_state_dict(template, seed: int):
    """A state dict with the template's keys / shapes / dtypes filled from numpy's RandomState in sorted key order:
    weights uniform in +-1/sqrt(fan_in), norm scales in [0.5, 1.5], running variances in [0.5, 1.5], counters 0 --
    the same values wherever it is regenerated (fixtures store outputs, never weights)."""
    import torch
    r = np.random.RandomState(seed)
    out = {}
    for key in sorted(template.keys()):
        t = template[key]
        shape = tuple(t.shape)
        if not t.dtype.is_floating_point:
            out[key] = torch.zeros(shape, dtype=t.dtype)
            continue
        if key.endswith("running_var") or (key.endswith(".weight") and len(shape) == 1):
            v = r.uniform(0.5, 1.5, size=shape)
        elif key.endswith("running_mean") or key.endswith(".bias"):
            v = r.uniform(-0.2, 0.2, size=shape)
        else:
            fan_in = int(np.prod(shape[1:])) if len(shape) > 1 else max(1, shape[0])
            v = r.uniform(-1.0, 1.0, size=shape) / np.sqrt(max(1, fan_in))
        out[key] = torch.from_numpy(np.asarray(v, dtype=np.float32)).to(t.dtype)
    return out

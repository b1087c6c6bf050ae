"""CPU restatement of src/geo/knn_graph_optimized.py (reference).  TEST INFRASTRUCTURE ONLY.

The neighbour search (sklearn in the reference, knn_graph_optimized.py:40-42) is restated as an exact
fp64-ranked brute force in geo_oracle.c; the CSR assembly / symmetrisation (:54-66) with plain numpy
key arithmetic; connected components (:173-181) with a scan-order flood fill.
"""
import ctypes
from typing import Dict, Optional, Tuple

import numpy as np
from scipy import sparse

from ._clib import lib


def _ptr(a):
    return ctypes.c_void_p(a.ctypes.data)


def knn_search(z: np.ndarray, n_neighbors: int) -> Tuple[np.ndarray, np.ndarray]:
    """(distances fp64 (N,kq), indices int64 (N,kq)) including self, sorted by (distance, index).

    sklearn's algorithm="auto" picks the kd-tree for d <= 15 and brute force above
    (sklearn/neighbors/_base.py:627): the squared distance is formed directly for the former and by
    the |x|^2 - 2xy + |y|^2 expansion for the latter.
    """
    z = np.ascontiguousarray(z, dtype=np.float32)
    N, d = z.shape
    idx = np.empty((N, n_neighbors), np.int64)
    d2 = np.empty((N, n_neighbors), np.float64)
    form = 1 if d > 15 else 0
    rc = lib().oracle_knn(_ptr(z), N, d, n_neighbors, form, 0, N, _ptr(idx), _ptr(d2))
    if rc != 0:
        raise RuntimeError(f"oracle_knn failed: {rc}")
    return np.sqrt(d2), idx


def drop_self(distances: np.ndarray, indices: np.ndarray):
    """knn_graph_optimized.py:45-52: drop column 0 when it is self everywhere, else each row's first minimum."""
    N = indices.shape[0]
    if (indices[:, 0] == np.arange(N)).all():
        return distances[:, 1:], indices[:, 1:]
    pos = np.argmin(distances, axis=1)
    keep = np.ones(distances.shape, dtype=bool)
    keep[np.arange(N), pos] = False
    return distances[keep].reshape(N, -1), indices[keep].reshape(N, -1)


def symmetrise(N: int, indices: np.ndarray, weights: np.ndarray, sym: str) -> sparse.csr_matrix:
    """Directed kNN lists -> canonical symmetric CSR (f32), zero diagonal, no stored zeros (:54-66)."""
    if sym not in ("mutual", "union"):
        raise ValueError(f"Invalid symmetry mode: {sym}")
    rows = np.repeat(np.arange(N, dtype=np.int64), indices.shape[1])
    cols = indices.ravel().astype(np.int64)
    w = weights.ravel().astype(np.float32)
    fwd = rows * N + cols
    bwd = cols * N + rows
    keys = np.union1d(fwd, bwd)
    a = np.zeros(keys.shape[0], np.float32)          # W[r, c], absent = 0
    b = np.zeros(keys.shape[0], np.float32)          # W[c, r]
    a[np.searchsorted(keys, fwd)] = w
    b[np.searchsorted(keys, bwd)] = w
    val = np.maximum(a, b) if sym == "union" else np.minimum(a, b)
    r, c = keys // N, keys % N
    keep = (r != c) & (val != 0)
    return sparse.csr_matrix((val[keep], (r[keep], c[keep])), shape=(N, N), dtype=np.float32)


def build_knn_graph_sklearn(z: np.ndarray, k: int = 10, metric: str = "euclidean", mode: str = "distance",
                            sym: str = "mutual") -> Tuple[sparse.csr_matrix, Dict[str, np.ndarray]]:
    assert z.ndim == 2, "z must be (N,D)"
    if metric not in ("euclidean", "cosine"):
        raise NotImplementedError("oracle restates the euclidean and cosine metrics only")
    N = z.shape[0]
    if N == 0:
        return (sparse.csr_matrix((0, 0), dtype=np.float32),
                {"distances": np.empty((0, 0), np.float32), "indices": np.empty((0, 0), dtype=int)})
    k_eff = max(0, min(k, N - 1))
    if k_eff == 0:
        return (sparse.csr_matrix((N, N), dtype=np.float32),
                {"distances": np.empty((N, 0), np.float32), "indices": np.empty((N, 0), dtype=int)})
    if metric == "cosine":
        # sklearn: cosine_distances = 1 - <x/|x|, y/|y|>, clipped to [0, 2] (sklearn/metrics/pairwise.py); for unit rows
        # that is |x^ - y^|^2 / 2, ranked here in fp64 on the float32-rounded unit rows (zero rows stay zero, as in
        # sklearn's normalize)
        z64 = z.astype(np.float64)
        nrm = np.sqrt((z64 * z64).sum(axis=1, keepdims=True))
        zn = (z64 / np.where(nrm == 0.0, 1.0, nrm)).astype(np.float32)
        dist, idx = knn_search(zn, min(k_eff + 1, N))
        dist = np.clip(dist * dist * 0.5, 0.0, 2.0)
    else:
        dist, idx = knn_search(z, min(k_eff + 1, N))
    dist, idx = drop_self(dist, idx)
    weights = dist if mode == "distance" else np.ones_like(dist)
    W = symmetrise(N, idx, weights, sym)
    return W, {"distances": dist.astype(np.float32), "indices": idx}


def build_knn_graph_auto(z, k=10, metric="euclidean", mode="distance", sym="mutual",
                         force_method: Optional[str] = None, size_threshold: int = 50000):
    if force_method == "faiss":
        raise RuntimeError("force_method='faiss' but FAISS not available")
    return build_knn_graph_sklearn(z, k=k, metric=metric, mode=mode, sym=sym)


def build_knn_graph(z, k=10, metric="euclidean", mode="distance", sym="mutual"):
    return build_knn_graph_auto(z, k=k, metric=metric, mode=mode, sym=sym)


def build_knn_graph_faiss_semantics(z, k=10, metric="euclidean", mode="distance", sym="mutual"):
    """knn_graph_optimized.py:70-126 with faiss.IndexFlatL2 / IndexFlatIP restated from their published definition
    (faiss 1.x, not installed here: PARITY UNPINNED): exhaustive search, `search` returns SQUARED L2 distances resp. inner
    products as float32, nearest / largest first.  Small inputs only (dense N x N matrix)."""
    z = np.asarray(z)
    N = z.shape[0]
    if metric == "euclidean":
        x = np.ascontiguousarray(z.astype(np.float32)).astype(np.float64)
        score = ((x[:, None, :] - x[None, :, :]) ** 2).sum(axis=2)
        order = np.argsort(score, axis=1, kind="stable")[:, :min(k + 1, N)]
        dist = np.take_along_axis(score, order, axis=1).astype(np.float32)
    elif metric == "cosine":
        x = np.ascontiguousarray((z / (np.linalg.norm(z, axis=1, keepdims=True) + 1e-8)).astype(np.float32)).astype(np.float64)
        sim = x @ x.T
        order = np.argsort(-sim, axis=1, kind="stable")[:, :min(k + 1, N)]
        dist = (np.float32(1.0) - np.take_along_axis(sim, order, axis=1).astype(np.float32)).astype(np.float32)
    else:
        raise ValueError(f"FAISS metric '{metric}' not supported. Use 'euclidean' or 'cosine'.")
    idx = order.astype(np.int64)
    if idx.shape[1] > 1 and (idx[:, 0] == np.arange(N)).all():
        dist, idx = dist[:, 1:], idx[:, 1:]
    kk = idx.shape[1]
    data = dist.ravel() if mode == "distance" else np.ones(N * kk, dtype=np.float32)
    W = sparse.csr_matrix((data, (np.repeat(np.arange(N), kk), idx.ravel())), shape=(N, N))
    if sym == "mutual":
        W = W.minimum(W.T)
    elif sym == "union":
        W = W.maximum(W.T)
    else:
        raise ValueError(f"Invalid symmetry mode: {sym}")
    W.setdiag(0.0)
    W.eliminate_zeros()
    return W.tocsr(), {"distances": dist, "indices": idx}


def connected_components(W: sparse.spmatrix) -> Tuple[int, np.ndarray]:
    W = sparse.csr_matrix(W)
    WT = W.T.tocsr()
    n = W.shape[0]
    labels = np.empty(n, np.int32)
    ip, ix = W.indptr.astype(np.int32), W.indices.astype(np.int32)
    ipT, ixT = WT.indptr.astype(np.int32), WT.indices.astype(np.int32)
    ncomp = lib().oracle_cc(n, _ptr(ip), _ptr(ix), _ptr(ipT), _ptr(ixT), _ptr(labels))
    return int(ncomp), labels


def largest_connected_component(W: sparse.spmatrix) -> np.ndarray:
    """knn_graph_optimized.py:173-181."""
    ncomp, labels = connected_components(W)
    if ncomp <= 1:
        return np.ones(W.shape[0], dtype=bool)
    return labels == np.argmax(np.bincount(labels))


def analyze_graph_connectivity(W: sparse.spmatrix) -> Dict:
    """knn_graph_optimized.py:184-219 (stats only, no prints)."""
    N = W.shape[0]
    ncomp, labels = connected_components(W)
    if ncomp > 1:
        largest = int(np.bincount(labels).max())
        ratio = largest / N
    else:
        largest, ratio = N, 1.0
    deg = np.asarray(W.sum(axis=1)).ravel()
    return {"n_nodes": N, "n_edges": W.nnz, "n_components": ncomp, "largest_component_size": largest,
            "connectivity_ratio": ratio, "avg_degree": deg.mean(), "min_degree": deg.min(),
            "max_degree": deg.max()}

"""CPU restatement of src/geo/geo_shortest_paths.py (reference).  TEST INFRASTRUCTURE ONLY."""
import ctypes

import numpy as np
from scipy import sparse

from ._clib import lib


def _ptr(a):
    return ctypes.c_void_p(a.ctypes.data)


def ensure_valid_graph(W):
    """geo_shortest_paths.py:13-21."""
    if not sparse.isspmatrix(W):
        raise TypeError("W must be a scipy sparse matrix")
    if W.shape[0] != W.shape[1]:
        raise ValueError("W must be square")
    if W.nnz > 0 and (W.data < 0).any():
        raise ValueError("Negative weights")
    return W.tocsr()


def dijkstra_multi_source(W, sources, directed=False, unweighted=False, return_predecessors=False,
                          dtype=np.float32):
    """geo_shortest_paths.py:24-50: fp64 path sums, cast to `dtype` at the end."""
    if len(sources) == 0:
        raise ValueError("sources must be a non-empty sequence of node indices")
    W = ensure_valid_graph(W)
    src = np.ascontiguousarray(np.asarray(sources, dtype=int), dtype=np.int64)
    n = W.shape[0]
    WT = W.T.tocsr()
    ip, ix = W.indptr.astype(np.int32), W.indices.astype(np.int32)
    ipT, ixT = WT.indptr.astype(np.int32), WT.indices.astype(np.int32)
    if unweighted:
        w, wT = np.ones(W.nnz, np.float64), np.ones(WT.nnz, np.float64)
    else:
        w, wT = W.data.astype(np.float64), WT.data.astype(np.float64)
    D = np.empty((len(src), n), np.float64)
    P = np.empty((len(src), n), np.int32) if return_predecessors else None
    rc = lib().oracle_sssp(n, _ptr(ip), _ptr(ix), _ptr(w), _ptr(ipT), _ptr(ixT), _ptr(wT),
                           1 if directed else 0, len(src), _ptr(src), _ptr(D),
                           _ptr(P) if P is not None else None)
    if rc != 0:
        raise IndexError(f"oracle_sssp failed: {rc}")
    D = D.astype(dtype, copy=False)
    return (D, P) if return_predecessors else D


def dijkstra_single_source(W, source, directed=False, unweighted=False, return_predecessors=False,
                           dtype=np.float32):
    """geo_shortest_paths.py:53-63."""
    out = dijkstra_multi_source(W, [int(source)], directed=directed, unweighted=unweighted,
                                return_predecessors=return_predecessors, dtype=dtype)
    if return_predecessors:
        return out[0][0], out[1][0]
    return out[0]


def distances_between(W, sources, targets, directed=False, unweighted=False, dtype=np.float32):
    """geo_shortest_paths.py:66-76."""
    if len(sources) == 0 or len(targets) == 0:
        raise ValueError("sources and targets must be non-empty.")
    targets = np.asarray(targets, dtype=int)
    D = dijkstra_multi_source(W, np.asarray(sources, dtype=int), directed=directed, unweighted=unweighted,
                              dtype=dtype)
    return D[:, targets]

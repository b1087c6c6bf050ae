"""Geodesic shortest paths on the MI355X -- same API as the reference's
src/geo/geo_shortest_paths.py (ensure_valid_graph :13, dijkstra_multi_source :24,
dijkstra_single_source :53, distances_between :66), with scipy's Dijkstra replaced by the fp64
label-correcting kernels of csrc/sssp.hip (geo_sssp_multi).  Host objects in, host objects out."""
from typing import Optional, Tuple

import numpy as np
import torch
from scipy import sparse

from .. import _lib
from .._device import DeviceCSR, device, ptr, stream_ptr, workspace


def ensure_valid_graph(W: sparse.spmatrix) -> sparse.spmatrix:
    """Same checks and exception classes as geo_shortest_paths.py:13-21."""
    if not sparse.isspmatrix(W):
        raise TypeError("W must be a scipy sparse matrix")
    if W.shape[0] != W.shape[1]:
        raise ValueError("W must be square")
    if W.nnz > 0 and (W.data < 0).any():
        raise ValueError("Negative weights")
    return W.tocsr()


def _pull_structure(W: sparse.csr_matrix, directed: bool) -> sparse.csr_matrix:
    """CSR whose row v lists the edges INTO v.  Undirected solves use every stored entry in both
    directions (scipy relaxes along csr and csr^T); duplicates are kept as parallel edges, the
    kernel takes the minimum over them."""
    WT = W.T.tocsr()
    if directed:
        return WT
    W = W.copy() if not W.has_sorted_indices else W
    W.sort_indices()
    WT.sort_indices()
    if (np.array_equal(W.indptr, WT.indptr) and np.array_equal(W.indices, WT.indices)
            and np.array_equal(W.data, WT.data)):
        return W
    n = W.shape[0]
    rows = np.concatenate([np.repeat(np.arange(n), np.diff(W.indptr)), np.repeat(np.arange(n), np.diff(WT.indptr))])
    cols = np.concatenate([W.indices, WT.indices])
    data = np.concatenate([W.data, WT.data])
    order = np.argsort(rows, kind="stable")
    indptr = np.zeros(n + 1, dtype=np.int64)
    np.cumsum(np.bincount(rows, minlength=n), out=indptr[1:])
    G = sparse.csr_matrix((n, n), dtype=W.dtype)
    G.indptr, G.indices, G.data = indptr.astype(np.int32), cols[order].astype(np.int32), data[order]
    return G


def sssp_multi_device(G: DeviceCSR, sources: torch.Tensor, *, unweighted: bool = False, want_D: bool = True,
                      want_P: bool = False, want_min: bool = False, out: Optional[torch.Tensor] = None):
    """Device-resident multi-source solve.  `sources` int32 on G's device.  Returns
    (D f32 [S,n] | None, P i32 [S,n] | None, dmin f32 [n] | None, argmin i32 [n] | None, sweeps).
    `out`: a contiguous f32 [S,n] block (e.g. rows of a resident all-pairs matrix) to receive D."""
    lib = _lib.load()
    dev = G.indptr.device
    S, n = int(sources.numel()), G.n
    if out is not None:
        assert out.dtype == torch.float32 and out.is_contiguous() and tuple(out.shape) == (S, n) and out.device == dev
    D = out if out is not None else (torch.empty((S, n), dtype=torch.float32, device=dev) if want_D else None)
    P = torch.empty((S, n), dtype=torch.int32, device=dev) if want_P else None
    dmin = torch.empty(n, dtype=torch.float32, device=dev) if want_min else None
    amin = torch.empty(n, dtype=torch.int32, device=dev) if want_min else None
    ws = workspace(lib.geo_sssp_workspace_bytes(n, G.nnz, S), dev)
    sweeps = np.zeros(1, dtype=np.int32)
    weights = None if unweighted else G.data
    with torch.cuda.device(dev):
        _lib.check(lib.geo_sssp_multi(ptr(G.indptr), ptr(G.indices), ptr(weights), n, G.nnz, ptr(sources), S,
                                      ptr(D), ptr(P), ptr(dmin), ptr(amin), ptr(ws), ws.numel(),
                                      sweeps.ctypes.data, stream_ptr()), "geo_sssp_multi")
    return D, P, dmin, amin, int(sweeps[0])


def nearest_source_device(G: DeviceCSR, sources: torch.Tensor, *, unweighted: bool = False, info: Optional[dict] = None):
    """(dmin f32 [n], argmin i32 [n], sweeps): D.min(axis=0) / D.argmin(axis=0) of dijkstra_multi_source's FLOAT32 matrix
    (geo_shortest_paths.py:50, kmeans_optimized.py:100: the lowest row whose distance rounds to the column minimum) from ONE
    label-carrying solve (csrc/sssp.hip, geo_sssp_nearest_source) -- K times less work than the matrix.  G must be symmetric.
    When the call declines (weights outside the exact units, too many float32 collisions) the K-source solve answers instead.
    `info`, if given, receives declined / reason / suspects / sweeps of the one-solve attempt."""
    lib = _lib.load()
    dev = G.indptr.device
    S, n = int(sources.numel()), G.n
    dmin = torch.empty(n, dtype=torch.float32, device=dev)
    amin = torch.empty(n, dtype=torch.int32, device=dev)
    ws = workspace(lib.geo_sssp_nearest_workspace_bytes(n, G.nnz), dev)
    status = np.zeros(4, dtype=np.int32)
    weights = None if unweighted else G.data
    with torch.cuda.device(dev):
        _lib.check(lib.geo_sssp_nearest_source(ptr(G.indptr), ptr(G.indices), ptr(weights), n, G.nnz, ptr(sources), S,
                                               ptr(dmin), ptr(amin), ptr(ws), ws.numel(), status.ctypes.data, stream_ptr()),
                   "geo_sssp_nearest_source")
    if info is not None:
        info.update(declined=bool(status[0]), reason=int(status[3]), suspects=int(status[2]), sweeps=int(status[1]))
    if status[0] == 0:
        return dmin, amin, int(status[1])
    _, _, dmin, amin, sweeps = sssp_multi_device(G, sources, unweighted=unweighted, want_D=False, want_min=True)
    return dmin, amin, sweeps


def _normalise_sources(sources, n: int) -> np.ndarray:
    src = np.asarray(sources, dtype=int).copy()
    src[src < 0] += n
    if src.size and (src.min() < 0 or src.max() >= n):
        raise ValueError(f"indices out of range 0...{n}")
    return src


def dijkstra_multi_source(W: sparse.spmatrix, sources, directed: bool = False, unweighted: bool = False,
                          return_predecessors: bool = False, dtype=np.float32) -> Tuple:
    """Multi-source geodesic distances: D[i, v] = dist(sources[i], v) (geo_shortest_paths.py:24-50)."""
    if len(sources) == 0:
        raise ValueError("sources must be a non-empty sequence of node indices")
    W = ensure_valid_graph(W)
    n = W.shape[0]
    src = _normalise_sources(sources, n)
    if n == 0:
        raise ValueError("graph has no nodes")
    dev = device()
    G = DeviceCSR.from_scipy(_pull_structure(W, directed), dev, with_data=not unweighted)
    src_t = torch.from_numpy(src.astype(np.int32)).to(dev)
    D, P, _, _, _ = sssp_multi_device(G, src_t, unweighted=unweighted, want_D=True, want_P=return_predecessors)
    D_host = D.cpu().numpy().astype(dtype, copy=False)
    if return_predecessors:
        return D_host, P.cpu().numpy().astype(np.int32, copy=False)
    return D_host


def dijkstra_single_source(W: sparse.spmatrix, source: int, directed: bool = False, unweighted: bool = False,
                           return_predecessors: bool = False, dtype=np.float32) -> Tuple:
    """Single-source wrapper returning 1-D arrays (geo_shortest_paths.py:53-63)."""
    out = dijkstra_multi_source(W, [int(source)], directed=directed, unweighted=unweighted,
                                return_predecessors=return_predecessors, dtype=dtype)
    if return_predecessors:
        return out[0][0], out[1][0]
    return out[0]


def distances_between(W: sparse.spmatrix, sources, targets, directed: bool = False, unweighted: bool = False,
                      dtype=np.float32) -> np.ndarray:
    """Compact (S x T) distance matrix (geo_shortest_paths.py:66-76)."""
    if len(sources) == 0 or len(targets) == 0:
        raise ValueError("sources and targets must be non-empty.")
    sources = np.asarray(sources, dtype=int)
    targets = np.asarray(targets, dtype=int)
    D = dijkstra_multi_source(W, sources, directed=directed, unweighted=unweighted,
                              return_predecessors=False, dtype=dtype)
    return D[:, targets]


def all_pairs_geodesic_device(G: DeviceCSR, block: int = 512, max_bytes: int = 200 << 30) -> torch.Tensor:
    """All-pairs geodesic distances as a resident f32 [n, n] matrix (row s = scipy Dijkstra from s rounded to f32, the
    rows geo_sssp_multi returns), filled `block` sources at a time.  Extension: the reference never
    forms the full matrix; 288 GB of HBM hold it up to n = 230 000 (14.4 GB at the 60 000-latent configuration)."""
    n = G.n
    if n * n * 4 > max_bytes:
        raise ValueError(f"all-pairs matrix of {n} nodes needs {n * n * 4 / 2**30:.1f} GiB (limit {max_bytes / 2**30:.0f} GiB)")
    dev = G.indptr.device
    D = torch.empty((n, n), dtype=torch.float32, device=dev)
    for s0 in range(0, n, block):
        s1 = min(n, s0 + block)
        sssp_multi_device(G, torch.arange(s0, s1, dtype=torch.int32, device=dev), out=D[s0:s1])
    return D


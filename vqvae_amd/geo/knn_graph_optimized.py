"""k-NN graph construction on the MI355X -- same API as the reference's
src/geo/knn_graph_optimized.py (build_knn_graph_sklearn :25, build_knn_graph_faiss :70,
build_knn_graph_auto :129, largest_connected_component :173, analyze_graph_connectivity :184,
build_knn_graph :223).

The neighbour search (sklearn / FAISS in the reference) is the exact fp64-ranked brute force of
csrc/knn.hip; the CSR assembly, union / mutual symmetrisation, diagonal and zero removal run in
csrc/graph.hip, as do connected components.  Results are returned as the same host objects
(scipy.sparse.csr_matrix float32, numpy arrays).
"""
from typing import Dict, Optional, Tuple

import numpy as np
import torch
from scipy import sparse

from .. import _lib
from .._device import DeviceCSR, device, ptr, stream_ptr, workspace

_SYM_MODE = {"union": 0, "mutual": 1}
MAX_NEIGHBORS = 256         # n_neighbors (k + 1) supported by the wave-resident top-k lists (geo_hip.h)


# --------------------------------------------------------------------------------- device level
def knn_search_device(z: torch.Tensor, n_neighbors: int, row0: int = 0, row1: Optional[int] = None):
    """(idx int32 [rows, n_neighbors], d2 float64 [rows, n_neighbors]) sorted by (distance, index),
    self included, for query rows [row0, row1) of the resident latents z (f32 [N, d])."""
    lib = _lib.load()
    N, d = z.shape
    row1 = N if row1 is None else row1
    if n_neighbors > MAX_NEIGHBORS:
        raise ValueError(f"k + 1 = {n_neighbors} neighbours exceed the supported maximum of {MAX_NEIGHBORS}")
    if d > 128:
        raise ValueError(f"latent dimension {d} exceeds the supported maximum of 128")
    idx = torch.empty((row1 - row0, n_neighbors), dtype=torch.int32, device=z.device)
    d2 = torch.empty((row1 - row0, n_neighbors), dtype=torch.float64, device=z.device)
    ws = workspace(lib.geo_knn_workspace_bytes(N, d), z.device)
    form = 1 if d > 15 else 0           # sklearn "auto": brute-force expansion above 15 dims, kd-tree below
    with torch.cuda.device(z.device):
        _lib.check(lib.geo_knn_topk(ptr(z), N, d, n_neighbors, form, row0, row1, ptr(idx), ptr(d2), ptr(ws),
                                    ws.numel(), stream_ptr()), "geo_knn_topk")
    return idx, d2


def symmetrize_device(nbr_idx: torch.Tensor, nbr_w: Optional[torch.Tensor], sym: str) -> DeviceCSR:
    """Directed neighbour lists (int32 [N,k], optional f32 weights) -> canonical symmetric CSR."""
    lib = _lib.load()
    n, k = nbr_idx.shape
    dev = nbr_idx.device
    ws = workspace(lib.geo_symmetrize_workspace_bytes(n, k), dev)
    indptr = torch.empty(n + 1, dtype=torch.int32, device=dev)
    nnz = np.zeros(1, dtype=np.int64)
    mode = _SYM_MODE[sym]
    with torch.cuda.device(dev):
        _lib.check(lib.geo_symmetrize_count(ptr(nbr_idx), ptr(nbr_w), n, k, mode, ptr(indptr), nnz.ctypes.data,
                                            ptr(ws), ws.numel(), stream_ptr()), "geo_symmetrize_count")
        indices = torch.empty(int(nnz[0]), dtype=torch.int32, device=dev)
        data = torch.empty(int(nnz[0]), dtype=torch.float32, device=dev)
        _lib.check(lib.geo_symmetrize_fill(ptr(nbr_idx), ptr(nbr_w), n, k, mode, ptr(indptr), ptr(indices), ptr(data),
                                           ptr(ws), ws.numel(), stream_ptr()), "geo_symmetrize_fill")
    return DeviceCSR(n, indptr, indices, data)


def _drop_self(dist: torch.Tensor, idx: torch.Tensor):
    """knn_graph_optimized.py:45-52 on device tensors."""
    N = idx.shape[0]
    me = torch.arange(N, device=idx.device, dtype=idx.dtype)
    if bool((idx[:, 0] == me).all()):
        return dist[:, 1:].contiguous(), idx[:, 1:].contiguous()
    # duplicates put another point in front of self somewhere: drop each row's first minimum (rare path,
    # done with numpy's first-index argmin on the host to keep its tie rule)
    dist_h, idx_h = dist.cpu().numpy(), idx.cpu().numpy()
    keep = np.ones(idx_h.shape, dtype=bool)
    keep[np.arange(N), np.argmin(dist_h, axis=1)] = False
    dist_k = torch.from_numpy(dist_h[keep].reshape(N, -1)).to(dist.device)
    idx_k = torch.from_numpy(idx_h[keep].reshape(N, -1)).to(idx.device)
    return dist_k.contiguous(), idx_k.contiguous()


_TORCH_METRICS = {"manhattan": 1.0, "cityblock": 1.0, "l1": 1.0, "chebyshev": float("inf"), "l2": 2.0, "minkowski": 2.0}


def _generic_metric_search(z: torch.Tensor, n_neighbors: int, metric: str):
    """Metrics without a HIP kernel (sklearn accepts them, knn_graph_optimized.py:40): blockwise torch.cdist + top-k on the
    GPU, ties by index.  Returns (idx int32, distance f64) including self."""
    if metric not in _TORCH_METRICS:
        raise ValueError(f"Metric '{metric}' not valid. Use 'euclidean', 'cosine' or one of {sorted(_TORCH_METRICS)}")
    N = z.shape[0]
    zd = z.to(torch.float64)
    idx_out = torch.empty((N, n_neighbors), dtype=torch.int32, device=z.device)
    d_out = torch.empty((N, n_neighbors), dtype=torch.float64, device=z.device)
    step = max(1, min(N, (1 << 27) // max(1, N)))                      # <= 1 GiB of fp64 distances per block
    for r0 in range(0, N, step):
        D = torch.cdist(zd[r0:r0 + step], zd, p=_TORCH_METRICS[metric])
        # stable ranking by (distance, index): sort is stable along the index axis
        vals, order = torch.sort(D, dim=1, stable=True)
        idx_out[r0:r0 + step] = order[:, :n_neighbors].to(torch.int32)
        d_out[r0:r0 + step] = vals[:, :n_neighbors]
    return idx_out, d_out


def _cosine_search_with_zero_rows(unit: torch.Tensor, n_neighbors: int):
    """Cosine search when some latent is exactly zero.  sklearn (cosine_distances behind knn_graph_optimized.py:40) leaves
    a zero row un-normalised, so its similarity to EVERY row is 0 and its distance 1.0 -- not the |x^ - y^|^2 / 2 = 0.5
    the unit-row Euclidean search would give.  Rare, so no kernel: 1 - <x^, y^> blockwise in fp64 on the GPU, clipped to
    [0, 2], self-distance 0, ranked by (distance, index).  Returns (idx int32, distance f64) including self."""
    N = unit.shape[0]
    idx_out = torch.empty((N, n_neighbors), dtype=torch.int32, device=unit.device)
    d_out = torch.empty((N, n_neighbors), dtype=torch.float64, device=unit.device)
    step = max(1, min(N, (1 << 27) // max(1, N)))
    for r0 in range(0, N, step):
        D = (1.0 - unit[r0:r0 + step] @ unit.T).clamp_(0.0, 2.0)
        rows = torch.arange(r0, min(r0 + step, N), device=unit.device)
        D[rows - r0, rows] = 0.0
        vals, order = torch.sort(D, dim=1, stable=True)
        idx_out[r0:r0 + step] = order[:, :n_neighbors].to(torch.int32)
        d_out[r0:r0 + step] = vals[:, :n_neighbors]
    return idx_out, d_out


def knn_graph_device(z: torch.Tensor, k: int, mode: str = "distance", sym: str = "mutual", group=None,
                     need_dist: bool = True, metric: str = "euclidean"):
    """Resident latents -> (DeviceCSR, distances f64 [N,k'] | None, indices int32 [N,k']); k' = min(k, N-1) >= 1.
    With an initialised process group the query rows are sharded over the ranks (parallel.sharded_knn); a
    connectivity graph with need_dist=False then never gathers the fp64 distances (returned as None)."""
    from ..parallel import sharded_knn
    if sym not in _SYM_MODE:
        raise ValueError(f"Invalid symmetry mode: {sym}")
    N = z.shape[0]
    k_eff = max(0, min(k, N - 1))
    if metric == "cosine":
        # cosine distance = 1 - <x^, y^> = |x^ - y^|^2 / 2 on the unit rows: the exact fp64-ranked Euclidean search of
        # csrc/knn.hip on the normalised latents ranks exactly by it (sklearn: cosine_distances, clipped to [0, 2])
        z64 = z.to(torch.float64)
        nrm = torch.linalg.vector_norm(z64, dim=1, keepdim=True)
        unit64 = z64 / torch.where(nrm == 0, torch.ones_like(nrm), nrm)
        if bool((nrm == 0).any()):                     # a zero latent is at cosine distance 1 from everything
            idx, dist = _cosine_search_with_zero_rows(unit64, min(k_eff + 1, N))
            dist, idx = _drop_self(dist, idx)
            weights = dist.to(torch.float32).contiguous() if mode == "distance" else None
            return symmetrize_device(idx, weights, sym), dist, idx
        z = unit64.to(torch.float32).contiguous()
    elif metric != "euclidean":
        idx, dist = _generic_metric_search(z, min(k_eff + 1, N), metric)
        dist, idx = _drop_self(dist, idx)
        weights = dist.to(torch.float32).contiguous() if mode == "distance" else None
        return symmetrize_device(idx, weights, sym), dist, idx
    idx, d2 = sharded_knn(z, min(k_eff + 1, N), knn_search_device, group,
                          gather_d2=need_dist or mode == "distance" or metric == "cosine")
    if callable(d2):                  # multi-rank connectivity graph: distances only if a self match is displaced
        me = torch.arange(N, device=idx.device, dtype=idx.dtype)
        if bool((idx[:, 0] == me).all()):          # same gathered idx on every rank -> same decision on every rank
            return symmetrize_device(idx[:, 1:].contiguous(), None, sym), None, idx[:, 1:].contiguous()
        d2 = d2()
    dist = torch.clamp(d2 * 0.5, 0.0, 2.0) if metric == "cosine" else torch.sqrt(d2)
    dist, idx = _drop_self(dist, idx)
    weights = dist.to(torch.float32).contiguous() if mode == "distance" else None
    return symmetrize_device(idx, weights, sym), dist, idx


def connected_components_device(G: DeviceCSR):
    """(n_components, labels int32 on device) for a structurally symmetric resident graph."""
    lib = _lib.load()
    dev = G.indptr.device
    labels = torch.empty(G.n, dtype=torch.int32, device=dev)
    ncomp = np.zeros(1, dtype=np.int32)
    ws = workspace(lib.geo_cc_workspace_bytes(G.n), dev)
    with torch.cuda.device(dev):
        _lib.check(lib.geo_connected_components(ptr(G.indptr), ptr(G.indices), G.n, ptr(labels), ncomp.ctypes.data,
                                                ptr(ws), ws.numel(), stream_ptr()), "geo_connected_components")
    return int(ncomp[0]), labels


def lcc_mask_device(G: DeviceCSR) -> torch.Tensor:
    """Boolean mask (device) of the largest component; first label on ties (np.argmax of bincount)."""
    ncomp, labels = connected_components_device(G)
    if ncomp <= 1:
        return torch.ones(G.n, dtype=torch.bool, device=G.indptr.device)
    counts = torch.bincount(labels.long(), minlength=ncomp).cpu().numpy()
    return labels == int(np.argmax(counts))


def compact_device(G: DeviceCSR, keep: Optional[torch.Tensor], drop_zero: bool) -> Tuple[DeviceCSR, torch.Tensor]:
    """Sub-graph on the kept nodes (W[mask][:, mask]) without zero-weight entries; also the old->new index map."""
    lib = _lib.load()
    dev = G.indptr.device
    ws = workspace(lib.geo_csr_compact_workspace_bytes(G.n), dev)
    keep_u8 = None if keep is None else keep.to(torch.uint8).contiguous()
    new_index = torch.empty(G.n, dtype=torch.int32, device=dev)
    indptr_new = torch.empty(G.n + 1, dtype=torch.int32, device=dev)
    n_new, nnz_new = np.zeros(1, dtype=np.int32), np.zeros(1, dtype=np.int64)
    with torch.cuda.device(dev):
        _lib.check(lib.geo_csr_compact_count(ptr(G.indptr), ptr(G.indices), ptr(G.data), G.n, ptr(keep_u8),
                                             1 if drop_zero else 0, ptr(new_index), ptr(indptr_new),
                                             n_new.ctypes.data, nnz_new.ctypes.data, ptr(ws), ws.numel(),
                                             stream_ptr()), "geo_csr_compact_count")
        indices = torch.empty(int(nnz_new[0]), dtype=torch.int32, device=dev)
        data = torch.empty(int(nnz_new[0]), dtype=torch.float32, device=dev)
        if int(n_new[0]) > 0:
            _lib.check(lib.geo_csr_compact_fill(ptr(G.indptr), ptr(G.indices), ptr(G.data), G.n, ptr(keep_u8),
                                                1 if drop_zero else 0, ptr(new_index), ptr(indptr_new), ptr(indices),
                                                ptr(data), stream_ptr()), "geo_csr_compact_fill")
    n = int(n_new[0])
    return DeviceCSR(n, indptr_new[: n + 1].contiguous(), indices, data), new_index


def upper_edges_device(G: DeviceCSR):
    """(src, dst int32 [E], entry_edge int32 [nnz]): stored entries with row < col in row-major order
    (build_codebook.py:43-45) and, for every entry, the index of its undirected edge."""
    lib = _lib.load()
    dev = G.indptr.device
    ws = workspace(lib.geo_cc_workspace_bytes(G.n), dev)
    upper_ptr = torch.empty(G.n + 1, dtype=torch.int32, device=dev)
    n_edges = np.zeros(1, dtype=np.int64)
    with torch.cuda.device(dev):
        _lib.check(lib.geo_upper_edges_count(ptr(G.indptr), ptr(G.indices), G.n, ptr(upper_ptr), n_edges.ctypes.data,
                                             ptr(ws), ws.numel(), stream_ptr()), "geo_upper_edges_count")
        E = int(n_edges[0])
        src = torch.empty(E, dtype=torch.int32, device=dev)
        dst = torch.empty(E, dtype=torch.int32, device=dev)
        entry_edge = torch.empty(G.nnz, dtype=torch.int32, device=dev)
        _lib.check(lib.geo_upper_edges_fill(ptr(G.indptr), ptr(G.indices), G.n, ptr(upper_ptr), ptr(src), ptr(dst),
                                            ptr(entry_edge), stream_ptr()), "geo_upper_edges_fill")
    return src, dst, entry_edge


def reweight_device(G: DeviceCSR, entry_edge: torch.Tensor, lengths: torch.Tensor) -> DeviceCSR:
    """W_geo = U + U^T (build_codebook.py:53-54): every stored entry takes its edge's length."""
    lib = _lib.load()
    data = torch.empty(G.nnz, dtype=torch.float32, device=G.indptr.device)
    with torch.cuda.device(G.indptr.device):
        _lib.check(lib.geo_gather_edge_weights(ptr(lengths), ptr(entry_edge), G.nnz, ptr(data), stream_ptr()),
                   "geo_gather_edge_weights")
    return DeviceCSR(G.n, G.indptr, G.indices, data)


# --------------------------------------------------------------------------------- reference API
def _empty_result(N: int):
    W = sparse.csr_matrix((N, N), dtype=np.float32)
    return W, {"distances": np.empty((N, 0), np.float32), "indices": np.empty((N, 0), dtype=int)}


def build_knn_graph_sklearn(z: np.ndarray, k: int = 10, metric: str = "euclidean", mode: str = "distance",
                            sym: str = "mutual") -> Tuple[sparse.csr_matrix, Dict[str, np.ndarray]]:
    """Exact k-NN graph (the reference's scikit-learn builder, knn_graph_optimized.py:25-67), computed on the GPU."""
    assert z.ndim == 2, "z must be (N,D)"
    N = z.shape[0]
    if N == 0:
        return (sparse.csr_matrix((0, 0), dtype=np.float32),
                {"distances": np.empty((0, 0), np.float32), "indices": np.empty((0, 0), dtype=int)})
    if max(0, min(k, N - 1)) == 0:
        return _empty_result(N)
    if sym not in _SYM_MODE:
        raise ValueError(f"Invalid symmetry mode: {sym}")
    dev = device()
    z_dev = torch.from_numpy(np.ascontiguousarray(z, dtype=np.float32)).to(dev)
    G, dist, idx = knn_graph_device(z_dev, k, mode=mode, sym=sym, metric=metric)
    info = {"distances": dist.to(torch.float32).cpu().numpy(), "indices": idx.cpu().numpy().astype(np.int64)}
    return G.to_scipy(), info


def _faiss_style_graph(idx: torch.Tensor, dist32: torch.Tensor, mode: str, sym: str):
    """knn_graph_optimized.py:102-124: drop column 0 only if it is the row itself in EVERY row, CSR from the remaining
    columns (squared distances or ones), mutual / union, zero diagonal, explicit zeros removed."""
    N = idx.shape[0]
    me = torch.arange(N, device=idx.device, dtype=idx.dtype)
    if idx.shape[1] > 1 and bool((idx[:, 0] == me).all()):
        idx, dist32 = idx[:, 1:].contiguous(), dist32[:, 1:].contiguous()
    weights = dist32.contiguous() if mode == "distance" else None
    return symmetrize_device(idx.contiguous(), weights, sym), dist32, idx


def build_knn_graph_faiss(z: np.ndarray, k: int = 10, metric: str = "euclidean", mode: str = "distance",
                          sym: str = "mutual") -> Tuple[sparse.csr_matrix, Dict[str, np.ndarray]]:
    """The reference's FAISS branch (knn_graph_optimized.py:70-126) WITHOUT FAISS: its semantics on the HIP search.
    What differs from the sklearn branch and is reproduced here: `euclidean` weights / info["distances"] are SQUARED L2
    distances in float32 (IndexFlatL2.search), `cosine` = 1 - <x^, y^> with x^ = x / (|x| + 1e-8) in float32
    (IndexFlatIP), the self column is dropped only when it leads every row, there is no k = 0 / N = 1 special case, and
    any other metric raises ValueError.  Neighbours are ranked by the exact fp64 distance (FAISS ranks by a float32
    ||x||^2 - 2 x.y + ||y||^2: the two orders differ only between candidates closer than float32 rounding).
    PARITY UNPINNED: FAISS is not installed in the build container, so no fixture of the reference's FAISS output
    exists; tests compare with a numpy restatement of the published IndexFlat algorithm (oracle/knn.py)."""
    if not FAISS_SEMANTICS:          # the reference without FAISS installed (knn_graph_optimized.py:73-74): the pinned default
        raise RuntimeError("FAISS not available, falling back to sklearn")
    assert z.ndim == 2, "z must be (N,D)"
    if metric not in ("euclidean", "cosine"):
        raise ValueError(f"FAISS metric '{metric}' not supported. Use 'euclidean' or 'cosine'.")
    if sym not in _SYM_MODE:
        raise ValueError(f"Invalid symmetry mode: {sym}")
    N = z.shape[0]
    dev = device()
    z32 = np.ascontiguousarray(z, dtype=np.float32)
    if metric == "cosine":
        z32 = np.ascontiguousarray((z / (np.linalg.norm(z, axis=1, keepdims=True) + 1e-8)).astype(np.float32))
    z_dev = torch.from_numpy(z32).to(dev)
    idx, d2 = knn_search_device(z_dev, min(k + 1, N))
    if metric == "euclidean":
        dist32 = d2.to(torch.float32)
    else:       # inner product of the float32 rows from the exact squared distance: <x, y> = (|x|^2 + |y|^2 - |x - y|^2) / 2
        sq = (z_dev.double() ** 2).sum(dim=1)
        ip = 0.5 * (sq[:, None] + sq[idx.long()] - d2)
        dist32 = (1.0 - ip.to(torch.float32)).to(torch.float32)
    G, dist32, idx = _faiss_style_graph(idx, dist32, mode, sym)
    return G.to_scipy(), {"distances": dist32.cpu().numpy(), "indices": idx.cpu().numpy().astype(np.int64)}


# False (default): behave like the reference WITHOUT faiss installed -- the FAISS entry points raise its RuntimeErrors.
# True: behave like a reference WITH faiss: build_knn_graph_faiss answers, build_knn_graph_auto picks it for N >= threshold.
FAISS_SEMANTICS = False


def build_knn_graph_auto(z: np.ndarray, k: int = 10, metric: str = "euclidean", mode: str = "distance",
                         sym: str = "mutual", force_method: Optional[str] = None, size_threshold: int = 50000):
    """Method selection of knn_graph_optimized.py:129-170.  Every search is the exact HIP brute force; `method` only
    selects the SEMANTICS: "sklearn" (true distances, fp64-ranked: pinned by fixtures) or "faiss" (squared float32
    distances etc., see build_knn_graph_faiss: unpinned).  Without FAISS the reference always takes the sklearn branch,
    which is the default here too (force_method="faiss" raises as it does there); FAISS_SEMANTICS = True opts into the
    behaviour of a reference that has FAISS."""
    N = z.shape[0]
    if force_method == "sklearn":
        method = "sklearn"
    elif force_method == "faiss":
        if not FAISS_SEMANTICS:
            raise RuntimeError("force_method='faiss' but FAISS not available")
        method = "faiss"
    else:
        method = "faiss" if (FAISS_SEMANTICS and N >= size_threshold) else "sklearn"
    print(f"Building k-NN graph: N={N}, k={k}, method={method if method == 'faiss' else 'hip'}")
    if method == "faiss":
        return build_knn_graph_faiss(z, k=k, metric=metric, mode=mode, sym=sym)
    return build_knn_graph_sklearn(z, k=k, metric=metric, mode=mode, sym=sym)


def _undirected_structure(W: sparse.spmatrix, dev) -> DeviceCSR:
    """Stored pattern of W made symmetric (connected_components(directed=False) uses edges both ways)."""
    W = sparse.csr_matrix(W)
    P = sparse.csr_matrix((np.ones(W.nnz, np.float32), W.indices, W.indptr), shape=W.shape)
    S = (P + P.T).tocsr()
    S.sort_indices()
    return DeviceCSR.from_scipy(S, dev, with_data=False)


def largest_connected_component(W: sparse.csr_matrix) -> np.ndarray:
    """Boolean mask of the nodes in the largest connected component (knn_graph_optimized.py:173-181)."""
    if W.shape[0] == 0:
        return np.ones(0, dtype=bool)
    return lcc_mask_device(_undirected_structure(W, device())).cpu().numpy()


def analyze_graph_connectivity(W: sparse.csr_matrix) -> Dict:
    """Connectivity statistics with the reference's printout (knn_graph_optimized.py:184-219)."""
    N = W.shape[0]
    n_components, labels = connected_components_device(_undirected_structure(W, device()))
    if n_components > 1:
        largest_size = int(torch.bincount(labels.long()).max())
        connectivity_ratio = largest_size / N
    else:
        largest_size, connectivity_ratio = N, 1.0
    degrees = np.array(W.sum(axis=1)).flatten()
    stats = {"n_nodes": N, "n_edges": W.nnz, "n_components": n_components,
             "largest_component_size": largest_size, "connectivity_ratio": connectivity_ratio,
             "avg_degree": degrees.mean(), "min_degree": degrees.min(), "max_degree": degrees.max()}
    print("Graph connectivity")
    print(f"nodes={N} edges={W.nnz} avg_deg={stats['avg_degree']:.1f}")
    print(f"components={n_components} largest={largest_size} ({100*connectivity_ratio:.1f}%)")
    if n_components > 1:
        print("disconnected -> will use LCC")
    return stats


def build_knn_graph(z: np.ndarray, k: int = 10, metric: str = "euclidean", mode: str = "distance",
                    sym: str = "mutual"):
    """Backward compatible entry point (knn_graph_optimized.py:223-231)."""
    return build_knn_graph_auto(z, k=k, metric=metric, mode=mode, sym=sym)

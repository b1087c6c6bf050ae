"""Graph-based geodesic K-medoids on the MI355X -- same API as the reference's
src/geo/kmeans_optimized.py (kpp_initialization_graph :14, assign_points_to_medoids :77,
compute_quantization_error :109, fit_kmedoids_optimized :141, fit_kmedoids_with_connectivity_check :186).

The graph stays resident in HBM.  The whole k-means++ chain runs on the GPU (csrc/kpp.hip): pruned frontier
solves from each new centre, the running (min, first-argmin) update, and numpy's legacy RandomState.choice draw
restated bit for bit (float32 D^2 weights, float32 pairwise add.reduce, fp64 cdf, searchsorted); the host only
supplies the uniform deviates of the same RandomState stream and steps in for the rare draws the device declines
(u within rounding reach of a cdf step, degenerate weights).  GEO_KPP_HOST_DRAW=1 selects the reference's own
structure instead -- one device solve + one numpy draw on the downloaded d_min per centre (kmeans_optimized.py:47-69);
the tests compare both.
"""
import os
import time
from typing import List, Optional, Tuple

import numpy as np
import torch
from scipy import sparse

from .. import _lib
from .._device import DeviceCSR, device, ptr, stream_ptr, workspace
from .geo_shortest_paths import nearest_source_device, _pull_structure, ensure_valid_graph, sssp_multi_device


def _to_device_graph(W) -> DeviceCSR:
    if isinstance(W, DeviceCSR):
        return W
    W = ensure_valid_graph(W)
    return DeviceCSR.from_scipy(_pull_structure(W, directed=False), device())


# Tuning switches of the chain, read ONCE at import (tests flip them through _KNOBS, bench leaves them alone).
_KNOBS = {"resident": os.environ.get("GEO_KPP_RESIDENT", "1") != "0",
          "resident_from": int(os.environ.get("GEO_KPP_RESIDENT_FROM", "0")),       # 0: by graph size, see _resident_from
          "log": os.environ.get("GEO_KPP_LOG", "0") == "1"}


def _resident_from(n: int) -> int:
    """First centre the resident workgroup takes over.  Centre t claims about n / t nodes; the workgroup's LDS table holds
    4096, and below ~1000 nodes per cell it beats the multi-workgroup step kernel (60 000 nodes: 64 gave 21.2 ms for the
    chain, 32 22.3 ms -- two early cells outgrew the table and were handed back -- 96 21.4 ms)."""
    return _KNOBS["resident_from"] or max(32, n // 1000)


class _Chain:
    """Running (min, first-argmin) over single-source solves: d_min of kmeans_optimized.py:36,44."""

    def __init__(self, G: DeviceCSR):
        self.G, self.lib = G, _lib.load()
        dev = G.indptr.device
        self.dmin = torch.full((G.n,), float("inf"), dtype=torch.float32, device=dev)
        self.arg = torch.zeros(G.n, dtype=torch.int32, device=dev)
        self.ws = workspace(self.lib.geo_sssp_workspace_bytes(G.n, G.nnz, 1), dev)
        self.host = torch.empty(G.n, dtype=torch.float32, pin_memory=True)
        self.sweeps = 0
        self.solves = 0

    def absorb(self, source: int, pos: int) -> np.ndarray:
        G = self.G
        sw = np.zeros(1, dtype=np.int32)
        with torch.cuda.device(G.indptr.device):
            _lib.check(self.lib.geo_sssp_single_update(
                ptr(G.indptr), ptr(G.indices), ptr(G.data), G.n, int(source), None, ptr(self.dmin),
                ptr(self.arg), int(pos), ptr(self.ws), self.ws.numel(), sw.ctypes.data, stream_ptr()),
                "geo_sssp_single_update")
            self.host.copy_(self.dmin, non_blocking=True)
            torch.cuda.current_stream().synchronize()
        self.sweeps += int(sw[0])
        self.solves += 1
        return self.host.numpy()


def _next_center(rng, N: int, d_min: np.ndarray, centers: List[int]) -> Optional[int]:
    """One k-means++ draw, kmeans_optimized.py:47-69 (d_min is the float32 host copy)."""
    finite = np.isfinite(d_min)
    if finite.any():
        safe = np.where(finite, d_min, np.max(d_min[finite]) * 2.0)
    else:
        safe = np.ones_like(d_min)
    probs = safe ** 2
    probs[centers] = 0.0
    total = probs.sum()
    if total > 0:
        probs /= total
        return int(rng.choice(N, p=probs))
    taken = set(centers)
    rest = [i for i in range(N) if i not in taken]
    if rest:
        return int(rng.choice(rest))
    return None


def _draw_with_u(N: int, d_min: np.ndarray, centers: List[int], u: float):
    """The draw of kmeans_optimized.py:47-61 with the uniform deviate given: RandomState.choice(N, p=probs) is
    cdf = cumsum(float64(p)); cdf /= cdf[-1]; searchsorted(cdf, random_sample(), 'right').  Returns None when
    the weights are degenerate (sum == 0), which the caller resolves with the reference's fallback.
    Same values with fewer passes over the N entries (this runs when the device declines a draw, about once per chain
    at a million latents): no copy when every distance is finite, the cumulative sum taken in float64 directly, and
    the division by cdf[-1] -- monotone, so it cannot reorder the entries -- applied only around the answer."""
    finite = np.isfinite(d_min)
    if finite.all():
        safe = d_min
    elif finite.any():
        safe = np.where(finite, d_min, np.max(d_min[finite]) * 2.0)
    else:
        safe = np.ones_like(d_min)
    probs = safe * safe                                   # == safe ** 2 (numpy squares by multiplying)
    probs[centers] = 0.0
    total = probs.sum()
    if not total > 0:
        return None
    probs /= total
    cdf = np.cumsum(probs, dtype=np.float64)              # == probs.astype(float64).cumsum()
    last = cdf[-1]
    # searchsorted(cdf / last, u, 'right') = number of entries with cdf[j] / last <= u: start from the position in the
    # undivided array and settle it with the exact quotients of its neighbours
    j = int(cdf.searchsorted(u * last, side="right"))
    while j < N and cdf[j] / last <= u:
        j += 1
    while j > 0 and not (cdf[j - 1] / last <= u):
        j -= 1
    return j


def _kpp_chain_host(G: DeviceCSR, K: int, seed: int, absorb_last: bool):
    """Host-sampled chain: one device solve + one numpy draw per centre (reference structure)."""
    N = G.n
    rng = np.random.RandomState(seed)
    centers = [int(rng.randint(0, N))]
    chain = _Chain(G)
    complete = True
    for _ in range(1, K):
        d_min = chain.absorb(centers[-1], len(centers) - 1)
        nxt = _next_center(rng, N, d_min, centers)
        if nxt is None:
            print(f"Warning: Could not find {K} valid centers, stopping at {len(centers)}")
            complete = False
            break
        centers.append(nxt)
    if absorb_last and complete:
        chain.absorb(centers[-1], len(centers) - 1)     # the last centre gets no solve inside k++
    return centers, chain


def _kpp_chain_device(G: DeviceCSR, K: int, seed: int, absorb_last: bool):
    """Device-resident chain (csrc/kpp.hip): the solves, the d_min update and numpy's draw all stay on the
    GPU; the host only supplies the uniform deviates of the same RandomState stream and steps in when the
    kernel declines a draw (u within rounding reach of a cdf boundary) or a solve needs more sweeps."""
    lib = _lib.load()
    N, dev = G.n, G.indptr.device
    rng = np.random.RandomState(seed)
    first = int(rng.randint(0, N))
    # One RandomState, advanced in lock-step with the committed draws: u[t] is the deviate the reference's
    # rng.choice(N, p=probs) consumes for centre t+1.  The deviates are pre-drawn from `base_state` (the stream position
    # before draw `base_t`); a uniform fallback (degenerate weights) consumes the stream differently, so it is replayed
    # from base_state and the remaining deviates are drawn afresh from the position it leaves behind.
    base_state, base_t = rng.get_state(), 0
    u = rng.random_sample(max(K - 1, 0)) if K > 1 else np.zeros(0)
    u = np.ascontiguousarray(u, dtype=np.float64)
    chain = _Chain(G)           # chain.ws and `ws` below alias the same cached workspace buffer: geo_kpp_chain
    #                             re-initialises all of its workspace state on every call, and chain.absorb (which
    #                             overwrites it) only runs between two geo_kpp_chain calls, never during one
    centers_d = torch.zeros(max(K, 1), dtype=torch.int32, device=dev)
    centers_d[0] = first
    is_center = torch.zeros(N, dtype=torch.uint8, device=dev)
    is_center[first] = 1
    ws = workspace(lib.geo_kpp_workspace_bytes(N), dev)
    it, it1 = 0, (K if absorb_last else K - 1)
    n_valid = K
    # While d_min still has unreachable (inf) entries the chain runs as separate kernels per phase with a sweep
    # budget per solve: a solve exits early once converged, but every enqueued launch costs ~2 us, and a solve that
    # needs more than were enqueued aborts and is redone.  The need is the hop radius of the new centre's (pruned)
    # cell: the whole graph for the first centre, then shrinking -- segments of doubling length, each taking its
    # budget from what the previous one needed.  Once d_min is finite everywhere (after the first centre on a
    # connected graph) the rest of the chain is ONE kernel launched over and over that always does "the next
    # step" (csrc/kpp.hip, kpp_step_kernel): no budget, no launch spent on an empty frontier.
    # Later centres have small cells (~N/t nodes): from RESIDENT_FROM on, the chain runs inside ONE resident
    # workgroup (kpp_resident_kernel), a single launch for all remaining centres; a cell that outgrows its LDS table
    # comes back with reason 4 and that one centre is run by the step kernel.
    fixed = os.environ.get("GEO_KPP_SWEEPS")
    budget, seg, cap = (int(fixed) if fixed else 16), 1, 4094
    finite = False
    resident_ok = _KNOBS["resident"] and N <= lib.geo_kpp_resident_max_nodes()
    one_step_at = -1                                     # iteration the resident kernel handed back (reason 4)
    status = np.zeros(4, dtype=np.int32)
    while it < it1:
        step_mode = finite and not fixed and K <= N
        r_from = _resident_from(N)
        resident = step_mode and resident_ok and it >= min(r_from, it1) and it != one_step_at
        if resident:
            seg_end = it1
        elif step_mode:
            seg_end = it + 1 if it == one_step_at else (min(it1, r_from) if resident_ok and it < r_from else it1)
        else:
            seg_end = min(it1, it + seg)
        t_call = time.perf_counter()
        with torch.cuda.device(dev):
            _lib.check(lib.geo_kpp_chain(ptr(G.indptr), ptr(G.indices), ptr(G.data), N, ptr(centers_d),
                                         ptr(is_center), ptr(chain.dmin), ptr(chain.arg), u.ctypes.data, it, seg_end,
                                         K, (-1 if resident else 0) if step_mode else budget, 1 if finite else 0, ptr(ws),
                                         ws.numel(), status.ctypes.data, stream_ptr()),
                       "geo_kpp_chain")
        t, reason, used = int(status[0]), int(status[1]), int(status[3])
        if _KNOBS["log"]:
            import sys
            print(f"[kpp-log] it {it}..{seg_end} mode {'resident' if resident else 'step' if step_mode else budget} -> "
                  f"abort {t} reason {reason} used {used} {1e3 * (time.perf_counter() - t_call):.2f} ms", file=sys.stderr)
        finite = finite or int(status[2]) == 0          # inf entries only ever disappear from d_min
        if reason == 4:                                  # cell too large for the resident table: that centre by step kernel
            chain.solves += t - it
            it, one_step_at = t, t
            continue
        if t < 0:
            chain.solves += seg_end - it
            it = seg_end
            seg = min(2 * seg, 256)
            if not fixed and not step_mode:
                budget = min(cap, max(4, used + used // 8 + 1))     # cells shrink: the next segment needs no more
            continue
        if reason == 1 and not step_mode and budget < cap:   # nothing of solve t was applied: redo it with more sweeps
            chain.solves += t - it
            budget, seg, it = min(cap, 4 * budget), 1, t
            continue
        chain.solves += t - it + 1
        centers_h = centers_d[: t + 1].cpu().numpy().astype(int).tolist()
        if reason == 1:                                  # beyond the device budget: host-driven solve
            chain.absorb(centers_h[t], t)
        if t + 1 >= K:
            break
        nxt = _draw_with_u(N, chain.dmin.cpu().numpy(), centers_h, float(u[t]))
        if nxt is None:                                  # degenerate weights: the reference's uniform fallback
            rng.set_state(base_state)
            if t > base_t:
                rng.random_sample(t - base_t)            # the p-weighted draws committed since base_t
            taken = set(centers_h)
            rest = [i for i in range(N) if i not in taken]
            if not rest:
                print(f"Warning: Could not find {K} valid centers, stopping at {len(centers_h)}")
                n_valid = t + 1
                break
            nxt = int(rng.choice(rest))                  # kmeans_optimized.py:66
            base_state, base_t = rng.get_state(), t + 1
            if K - 2 - t > 0:
                u[t + 1:] = rng.random_sample(K - 2 - t)
        centers_d[t + 1] = nxt
        is_center[nxt] = 1
        it = t + 1
    centers = centers_d[:n_valid].cpu().numpy().astype(int).tolist()
    return centers, chain


def _kpp_chain(G: DeviceCSR, K: int, seed: int, absorb_last: bool):
    print(f"[kpp] Selecting {K} centers among {G.n} nodes")
    if os.environ.get("GEO_KPP_HOST_DRAW", "0") == "1":
        centers, chain = _kpp_chain_host(G, K, seed, absorb_last)
    else:
        centers, chain = _kpp_chain_device(G, K, seed, absorb_last)
    print(f"[kpp] Selected {len(centers)} centers")
    return centers, chain


def kpp_initialization_graph(W, K: int, seed: int = 42) -> List[int]:
    """K-means++ initialisation over graph distances (kmeans_optimized.py:14-74)."""
    centers, _ = _kpp_chain(_to_device_graph(W), K, seed, absorb_last=False)
    return centers


def _assign_device(G: DeviceCSR, medoids: np.ndarray):
    # nearest medoid + distance in ONE label-carrying solve (geo_sssp_nearest_source: exact fixed-point units), the K-source
    # solve only when the weights do not qualify
    src = torch.from_numpy(np.asarray(medoids, dtype=np.int32)).to(G.indptr.device)
    dmin, arg, _ = nearest_source_device(G, src)
    return dmin, arg


def _print_sizes(assign: np.ndarray, K: int) -> None:
    counts = np.bincount(assign, minlength=K)
    print(f"[assign] sizes min={counts.min()}, max={counts.max()}, mean={counts.mean():.1f}")


def assign_points_to_medoids(W, medoids: np.ndarray) -> np.ndarray:
    """Nearest medoid per node, first index on ties (kmeans_optimized.py:77-106)."""
    G = _to_device_graph(W)
    K = len(medoids)
    print(f"[assign] {G.n} points to {K} medoids")
    _, arg = _assign_device(G, medoids)
    assign = arg.cpu().numpy().astype(int)
    _print_sizes(assign, K)
    return assign


def _qe_from(dist_to_assigned: np.ndarray) -> float:
    finite = np.isfinite(dist_to_assigned)
    if finite.any():
        return float(np.sum(dist_to_assigned[finite] ** 2))
    return float("inf")


def compute_quantization_error(W, medoids: np.ndarray, assign: np.ndarray) -> float:
    """Sum of squared geodesic distances to the assigned medoid (kmeans_optimized.py:109-138)."""
    G = _to_device_graph(W)
    dev = G.indptr.device
    src = torch.from_numpy(np.asarray(medoids, dtype=np.int32)).to(dev)
    a = torch.from_numpy(np.asarray(assign, dtype=np.int64)).to(dev)
    # the usual call hands in the nearest-medoid assignment: then D[assign[v]][v] IS the column minimum, which one
    # label-carrying solve gives (geo_sssp_nearest_source) -- no K x N matrix; any other assignment takes the matrix
    dmin, amin, _ = nearest_source_device(G, src)
    if bool((amin.long() == a).all()):
        return _qe_from(dmin.cpu().numpy())
    D, _, _, _, _ = sssp_multi_device(G, src, want_D=True)
    picked = D.gather(0, a.view(1, -1)).view(-1)
    return _qe_from(picked.cpu().numpy())


def fit_kmedoids_optimized(W, K: int = 512, init: str = "kpp", seed: int = 42) -> Tuple[np.ndarray, np.ndarray, float]:
    """Graph-based geodesic K-medoids (kmeans_optimized.py:141-183).

    init="kpp" needs K solves instead of the reference's 3K-1: the running minimum / first-argmin
    kept during seeding plus one solve for the last centre IS the assignment and its distances."""
    if init not in ("kpp", "random"):
        raise ValueError("init must be 'kpp' or 'random'")
    G = _to_device_graph(W)
    N = G.n
    print(f"[kmedoids] N={N}, K={K}, edges={G.nnz}, avg_deg={G.nnz/max(1,N):.1f}")
    if init == "kpp":
        centers, chain = _kpp_chain(G, K, seed, absorb_last=True)
        medoids = np.array(centers, dtype=int)
        print(f"[assign] {N} points to {len(medoids)} medoids")
        assign = chain.arg.cpu().numpy().astype(int)
        d_assigned = chain.dmin.cpu().numpy()
    else:
        rng = np.random.RandomState(seed)
        medoids = rng.choice(N, size=min(K, N), replace=False).astype(int)
        print(f"[assign] {N} points to {len(medoids)} medoids")
        dmin, arg = _assign_device(G, medoids)
        assign = arg.cpu().numpy().astype(int)
        d_assigned = dmin.cpu().numpy()
    _print_sizes(assign, len(medoids))
    qe = _qe_from(d_assigned)
    print(f"[kmedoids] Done: clusters={len(medoids)}, qe={qe:.3f}")
    return medoids, assign, qe


def fit_kmedoids_with_connectivity_check(W, K: int = 512, init: str = "kpp", seed: int = 42):
    """K-medoids plus connectivity metadata (kmeans_optimized.py:186-227)."""
    from .knn_graph_optimized import connected_components_device
    G = _to_device_graph(W)
    n_components, labels = connected_components_device(G)
    sizes = np.bincount(labels.cpu().numpy())
    metadata = {"n_nodes": G.n, "n_edges": G.nnz, "n_components": n_components,
                "largest_component_size": sizes.max() if n_components > 0 else G.n}
    print(f"[graph] components={n_components}, largest={metadata['largest_component_size']}")
    medoids, assign, qe = fit_kmedoids_optimized(G, K=K, init=init, seed=seed)
    metadata.update({"n_medoids": len(medoids), "quantization_error": qe, "method": "optimized_kmedoids"})
    return medoids, assign, qe, metadata


# ---- extension: medoid update (Voronoi iteration) over the resident all-pairs matrix -------------------------------------
def medoid_update_device(D: torch.Tensor, assign: torch.Tensor, medoids: torch.Tensor, power: int = 2):
    """One medoid update: per cluster the member with the smallest sum of D[i][j]^power over the cluster's members
    (lowest node index on ties); a cluster without members keeps its medoid.  D f32 [n,n], assign i32/i64 [n],
    medoids i32 [K], all on the GPU.  Returns (new medoids i32 [K], cost f64 [n])."""
    lib = _lib.load()
    dev = D.device
    n, K = int(D.shape[0]), int(medoids.numel())
    a64 = assign.to(torch.int64)
    a32 = assign.to(torch.int32).contiguous()
    order = torch.argsort(a64, stable=True).to(torch.int32).contiguous()        # members ascending inside a cluster
    counts = torch.bincount(a64, minlength=K)
    offsets = torch.zeros(K + 1, dtype=torch.int32, device=dev)
    offsets[1:] = torch.cumsum(counts, 0).to(torch.int32)
    cost = torch.empty(n, dtype=torch.float64, device=dev)
    with torch.cuda.device(dev):
        _lib.check(lib.geo_cluster_costs(ptr(D), D.stride(0), ptr(a32), ptr(order), ptr(offsets), n, int(power),
                                         ptr(cost), stream_ptr()), "geo_cluster_costs")
    cmin = torch.full((K,), float("inf"), dtype=torch.float64, device=dev).scatter_reduce(0, a64, cost, "amin")
    idx = torch.arange(n, dtype=torch.int64, device=dev)
    cand = torch.where(cost == cmin[a64], idx, torch.full_like(idx, n))
    first = torch.full((K,), n, dtype=torch.int64, device=dev).scatter_reduce(0, a64, cand, "amin")
    new = torch.where(first < n, first, medoids.to(torch.int64)).to(torch.int32)
    return new, cost


def assign_from_rows_device(D: torch.Tensor, medoids: torch.Tensor):
    """(dmin f32 [n], argmin i32 [n]): nearest medoid row of D per node, first medoid on ties."""
    lib = _lib.load()
    dev = D.device
    n = int(D.shape[1])
    rows = medoids.to(torch.int32).contiguous()
    dmin = torch.empty(n, dtype=torch.float32, device=dev)
    arg = torch.empty(n, dtype=torch.int32, device=dev)
    with torch.cuda.device(dev):
        _lib.check(lib.geo_rows_argmin(ptr(D), D.stride(0), ptr(rows), int(rows.numel()), n, ptr(dmin), ptr(arg),
                                       stream_ptr()), "geo_rows_argmin")
    return dmin, arg


def fit_kmedoids_voronoi(W, K: int = 512, init: str = "kpp", seed: int = 42, max_iter: int = 10, power: int = 2,
                         D: Optional[torch.Tensor] = None):
    """fit_kmedoids_optimized followed by medoid updates until the medoids stop changing (at most max_iter updates).

    Extension of the reference (its k-medoids is seeding + one assignment, kmeans_optimized.py:141-183; SURVEY.md section 8
    f4): the all-pairs geodesic matrix is formed once on the GPU (all_pairs_geodesic_device) unless given, every
    iteration is one geo_cluster_costs + one geo_rows_argmin.  power=2 minimises the reference's quantisation error
    (sum of squared distances), which therefore never increases.  Returns (medoids, assign, qe, history) with
    history = [qe after the initial assignment, after update 1, ...]."""
    from .geo_shortest_paths import all_pairs_geodesic_device
    G = _to_device_graph(W)
    dev = G.indptr.device
    medoids0, assign0, qe0 = fit_kmedoids_optimized(W, K=K, init=init, seed=seed)
    if D is None:
        D = all_pairs_geodesic_device(G)
    med = torch.from_numpy(np.asarray(medoids0, dtype=np.int32)).to(dev)
    assign = torch.from_numpy(np.asarray(assign0, dtype=np.int32)).to(dev)
    history = [qe0]
    for _ in range(max_iter):
        new, _ = medoid_update_device(D, assign, med, power)
        if bool((new == med).all()):
            break
        med = new
        dmin, assign = assign_from_rows_device(D, med)
        history.append(_qe_from(dmin.cpu().numpy()))
    medoids = med.cpu().numpy().astype(int)
    assign_h = assign.cpu().numpy().astype(int)
    _print_sizes(assign_h, len(medoids))
    print(f"[kmedoids] Voronoi iterations: {len(history) - 1}, qe {history[0]:.3f} -> {history[-1]:.3f}")
    return medoids, assign_h, history[-1], history



# ---- extension: PAM swap over the resident all-pairs matrix ---------------------------------------------------------------
def pam_swap_pass_device(D: torch.Tensor, medoids: torch.Tensor, power: int = 2):
    """PAM's SWAP evaluation for ALL (medoid, candidate) pairs in one pass over the resident matrix (csrc/medoid.hip:
    pam_swap_kernel, FastPAM1 form).  Returns (delta, i, x, total): the swap medoids[i] -> x with the most negative change of
    the total cost sum_j D[nearest(j)][j]^power (ties: lowest x, then first medoid), and the current total cost."""
    lib = _lib.load()
    dev = D.device
    n, K = int(D.shape[0]), int(medoids.numel())
    med = medoids.to(torch.int64)
    dmin, near = assign_from_rows_device(D, medoids)                 # nearest medoid, first on ties
    near = near.to(torch.int64)
    c1 = dmin.double() ** power
    if K == 1:                                                       # nothing to fall back on: the swap replaces the only medoid
        tot = torch.cat([(D[r0:r0 + 4096].double() ** power).sum(dim=1) for r0 in range(0, n, 4096)])
        tot[med] = float("inf")
        delta = tot.min() - c1.sum()
        x = int(torch.nonzero(tot == tot.min())[0])
        return float(delta), 0, x, float(c1.sum())
    rows = D[med]                                                    # K x n float32 (a copy)
    rows.scatter_(0, near[None, :], float("inf"))
    d2 = rows.min(dim=0).values
    del rows
    # base[i] = sum over the nodes of medoid i of (c2 - c1), members in ascending node order, sequential fp64 sums
    # (segment sums through the sorted order: deterministic)
    gain = d2.double() ** power - c1
    order = torch.argsort(near, stable=True)
    counts = torch.bincount(near, minlength=K)
    base = torch.segment_reduce(gain[order], "sum", lengths=counts)
    is_med = torch.zeros(n, dtype=torch.uint8, device=dev)
    is_med[med] = 1
    near32, d1, d2 = near.to(torch.int32).contiguous(), dmin.contiguous(), d2.contiguous()
    best = torch.empty(n, dtype=torch.float64, device=dev)
    which = torch.empty(n, dtype=torch.int32, device=dev)
    with torch.cuda.device(dev):
        _lib.check(lib.geo_pam_swap_deltas(ptr(D), D.stride(0), ptr(near32), ptr(d1), ptr(d2), ptr(base.contiguous()), ptr(is_med), n, K,
                                           int(power), ptr(best), ptr(which), stream_ptr()), "geo_pam_swap_deltas")
    delta = best.min()
    x = int(torch.nonzero(best == delta)[0])                         # lowest candidate among equal changes
    return float(delta), int(which[x]), x, float(c1.sum())


def fit_kmedoids_pam(W, K: int = 512, init: str = "kpp", seed: int = 42, max_swaps: int = 50, power: int = 2,
                     D: Optional[torch.Tensor] = None, rel_tol: float = 1e-12):
    """fit_kmedoids_optimized followed by PAM swaps: the best (medoid -> non-medoid) exchange is applied while it lowers the
    total cost (power=2: the reference's quantisation error).  Extension (the reference's k-medoids
    is seeding + one assignment, kmeans_optimized.py:141-183).  Every swap evaluation reads the resident N x N matrix once.
    Returns (medoids, assign, qe, history of total costs)."""
    from .geo_shortest_paths import all_pairs_geodesic_device
    G = _to_device_graph(W)
    dev = G.indptr.device
    medoids0, _, _ = fit_kmedoids_optimized(W, K=K, init=init, seed=seed)
    if D is None:
        D = all_pairs_geodesic_device(G)
    med = torch.from_numpy(np.asarray(medoids0, dtype=np.int32)).to(dev)
    history = []
    for _ in range(max_swaps + 1):
        delta, i, x, total = pam_swap_pass_device(D, med, power)
        history.append(total)
        if len(history) > max_swaps or not delta < -rel_tol * total:
            break
        med[i] = x
    dmin, assign = assign_from_rows_device(D, med)
    medoids = med.cpu().numpy().astype(int)
    assign_h = assign.cpu().numpy().astype(int)
    _print_sizes(assign_h, len(medoids))
    print(f"[kmedoids] PAM swaps: {len(history) - 1}, total cost {history[0]:.3f} -> {history[-1]:.3f}")
    return medoids, assign_h, _qe_from(dmin.cpu().numpy()), history

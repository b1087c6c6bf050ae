"""CPU restatement of src/geo/kmeans_optimized.py (reference).  TEST INFRASTRUCTURE ONLY.

Two drivers are given: `fit_kmedoids_optimized` follows the reference's three stages literally
(3K-1 shortest-path solves), `fit_kmedoids_single_pass` is the K-solve formulation the HIP path
uses (running min / first-argmin during seeding); tests assert they agree bit for bit.
"""
from typing import List

import numpy as np

from .sssp import dijkstra_multi_source, dijkstra_single_source


def _seeding_probs(d_min: np.ndarray, centers: List[int]) -> np.ndarray:
    """kmeans_optimized.py:47-57: float32 D^2 weights with the inf -> 2*max_finite substitution."""
    finite = np.isfinite(d_min)
    if finite.any():
        safe = np.where(finite, d_min, np.max(d_min[finite]) * 2.0)
    else:
        safe = np.ones_like(d_min)
    probs = safe ** 2
    probs[centers] = 0.0
    return probs


def _draw_next(rng, N: int, probs: np.ndarray, centers: List[int]):
    """kmeans_optimized.py:59-69.  Returns None when no candidate is left."""
    total = probs.sum()
    if total > 0:
        probs /= total
        return int(rng.choice(N, p=probs))
    rest = [i for i in range(N) if i not in centers]
    if rest:
        return int(rng.choice(rest))
    return None


def kpp_initialization_graph(W, K: int, seed: int = 42) -> List[int]:
    """kmeans_optimized.py:14-74."""
    N = W.shape[0]
    rng = np.random.RandomState(seed)
    centers = [int(rng.randint(0, N))]
    d_min = np.full(N, np.inf, dtype=np.float32)
    for _ in range(1, K):
        d_min = np.minimum(d_min, dijkstra_single_source(W, centers[-1], dtype=np.float32))
        nxt = _draw_next(rng, N, _seeding_probs(d_min, centers), centers)
        if nxt is None:
            break
        centers.append(nxt)
    return centers


def assign_points_to_medoids(W, medoids) -> np.ndarray:
    """kmeans_optimized.py:77-106."""
    D = dijkstra_multi_source(W, medoids, dtype=np.float32)
    return D.argmin(axis=0).astype(int)


def quantization_error_from(dist_to_assigned: np.ndarray) -> float:
    """kmeans_optimized.py:131-136: float32 squares, numpy sum over the finite ones."""
    finite = np.isfinite(dist_to_assigned)
    if finite.any():
        return float(np.sum(dist_to_assigned[finite] ** 2))
    return float("inf")


def compute_quantization_error(W, medoids, assign) -> float:
    """kmeans_optimized.py:109-138."""
    D = dijkstra_multi_source(W, medoids, dtype=np.float32)
    return quantization_error_from(D[assign, np.arange(len(assign))])


def _init_medoids(W, K, init, seed):
    if init == "kpp":
        return np.array(kpp_initialization_graph(W, K, seed=seed), dtype=int)
    if init == "random":
        rng = np.random.RandomState(seed)
        return rng.choice(W.shape[0], size=min(K, W.shape[0]), replace=False).astype(int)
    raise ValueError("init must be 'kpp' or 'random'")


def fit_kmedoids_optimized(W, K: int = 512, init: str = "kpp", seed: int = 42):
    """kmeans_optimized.py:141-183, stage by stage."""
    medoids = _init_medoids(W, K, init, seed)
    assign = assign_points_to_medoids(W, medoids)
    qe = compute_quantization_error(W, medoids, assign)
    return medoids, assign, qe


def fit_kmedoids_single_pass(W, K: int = 512, seed: int = 42):
    """Same outputs as fit_kmedoids_optimized(init='kpp') from K solves instead of 3K-1."""
    N = W.shape[0]
    rng = np.random.RandomState(seed)
    centers = [int(rng.randint(0, N))]
    d_min = np.full(N, np.inf, dtype=np.float32)
    arg = np.zeros(N, dtype=int)

    def absorb(pos):
        nonlocal d_min
        d = dijkstra_single_source(W, centers[pos], dtype=np.float32)
        better = d < d_min                      # strict: the first minimal row index wins
        arg[better] = pos
        d_min = np.where(better, d, d_min)

    for _ in range(1, K):
        absorb(len(centers) - 1)
        nxt = _draw_next(rng, N, _seeding_probs(d_min, centers), centers)
        if nxt is None:
            break
        centers.append(nxt)
    else:
        absorb(len(centers) - 1)                # the last centre gets no solve inside k++
    # (after an early break every listed centre has already been absorbed)
    return np.array(centers, dtype=int), arg, quantization_error_from(d_min)


# ---- extension without a reference implementation: medoid update over the all-pairs matrix ------
def all_pairs(W) -> np.ndarray:
    """f32 [n, n]: scipy Dijkstra from every node (dijkstra_multi_source over range(n))."""
    from .sssp import dijkstra_multi_source
    return dijkstra_multi_source(W, np.arange(W.shape[0]))


def _tree_sum_rows(X: np.ndarray) -> np.ndarray:
    """Row sums of fp64 X [r, m] in the order of geo_cluster_costs: 64 strided partial sums (element l, l+64, ...
    added in that order), then the xor butterfly 32, 16, ..., 1."""
    r, m = X.shape
    acc = np.zeros((r, 64), dtype=np.float64)
    for b in range(0, m, 64):
        blk = X[:, b:b + 64]
        acc[:, :blk.shape[1]] = acc[:, :blk.shape[1]] + blk
    lanes = np.arange(64)
    off = 32
    while off >= 1:
        acc = acc + acc[:, lanes ^ off]
        off >>= 1
    return acc[:, 0]


def medoid_update(D: np.ndarray, assign: np.ndarray, medoids: np.ndarray, power: int = 2):
    """Per cluster the member with the smallest sum of D[i][j]^power over the members (lowest index on ties);
    an empty cluster keeps its medoid.  Returns (new medoids, cost f64 [n])."""
    n, K = D.shape[0], len(medoids)
    cost = np.zeros(n, dtype=np.float64)
    new = np.array(medoids, dtype=np.int64).copy()
    for c in range(K):
        members = np.flatnonzero(assign == c)
        if members.size == 0:
            continue
        sub = D[np.ix_(members, members)].astype(np.float64)
        if power == 2:
            sub = sub * sub
        cost[members] = _tree_sum_rows(sub)
        new[c] = members[int(np.argmin(cost[members]))]
    return new, cost


def voronoi_iteration(W, medoids0, max_iter: int = 10, power: int = 2, D: np.ndarray = None):
    """Assignment / medoid update until the medoids stop changing.  Returns (medoids, assign, qe, history)."""
    if D is None:
        D = all_pairs(W)
    med = np.asarray(medoids0, dtype=np.int64)
    assign = np.argmin(D[med], axis=0)
    history = [quantization_error_from(D[med][assign, np.arange(D.shape[0])])]
    for _ in range(max_iter):
        new, _ = medoid_update(D, assign, med, power)
        if np.array_equal(new, med):
            break
        med = new
        assign = np.argmin(D[med], axis=0)
        history.append(quantization_error_from(D[med][assign, np.arange(D.shape[0])]))
    return med, assign, history[-1], history



# ---- extension: PAM swap (no reference implementation: parity against this restatement only) ---------------------------
def total_cost(D: np.ndarray, medoids, power: int = 2) -> float:
    """sum_j (distance of j to its nearest medoid)^power, fp64."""
    return float((D[np.asarray(medoids)].astype(np.float64).min(axis=0) ** power).sum())


def pam_swap_pass(D: np.ndarray, medoids, power: int = 2):
    """The best single swap (medoid position i -> node x) BY DEFINITION: every (i, x) pair is tried and the total cost
    recomputed.  Returns (delta, i, x) with the most negative delta; ties: lowest x, then lowest i.  O(K n * K n): small
    inputs only."""
    med = np.asarray(medoids, dtype=np.int64)
    n, K = D.shape[0], len(med)
    base = total_cost(D, med, power)
    best = (np.inf, 0, 0)
    is_med = np.zeros(n, bool)
    is_med[med] = True
    Dp = D.astype(np.float64) ** power
    for x in range(n):
        if is_med[x]:
            continue
        for i in range(K):
            trial = med.copy()
            trial[i] = x
            delta = float(Dp[trial].min(axis=0).sum()) - base
            if delta < best[0]:
                best = (delta, i, x)
    return best


def pam(D: np.ndarray, medoids0, power: int = 2, max_swaps: int = 100, rel_tol: float = 1e-12):
    """Classic PAM: apply the best swap while it lowers the total cost.  Returns (medoids, assign, cost history)."""
    med = np.asarray(medoids0, dtype=np.int64).copy()
    history = [total_cost(D, med, power)]
    for _ in range(max_swaps):
        delta, i, x = pam_swap_pass(D, med, power)
        if not delta < -rel_tol * history[-1]:
            break
        med[i] = x
        history.append(total_cost(D, med, power))
    return med, np.argmin(D[med], axis=0), history

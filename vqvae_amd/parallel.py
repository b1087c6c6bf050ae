"""Single-node multi-GPU sharding of the geodesic-codebook path: one process per
GPU, `torch.distributed` collectives (backend "nccl" = RCCL over xGMI on the MI355X node, "gloo" in the
CPU tests).  The reference has no distributed code; this layer is new.

What shards and what is exchanged
  kNN             query rows are block-sharded, the corpus is replicated (all-gather of the latent row
                  shards, N*d*4 bytes) -> all-gather of the neighbour lists (N*(k+1)*4 bytes; the fp64 distances,
                  another N*(k+1)*8 bytes, only for mode="distance" graphs or when duplicates displace a self match).
  edge lengths    sharded by CHUNK index (a chunk of `batch_size` consecutive edges is one BatchNorm batch,
                  so train-mode statistics do not change) -> all-gather of the lengths (E*4 bytes).
  K-source solve  sources are block-sharded -> every rank reduces its sources to (dmin, argmin) per node ->
                  all-gather (world*N*8 bytes) and a (min, lowest source index) merge: the first-index tie
                  rule of D.argmin(axis=0) (kmeans_optimized.py:100) is not an all-reduce(min).
  k-means++ chain inherently serial (each draw depends on all previous solves): replicated on every rank;
                  it is deterministic, so all ranks hold the same centres without communication.

The functions take the local compute as a callable, so the distributed logic is the same code on the GPU
(HIP kernels) and in the gloo tests (which plug in the CPU oracle as the compute).

Several sharded builds in flight per rank (`CollectiveOrder`): a communicator executes collectives in the order they are
issued, and every rank must issue them in the SAME order.  With one host thread per build that order is no longer program
order, so each collective carries a ticket that all ranks compute alike, and a thread issues its collective only when every
smaller ticket has been issued or given up.  One communicator, one order: nothing here relies on concurrent communicators.
"""
import threading
from typing import Callable, List, Optional, Tuple

import torch
import torch.distributed as dist

# collectives of one build, in program order; the last two follow the serial k-means++ chain
STAGES = ("latents", "knn_idx", "knn_d2", "edge_lengths", "bn_fold", "assign_d", "assign_a")
LATE = {"assign_d", "assign_a"}


class CollectiveOrder:
    """Issue order of the collectives of builds 0 .. n_builds-1 when `depth` of them are in flight on every rank.

    Ticket of (build i, stage s) = (i + lag(s), index of s, i) with lag = depth - 1 for the stages behind the chain and 0 for
    the others: the early collectives of build i + depth - 1 (which starts when build i - 1 ends) come before the final
    ones of build i, so a build's kNN / JVP exchange never queues behind an older build's chain.  A build issues its stages
    in program order and may skip some (e.g. the fp64 kNN distances); a stage counts as given up once its build has moved
    past it or finished.  Deadlock-free: a waiting ticket only waits for builds that are already in flight (run_pipelined
    hands out builds in order, `depth` at a time) and whose progress does not depend on it."""

    def __init__(self, n_builds: int, depth: int):
        self.n, self.lag = int(n_builds), max(0, int(depth) - 1)
        self.done = [0] * self.n                   # per build: number of leading stages issued or given up
        self.cv = threading.Condition()
        self.aborted: Optional[BaseException] = None

    def _ticket(self, build: int, stage_idx: int):
        return (build + (self.lag if STAGES[stage_idx] in LATE else 0), stage_idx, build)

    def _clear(self, ticket) -> bool:
        for b in range(self.n):                    # (a handful of builds are in flight: a linear scan is fine)
            d = self.done[b]
            if d < len(STAGES) and self._ticket(b, d) < ticket:
                return False                       # build b has not reached / passed a stage that must come first
        return True

    def issue(self, build: int, stage: str, fn: Callable):
        """Runs fn() (which issues ONE collective) when the ticket comes up."""
        si = STAGES.index(stage)
        with self.cv:
            assert self.done[build] <= si, (build, stage, self.done[build])
            self.done[build] = si                  # earlier stages of this build are given up
            self.cv.notify_all()
            me = self._ticket(build, si)
            while not self._clear(me):
                if self.aborted is not None:
                    raise RuntimeError("another build of the pipeline failed") from self.aborted
                self.cv.wait(timeout=1.0)
            try:
                return fn()                        # issued under the lock: the order of issue IS the ticket order
            finally:
                self.done[build] = si + 1
                self.cv.notify_all()

    def finish(self, build: int) -> None:
        with self.cv:
            self.done[build] = len(STAGES)
            self.cv.notify_all()

    def abort(self, error: BaseException) -> None:
        """A build failed: wake the waiting ones (they raise) instead of leaving them parked behind a ticket that never comes."""
        with self.cv:
            self.aborted = error
            self.cv.notify_all()


class OrderedGroup:
    """What a build passes as `group` when several sharded builds are in flight: the process group + its place in the order."""

    def __init__(self, order: CollectiveOrder, build: int, pg=None):
        self.order, self.build, self.pg = order, build, pg


def _pg(group):
    return group.pg if isinstance(group, OrderedGroup) else group


def _all_gather(out: torch.Tensor, inp: torch.Tensor, group, stage: str) -> None:
    if isinstance(group, OrderedGroup):
        group.order.issue(group.build, stage, lambda: dist.all_gather_into_tensor(out, inp, group=group.pg))
    else:
        dist.all_gather_into_tensor(out, inp, group=group)


def world_info(group=None) -> Tuple[int, int]:
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(_pg(group)), dist.get_world_size(_pg(group))
    return 0, 1


def block_range(n: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous balanced partition of range(n): the first n % world ranks get one extra item."""
    base, extra = divmod(n, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def all_gather_rows(local: torch.Tensor, counts: List[int], group=None, stage: str = "latents") -> torch.Tensor:
    """Concatenate per-rank row blocks of known sizes `counts` (rows may differ by rank)."""
    rank, world = world_info(group)
    if world == 1:
        return local
    width = max(counts)
    pad_shape = (width,) + tuple(local.shape[1:])
    padded = torch.zeros(pad_shape, dtype=local.dtype, device=local.device)
    padded[: local.shape[0]] = local
    out = torch.empty((world,) + pad_shape, dtype=local.dtype, device=local.device)
    _all_gather(out.view(-1), padded.view(-1), group, stage)
    return torch.cat([out[r, : counts[r]] for r in range(world)], dim=0)


def gather_latents(z_shard: torch.Tensor, n_total: int, group=None) -> torch.Tensor:
    """Row shards of the latent set -> the full corpus on every rank."""
    _, world = world_info(group)
    counts = [block_range(n_total, r, world)[1] - block_range(n_total, r, world)[0] for r in range(world)]
    return all_gather_rows(z_shard.contiguous(), counts, group, "latents")


def sharded_knn(z: torch.Tensor, n_neighbors: int, search_fn: Callable, group=None, gather_d2: bool = True):
    """search_fn(z, n_neighbors, row0, row1) -> (idx [rows, n_neighbors], d2 [rows, n_neighbors]).
    Returns the full idx on every rank and the full d2 when gather_d2 (the fp64 distances are 2/3 of the bytes and a
    connectivity graph never reads them); otherwise a callable that gathers d2 on demand (collective: every rank
    must call it or none)."""
    rank, world = world_info(group)
    n = z.shape[0]
    r0, r1 = block_range(n, rank, world)
    idx, d2 = search_fn(z, n_neighbors, r0, r1)
    if world == 1:
        return idx, d2
    counts = [block_range(n, r, world)[1] - block_range(n, r, world)[0] for r in range(world)]
    idx_all = all_gather_rows(idx, counts, group, "knn_idx")
    if gather_d2:
        return idx_all, all_gather_rows(d2, counts, group, "knn_d2")
    return idx_all, (lambda: all_gather_rows(d2, counts, group, "knn_d2"))


def chunk_range(n_edges: int, batch_size: int, rank: int, world: int) -> Tuple[int, int]:
    """Edge range [e0, e1) of this rank: whole chunks of `batch_size` edges, block-partitioned."""
    n_chunks = (n_edges + batch_size - 1) // batch_size
    c0, c1 = block_range(n_chunks, rank, world)
    return min(c0 * batch_size, n_edges), min(c1 * batch_size, n_edges)


def sharded_edge_lengths(n_edges: int, batch_size: int, length_fn: Callable, group=None) -> torch.Tensor:
    """length_fn(e0, e1) -> f32 [e1 - e0] for edges [e0, e1) (e0 is a multiple of batch_size).
    Returns all n_edges lengths on every rank."""
    rank, world = world_info(group)
    e0, e1 = chunk_range(n_edges, batch_size, rank, world)
    local = length_fn(e0, e1)
    if world == 1:
        return local
    counts = []
    for r in range(world):
        a, b = chunk_range(n_edges, batch_size, r, world)
        counts.append(b - a)
    return all_gather_rows(local, counts, group, "edge_lengths")


class RunningStatFold:
    """BatchNorm running statistics of a train-mode decoder when the JVP chunks are sharded over ranks.

    torch folds every decoder call into the buffers in order, r <- (1 - m) r + m s_t (riemannian_metric.py:57-58: two
    calls per chunk).  That recurrence is linear, so a rank that starts its block from ZERO buffers ends with
    sum_t m (1 - m)^(T_b - 1 - t) s_t over ITS calls, and the single-process result is
        (1 - m)^T r_0 + sum_b (1 - m)^(calls after block b) * fold_b .
    `start()` zeroes this rank's buffers (remembering r_0), `finish()` all-gathers the per-rank folds and writes the
    combination back, so every rank -- and a checkpoint saved by any of them -- holds the same buffers, equal to the
    single-process ones up to float32 rounding of the re-associated sum; `num_batches_tracked` advances by the calls of
    ALL ranks."""

    KEYS = ("rm1", "rv1", "rm2", "rv2")

    def __init__(self, export, group=None):
        self.export, self.group = export, group
        self.rank, self.world = world_info(group)
        self.active = self.world > 1 and export.tracks_running
        self.r0 = None

    def start(self) -> None:
        if not self.active:
            return
        self.r0 = torch.cat([self.export.tensors[k].flatten() for k in self.KEYS]).double()
        for k in self.KEYS:
            self.export.tensors[k].zero_()

    def finish(self, n_edges: int, batch_size: int) -> None:
        if not self.active:
            return
        calls = []
        for r in range(self.world):
            a, b = chunk_range(n_edges, batch_size, r, self.world)
            calls.append(2 * ((b - a + batch_size - 1) // batch_size))
        mine = torch.cat([self.export.tensors[k].flatten() for k in self.KEYS]).contiguous()
        folds = torch.empty((self.world, mine.numel()), dtype=mine.dtype, device=mine.device)
        _all_gather(folds.view(-1), mine, self.group, "bn_fold")
        keep = 1.0 - float(self.export.desc.momentum)
        total = self.r0 * keep ** sum(calls)
        for r in range(self.world):
            total = total + folds[r].double() * keep ** sum(calls[r + 1:])
        off = 0
        for k in self.KEYS:
            t = self.export.tensors[k]
            t.copy_(total[off:off + t.numel()].view_as(t).to(t.dtype))
            off += t.numel()
        self.export.commit_running_stats(sum(calls) - calls[self.rank])      # copies back, counts the other ranks' calls


def merge_min_argmin(dmin: torch.Tensor, arg: torch.Tensor, offsets: List[int], group=None):
    """Per-rank (column minimum, first local row attaining it) over disjoint source blocks -> the global
    (minimum, first row) with numpy's argmin tie rule: lowest global row index among equal minima;
    an all-inf column keeps row 0."""
    rank, world = world_info(group)
    if world == 1:
        return dmin, arg
    n = dmin.shape[0]
    all_d = torch.empty((world, n), dtype=dmin.dtype, device=dmin.device)
    all_a = torch.empty((world, n), dtype=arg.dtype, device=arg.device)
    _all_gather(all_d.view(-1), dmin.contiguous(), group, "assign_d")
    _all_gather(all_a.view(-1), (arg + offsets[rank]).contiguous(), group, "assign_a")
    best_d, best_a = all_d[0].clone(), all_a[0].clone()
    for r in range(1, world):                      # ranks own increasing source blocks: strict < keeps the lowest index
        better = all_d[r] < best_d
        best_d = torch.where(better, all_d[r], best_d)
        best_a = torch.where(better, all_a[r], best_a)
    return best_d, best_a


def sharded_assign(n_sources: int, solve_fn: Callable, group=None):
    """solve_fn(s0, s1) -> (dmin f32 [n], argmin i32 [n] local to [s0, s1)) for the source block [s0, s1)
    (an empty block must return (+inf, 0)).  Returns the merged (dmin, argmin) on every rank."""
    rank, world = world_info(group)
    s0, s1 = block_range(n_sources, rank, world)
    dmin, arg = solve_fn(s0, s1)
    offsets = [block_range(n_sources, r, world)[0] for r in range(world)]
    return merge_min_argmin(dmin, arg, offsets, group)

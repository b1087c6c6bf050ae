// kpp.hip -- the k-means++ seeding chain of the reference, resident on the device (gfx950).
//
// Replaces the loop of src/geo/kmeans_optimized.py:40-71 (kpp_initialization_graph): per centre one
// single-source solve (scipy Dijkstra there, csrc/sssp_device.h here), d_min = minimum(d_min, d),
// float32 D^2 weights with the inf -> 2*max_finite rule, and numpy's legacy RandomState.choice.
//
// The draw is reproduced bit for bit on the GPU, so the chain needs no host round trip per centre:
//   * probs = d_safe**2, probs[centres] = 0                    -- float32, elementwise;
//   * total = probs.sum()                                      -- numpy's float32 add.reduce: pairwise
//       sums (8 strided accumulators on <=128-element leaves, halving tree above) of 8192-element
//       buffer chunks, accumulated chunk after chunk.  The tree is rebuilt here from a host-made
//       plan that depends only on N, every add in float32 in numpy's order;
//   * p = probs / total (float32, correctly rounded division), cdf = cumsum(float64(p)),
//     cdf /= cdf[-1], idx = searchsorted(cdf, u, 'right')      -- RandomState.choice.
//       The fp64 cumsum is a parallel scan; it differs from numpy's sequential one by at most
//       ~N ulp.  The pick is accepted only if u clears both neighbouring cdf values by a margin
//       far above that bound; otherwise (probability ~1e-6 per draw) the chain stops at that
//       iteration and the caller repeats the draw on the host with numpy itself.
// The uniform deviates u[t] are drawn by the host from the same RandomState stream and passed in.
//
// Every kernel first checks the control block's abort flag, so after an abort (solve not converged
// within the enqueued sweeps, margin failure, degenerate weights) the device state stays exactly
// as it was when the failing iteration started its draw / before it absorbed its solve.
#include "geo_common.h"
#include "sssp_device.h"

#include <cstring>
#include <vector>

namespace {

constexpr int NP_BUFSIZE = 8192;      // numpy's default ufunc buffer size (elements per reduction chunk)
constexpr int PW_BLOCK = 128;         // numpy pairwise-sum leaf size
constexpr int SCAN_T = 256, SCAN_I = 8, SCAN_TILE = SCAN_T * SCAN_I;
constexpr int FINISH_GRID = 256;

// Words that take atomics (frontier counts, tickets) each sit on a 128-byte line of their own, away from the words every
// block of every launch reads (abort flag, total, margin): an atomic is executed at the memory side and drops the line
// from the L2s, so readers of a shared line would all go to memory behind the atomics.
struct KppCtl {
    int32_t abort_iter, abort_reason;      // -1 / 0 while healthy; reason 1 solve, 2 margin, 3 degenerate, 4 cell too large
    float total;
    float maxf;                            // max finite d_min (-1: none), basis of the pruning margin
    int32_t n_inf;                         // unreachable (inf) entries of d_min at the last max pass
    int32_t max_sw;                        // most sweeps any solve of this call needed (non-empty frontiers)
    int32_t pad0[26];
    int32_t fcount[3][32];                 // frontier sizes, ring over sweeps: [k][0]
    int32_t ticket[2][32];                 // "last block finishes the job" counters of the sum / draw kernels: [k][0]
};

__device__ __forceinline__ double inf64() { return __longlong_as_double(0x7ff0000000000000LL); }
__device__ __forceinline__ float inf32() { return __int_as_float(0x7f800000); }

// ---- pruned frontier solve -------------------------------------------------------------------
// Only nodes the new centre can still improve matter for d_min / argmin: a node whose tentative
// distance from the new centre exceeds its current d_min by more than tau = 1e-6 * max_finite(d_min)
// is not expanded.  Every path through such a node reaches its successors later than their current
// d_min (triangle inequality through that node's own centre; tau is ~8x the worst f32/fp64
// rounding of the quantities involved, all of which are < 2*max_finite), so no update is lost, and
// nodes that ARE updated have an all-unpruned optimal path, hence their exact fixed-point distance.
// Work per solve drops from O(nnz) per sweep to the size of the new centre's cell.
// One stamp per node says "already queued for the sweep after sweep sw of solve s": s * 4096 + sw + 1.
__device__ __forceinline__ void kpp_begin(KppCtl *ctl, const int32_t *centers, int32_t pos, double *d, int32_t *front0) {
    const int32_t src = centers[pos];
    d[src] = 0.0;
    front0[0] = src;
    ctl->fcount[0][0] = 1;
    ctl->fcount[1][0] = 0;
    ctl->fcount[2][0] = 0;
}

__global__ void kpp_begin_kernel(KppCtl *ctl, const int32_t *__restrict__ centers, int32_t pos, double *d,
                                 int32_t *front0) {
    if (ctl->abort_iter >= 0 || threadIdx.x != 0 || blockIdx.x != 0) return;
    kpp_begin(ctl, centers, pos, d, front0);
}

// One sweep: every queued node u (32 lanes share its adjacency row; `d[u]` was written before this launch)
// lowers its neighbours with a 64-bit atomicMin on the bit pattern; lowered, unpruned neighbours are queued
// once for the next sweep.  The queue tail is advanced once per wave (ballot), not once per node.
template <bool WEIGHTED>
__device__ __forceinline__ void kpp_push_body(KppCtl *ctl, const int32_t *__restrict__ indptr,
                                              const int32_t *__restrict__ indices, const float *__restrict__ weights,
                                              double *d, const float *__restrict__ dmin, int32_t *mark,
                                              const int32_t *__restrict__ fin, int32_t *__restrict__ fout, int32_t cnt,
                                              int next, int32_t stamp, bool pretest) {
    const double tau = ctl->maxf > 0.f ? 1e-6 * (double)ctl->maxf : 0.0;
    const int sub = threadIdx.x & 31, lane = threadIdx.x & 63;       // 32 lanes per frontier node (mean degree ~31)
    const int grp = (blockIdx.x * blockDim.x + threadIdx.x) >> 5;
    const int ngrp = (gridDim.x * blockDim.x) >> 5;
    unsigned long long *dbits = reinterpret_cast<unsigned long long *>(d);
    for (int32_t i = grp; i < cnt; i += ngrp) {
        const int32_t u = fin[i];
        const double du = d[u];
        if (du > (double)dmin[u] + tau) continue;           // pruned: cannot improve anything behind it
        const int32_t e0 = indptr[u], e1 = indptr[u + 1];
        for (int32_t e = e0 + sub; e < e1; e += 32) {
            const int32_t v = indices[e];
            const double cand = du + (WEIGHTED ? (double)weights[e] : 1.0);
            const unsigned long long cb = (unsigned long long)__double_as_longlong(cand);
            bool queue = false;
            // large frontiers: a plain read first (monotone: values only decrease) spares most atomics;
            // small ones: the atomic straight away spares a dependent round trip
            bool go = true;
            if (pretest) go = cb < dbits[v];
            if (go) {
                const unsigned long long old = atomicMin(&dbits[v], cb);
                if (cb < old && cand <= (double)dmin[v] + tau) queue = atomicMax(&mark[v], stamp) < stamp;
            }
            const unsigned long long m = __ballot(queue);
            if (m) {
                const int leader = __ffsll((long long)m) - 1;
                int32_t at = 0;
                if (lane == leader) at = atomicAdd(&ctl->fcount[next][0], __popcll(m));
                at = __shfl(at, leader, 64);
                if (queue) fout[at + __popcll(m & ((1ull << lane) - 1ull))] = v;
            }
        }
    }
}

template <bool WEIGHTED>
__global__ __launch_bounds__(256) void kpp_push_kernel(KppCtl *ctl, const int32_t *__restrict__ indptr,
                                                      const int32_t *__restrict__ indices,
                                                      const float *__restrict__ weights, double *d,
                                                      const float *__restrict__ dmin, int32_t *mark,
                                                      const int32_t *__restrict__ fin, int32_t *__restrict__ fout,
                                                      int cur, int next, int clear, int32_t stamp_solve, int32_t sw) {
    if (ctl->abort_iter >= 0) return;
    if (blockIdx.x == 0 && threadIdx.x == 0) ctl->fcount[clear][0] = 0;
    const int32_t cnt = ctl->fcount[cur][0];
    if (cnt == 0) return;
    if (blockIdx.x == 0 && threadIdx.x == 0 && sw + 1 > ctl->max_sw) ctl->max_sw = sw + 1;   // launches are serial
    kpp_push_body<WEIGHTED>(ctl, indptr, indices, weights, d, dmin, mark, fin, fout, cnt, next,
                            stamp_solve * 4096 + sw + 1, true);
}

// d_min / argmin update (kmeans_optimized.py:44 + the single-pass assignment) outside the fused path below;
// resets the solve's distances.  Aborts (nothing applied) when the frontier is not empty.
__global__ __launch_bounds__(256) void kpp_finish_kernel(KppCtl *ctl, double *d, float *__restrict__ dmin,
                                                        int32_t *__restrict__ argmin, int32_t n, int last_next,
                                                        int32_t pos) {
    if (ctl->abort_iter >= 0) return;
    if (ctl->fcount[last_next][0] != 0) {
        if (blockIdx.x == 0 && threadIdx.x == 0) { ctl->abort_iter = pos; ctl->abort_reason = 1; }
        return;                                             // every block sees the same final count
    }
    for (int32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const double dd = d[i];
        if (dd < inf64()) {
            const float x = (float)dd;
            if (x < dmin[i]) { dmin[i] = x; argmin[i] = pos; }
            d[i] = inf64();
        }
    }
}

// per-block maxima of the finite d_min for the inf -> 2*max_finite rule (:47-50) and the pruning margin,
// and the number of unreachable (inf) entries
__global__ __launch_bounds__(256) void kpp_max_kernel(const KppCtl *ctl, const float *__restrict__ dmin, int32_t n,
                                                     float *__restrict__ part_max, int32_t *__restrict__ part_inf) {
    if (ctl->abort_iter >= 0) return;
    __shared__ float smax[4];
    __shared__ int32_t sinf[4];
    float m = -1.0f;                                   // distances are >= 0: -1 means "no finite value seen"
    int32_t ninf = 0;
    for (int32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const float cur = dmin[i];
        if (cur < inf32()) m = fmaxf(m, cur); else ++ninf;
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        m = fmaxf(m, __shfl_xor(m, off, 64));
        ninf += __shfl_xor(ninf, off, 64);
    }
    if ((threadIdx.x & 63) == 0) { smax[threadIdx.x >> 6] = m; sinf[threadIdx.x >> 6] = ninf; }
    __syncthreads();
    if (threadIdx.x == 0) {
        part_max[blockIdx.x] = fmaxf(fmaxf(smax[0], smax[1]), fmaxf(smax[2], smax[3]));
        part_inf[blockIdx.x] = sinf[0] + sinf[1] + sinf[2] + sinf[3];
    }
}

__global__ void kpp_maxfin_kernel(KppCtl *ctl, const float *__restrict__ part_max,
                                  const int32_t *__restrict__ part_inf, int n_part) {
    if (ctl->abort_iter >= 0) return;
    float m = -1.0f;
    int32_t ninf = 0;
    for (int i = threadIdx.x; i < n_part; i += 64) { m = fmaxf(m, part_max[i]); ninf += part_inf[i]; }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        m = fmaxf(m, __shfl_xor(m, off, 64));
        ninf += __shfl_xor(ninf, off, 64);
    }
    if (threadIdx.x == 0) { ctl->maxf = m; ctl->n_inf = ninf; }
}

__global__ __launch_bounds__(256) void kpp_fill_inf_kernel(double *__restrict__ d, int32_t n) {
    for (int32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) d[i] = inf64();
}

// "Last block finishes the job": values other blocks must see inside the same launch are written and read with
// agent-scope accesses (st_dev / ld_dev: they go to the device-coherent level, no L2 write-back or invalidate is
// needed), every block waits for its own stores and takes a ticket, the block drawing the last ticket continues.
template <typename T>
__device__ __forceinline__ void st_dev(T *p, T v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
template <typename T>
__device__ __forceinline__ T ld_dev(const T *p) {
    return __hip_atomic_load(const_cast<T *>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ bool last_block_done(int32_t *ticket, int32_t participants) {
    __shared__ int32_t s_last;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");      // this thread's stores have been acknowledged
    __syncthreads();
    if (threadIdx.x == 0)
        s_last = (__hip_atomic_fetch_add(ticket, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == participants - 1) ? 1 : 0;
    __syncthreads();
    return s_last != 0;
}

// probs = d_safe**2 with probs[centres] = 0 (kmeans_optimized.py:47-57), written for the cdf pass, and numpy's
// float32 add.reduce over it (see file header).
//   phase 1: each block owns SUM_LEAVES consecutive leaves (<= 4096 elements); all threads compute their
//            probabilities (coalesced) into LDS and global memory.  `fuse_finish`: the d_min / argmin update of
//            the solve just finished (and the reset of its distances) happens here, element by element;
//   phase 2: leaf sums -- 8 lanes own the 8 strided accumulators r[0..7] of one <=128-element leaf, combined as
//            ((r0+r1)+(r2+r3))+((r4+r5)+(r6+r7));
//   phase 3 (last block only): the halving tree above the leaves, then the chunk roots accumulated in order.
// `exact_max`: the per-block maxima of kpp_max_kernel are reduced here (needed when d_min still holds inf
// entries, which are replaced by 2*max_finite); otherwise every entry is finite and no maximum is needed.
constexpr int SUM_LEAVES = 16, TREE_LDS_NODES = 1024, SP_FLOATS = 4096;   // (16 leaves per block: 489 blocks at one million latents)
struct SumPlan {                          // numpy's reduction tree over n float32 (host-built, see pw_build)
    const int32_t *leaf_start, *leaf_len, *node_l, *node_r, *level_off, *chunk_root;
    int n_leaves, n_levels, n_chunks, n_nodes;
    float *val;
    // large n: the tree of ONE numpy buffer (<= 8 192 elements, <= 128 leaves) in local ids -- [0] a full buffer, [1] the last
    // (partial) one, CPLAN_INTS ints each: {leaves, nodes, levels, root, level_off[12], node_l[128], node_r[128]} -- a ticket per
    // buffer and the buffers' sums
    const int32_t *cplan;
    int32_t *chunk_ticket;
    float *chunk_val;
};
constexpr int CPLAN_MAX_LEAVES = 128, CPLAN_INTS = 16 + 2 * CPLAN_MAX_LEAVES;

// Block `bid` of `nblocks`.  Returns true (for all its threads) in the block that finished the tree; its thread 0
// then holds the total in *total_out.
__device__ __forceinline__ bool kpp_sum_body(KppCtl *ctl, float *__restrict__ dmin, int32_t *__restrict__ argmin,
                                             double *__restrict__ d, int fuse_finish, int32_t pos,
                                             const uint8_t *__restrict__ is_center, const float *__restrict__ part_max,
                                             const int32_t *__restrict__ part_inf, int n_part, int exact_max,
                                             float *__restrict__ probs, const SumPlan &pl, int bid, int nblocks,
                                             float *total_out, double *__restrict__ leaf_a = nullptr) {
    __shared__ float sp[SP_FLOATS];
    __shared__ float smax;
    __shared__ int32_t sinf;
    const int n_leaves = pl.n_leaves;
    float *val = pl.val;
    float maxf = 0.0f;
    bool any_finite = true;
    if (exact_max) {
        if (threadIdx.x < 64) {
            float m = -1.0f;
            int32_t ninf = 0;
            for (int i = threadIdx.x; i < n_part; i += 64) { m = fmaxf(m, part_max[i]); ninf += part_inf[i]; }
#pragma unroll
            for (int off = 32; off >= 1; off >>= 1) {
                m = fmaxf(m, __shfl_xor(m, off, 64));
                ninf += __shfl_xor(ninf, off, 64);
            }
            if (threadIdx.x == 0) { smax = m; sinf = ninf; }
        }
        __syncthreads();
        maxf = smax;
        any_finite = maxf >= 0.0f;
        if (bid == 0 && threadIdx.x == 0) { ctl->maxf = maxf; ctl->n_inf = sinf; }   // margin of the next solve
    }
    const float sub = maxf * 2.0f;
    const int l0 = bid * SUM_LEAVES;
    const int l1 = l0 + SUM_LEAVES < n_leaves ? l0 + SUM_LEAVES : n_leaves;
    const int32_t b0 = pl.leaf_start[l0], b1 = pl.leaf_start[l1 - 1] + pl.leaf_len[l1 - 1];
    // (all of a thread's loads first: 16 dependent iterations of load -> store cost a memory latency each, 20-30 us per centre at
    // one million latents with 245 blocks on the chip)
    {
        constexpr int PER = SUM_LEAVES * PW_BLOCK / 256;
        float xs[PER];
        double dds[PER];
        uint8_t ics[PER];
#pragma unroll
        for (int k = 0; k < PER; ++k) {
            const int32_t i = b0 + threadIdx.x + k * 256;
            xs[k] = 0.0f; dds[k] = inf64(); ics[k] = 0;
            if (i < b1) {
                xs[k] = dmin[i];
                if (fuse_finish) dds[k] = d[i];
                ics[k] = is_center[i];
            }
        }
#pragma unroll
        for (int k = 0; k < PER; ++k) {
            const int32_t i = b0 + threadIdx.x + k * 256;
            if (i < b1) {
                float x = xs[k];
                if (fuse_finish && dds[k] < inf64()) {             // reached by the solve of centre `pos`
                    const float xd = (float)dds[k];
                    // (agent-scope stores: the launch's last block reads d_min of the picked leaf and opens the next solve
                    // with a store to d[] -- a plain store could still sit in another XCD's L2 then; only touched nodes pay)
                    if (xd < x) { x = xd; st_dev(&dmin[i], xd); argmin[i] = pos; }
                    st_dev(&d[i], inf64());
                }
                const float safe = any_finite ? (x < inf32() ? x : sub) : 1.0f;
                const float p = ics[k] ? 0.0f : safe * safe;
                probs[i] = p;
                sp[i - b0] = p;
            }
        }
    }
    __syncthreads();
    {
        const int j = threadIdx.x & 7;
        const int leaf = l0 + (threadIdx.x >> 3);
        const bool live = leaf < l1;
        const float *a = sp + (live ? pl.leaf_start[leaf] - b0 : 0);
        const int len = live ? pl.leaf_len[leaf] : 0;
        const int m8 = len - (len % 8);
        float r = 0.0f;
        double acc = 0.0;                                  // fp64 sum of the same weights: the approximate cdf of kpp_approx_draw
        if (len >= 8) {
            r = a[j];
            acc = (double)a[j];
            for (int i = 8; i < m8; i += 8) { r += a[i + j]; acc += (double)a[i + j]; }
        }
        // lanes (0,1) (2,3) (4,5) (6,7) -> lanes 0,2,4,6 ; then (0,2) (4,6) -> 0,4 ; then (0,4) -> 0
        float o = __shfl_down(r, 1, 8);
        if ((j & 1) == 0) r += o;
        o = __shfl_down(r, 2, 8);
        if ((j & 3) == 0) r += o;
        o = __shfl_down(r, 4, 8);
        acc += __shfl_down(acc, 1, 8);
        acc += __shfl_down(acc, 2, 8);
        acc += __shfl_down(acc, 4, 8);
        if (j == 0) {
            r += o;
            if (len < 8) {
                r = 0.0f;
                acc = 0.0;
                for (int i = 0; i < len; ++i) { r += a[i]; acc += (double)a[i]; }
            } else {
                for (int i = m8; i < len; ++i) { r += a[i]; acc += (double)a[i]; }
            }
            if (live) st_dev(&val[leaf], r);
            if (live && leaf_a) st_dev(&leaf_a[leaf], acc);  // (read by this launch's last block)
        }
    }
    __shared__ int32_t s_nl[TREE_LDS_NODES], s_nr[TREE_LDS_NODES], s_lo[32], s_cr[64];
    const int n_levels = pl.n_levels, n_chunks = pl.n_chunks;
    const int n_nodes = pl.n_nodes;
    float total = 0.0f;
    if (n_nodes <= TREE_LDS_NODES && n_leaves + n_nodes <= SP_FLOATS && n_levels < 32 && n_chunks <= 64) {
        if (!last_block_done(&ctl->ticket[0][0], nblocks)) return false;
        // small tree: one round trip brings leaves and plan into LDS, the levels then cost LDS latency only
        for (int j = threadIdx.x; j < n_leaves; j += 256) sp[j] = ld_dev(&val[j]);
        for (int j = threadIdx.x; j < n_nodes; j += 256) { s_nl[j] = pl.node_l[j]; s_nr[j] = pl.node_r[j]; }
        if (threadIdx.x <= n_levels) s_lo[threadIdx.x] = pl.level_off[threadIdx.x];
        if (threadIdx.x < n_chunks) s_cr[threadIdx.x] = pl.chunk_root[threadIdx.x];
        __syncthreads();
        for (int lv = 0; lv < n_levels; ++lv) {
            for (int j = s_lo[lv] + threadIdx.x; j < s_lo[lv + 1]; j += 256) sp[n_leaves + j] = sp[s_nl[j]] + sp[s_nr[j]];
            __syncthreads();
        }
        if (threadIdx.x == 0)
            for (int c = 0; c < n_chunks; ++c) total += sp[s_cr[c]];
    } else {
        // Large n: the tree is a forest -- one halving tree per numpy buffer of 8 192 elements, whose sums are then added in order.
        // The LAST of the (four) blocks of a buffer reduces that buffer's <= 128 leaf sums in LDS (one device-scope round trip for
        // the leaves, the levels at LDS latency) and publishes the buffer's sum; the last buffer to finish adds the sums in order.
        // Round 4 walked all levels of all buffers in the launch's last block through global memory: six levels x (index loads ->
        // dependent device-scope value loads -> stores -> barrier), ~18 of the ~33 us this launch took per centre at one million
        // latents (kernel trace of the C4 chain).
        constexpr int LPC = NP_BUFSIZE / PW_BLOCK, BPC = LPC / SUM_LEAVES;      // leaves / blocks of a full buffer
        const int chunk = l0 / LPC < n_chunks - 1 ? l0 / LPC : n_chunks - 1;
        const int participants = chunk < n_chunks - 1 ? BPC : nblocks - (n_chunks - 1) * BPC;
        if (!last_block_done(&pl.chunk_ticket[chunk], participants)) return false;
        const int32_t *cp = pl.cplan + (chunk == n_chunks - 1 ? CPLAN_INTS : 0);
        const int cl = cp[0], cn = cp[1], clv = cp[2], croot = cp[3];
        const int first_leaf = chunk * LPC;
        for (int j = threadIdx.x; j < cl; j += 256) sp[j] = ld_dev(&val[first_leaf + j]);
        for (int j = threadIdx.x; j < cn; j += 256) { s_nl[j] = cp[16 + j]; s_nr[j] = cp[16 + CPLAN_MAX_LEAVES + j]; }
        if ((int)threadIdx.x <= clv) s_lo[threadIdx.x] = cp[4 + threadIdx.x];
        __syncthreads();
        for (int lv = 0; lv < clv; ++lv) {
            for (int j = s_lo[lv] + threadIdx.x; j < s_lo[lv + 1]; j += 256) sp[cl + j] = sp[s_nl[j]] + sp[s_nr[j]];
            __syncthreads();
        }
        if (threadIdx.x == 0) {
            st_dev(&pl.chunk_val[chunk], sp[croot]);
            st_dev(&pl.chunk_ticket[chunk], 0);                 // (every block of the buffer has drawn its ticket)
        }
        if (!last_block_done(&ctl->ticket[0][0], n_chunks)) return false;
        // numpy adds its buffers one after the other: fetched by all threads, added in order by one
        if (n_chunks <= SP_FLOATS) {
            for (int c = threadIdx.x; c < n_chunks; c += blockDim.x) sp[c] = ld_dev(&pl.chunk_val[c]);
            __syncthreads();
            if (threadIdx.x == 0)
                for (int c = 0; c < n_chunks; ++c) total += sp[c];
        } else if (threadIdx.x == 0) {
            for (int c = 0; c < n_chunks; ++c) total += ld_dev(&pl.chunk_val[c]);
        }
    }
    *total_out = total;
    return true;
}

__global__ __launch_bounds__(256) void kpp_sum_kernel(KppCtl *ctl, float *__restrict__ dmin,
                                                     int32_t *__restrict__ argmin, double *__restrict__ d,
                                                     int fuse_finish, int last_next, int32_t pos,
                                                     const uint8_t *__restrict__ is_center,
                                                     const float *__restrict__ part_max,
                                                     const int32_t *__restrict__ part_inf, int n_part, int exact_max,
                                                     float *__restrict__ probs, SumPlan pl) {
    if (ctl->abort_iter >= 0) return;
    if (fuse_finish && ctl->fcount[last_next][0] != 0) {          // the solve did not converge: apply nothing
        if (blockIdx.x == 0 && threadIdx.x == 0) { ctl->abort_iter = pos; ctl->abort_reason = 1; }
        return;
    }
    float total = 0.0f;
    if (!kpp_sum_body(ctl, dmin, argmin, d, fuse_finish, pos, is_center, part_max, part_inf, n_part, exact_max, probs,
                      pl, blockIdx.x, gridDim.x, &total))
        return;
    if (threadIdx.x == 0) {
        ctl->total = total;
        ctl->ticket[0][0] = 0;
        if (!(total > 0.0f)) { ctl->abort_iter = pos; ctl->abort_reason = 3; }
    }
}

// The draw.  Every block: tile-local inclusive fp64 scan of p = float64(probs / total) and the tile total.
// Last block: exclusive tile offsets, idx = searchsorted(cdf / cdf[-1], u, side='right') with a safety margin
// around u -- first the tile whose first value is the last one <= u, then the position inside it.  Returns true in
// that block, with the pick in pick[0..2] = {found, index, margin ok} (shared memory).
// One tile's inclusive fp64 scan of p = float64(probs / total): thread t owns SCAN_I consecutive elements (sequential sums), wave
// scan, wave bases.  v[i] + *excl is the tile-local inclusive value of the thread's i-th element; returns the tile total.  The same
// code produces the tile totals (every block) and, in the last block, the values inside the picked tile: identical arithmetic.
__device__ __forceinline__ double kpp_tile_scan(const float *__restrict__ probs, float total, int32_t n, int tile, double (&v)[SCAN_I],
                                                double *excl, double *wtot) {
    const int64_t base = (int64_t)tile * SCAN_TILE + (int64_t)threadIdx.x * SCAN_I;
    double run = 0.0;
#pragma unroll
    for (int i = 0; i < SCAN_I; ++i) {
        const double p = (base + i < n) ? (double)(probs[base + i] / total) : 0.0;
        run += p;
        v[i] = run;
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    double inc = run;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const double t = __shfl_up(inc, off, 64);
        if (lane >= off) inc += t;
    }
    if (lane == 63) wtot[wave] = inc;
    __syncthreads();
    double wbase = 0.0, tot = 0.0;
#pragma unroll
    for (int w = 0; w < SCAN_T / 64; ++w) {
        if (w < wave) wbase += wtot[w];
        tot += wtot[w];
    }
    *excl = wbase + inc - run;
    return tot;
}

// The draw.  Every block: its tile's total (the scan above).  Last block: exclusive tile offsets, idx = searchsorted(cdf / cdf[-1],
// u, side='right') with a safety margin around u -- first the tile whose first value is the last one <= u, then the position inside
// it, from a second scan of THAT tile (round 3 wrote all n cdf values with device-scope stores in every draw and read a few of them
// back: 8 MB per centre at one million latents).  Returns true in that block, with the pick in pick[0..2] = {found, index, margin ok}.
__device__ __forceinline__ bool kpp_draw_body(KppCtl *ctl, float total, const float *__restrict__ probs, int32_t n,
                                              double *cdf, double *tile_sum, int n_tiles, double u, double tol,
                                              int bid, int nblocks, int32_t *pick) {
    extern __shared__ __attribute__((aligned(16))) double toff[];          // [n_tiles + 1]
    __shared__ double wtot[SCAN_T / 64];
    __shared__ double ctile[SCAN_TILE + 1];                                // cdf / cdf[-1] inside the picked tile (+ the next value)
    __shared__ int32_t s_tile;
    (void)cdf;
    {
        double v[SCAN_I], excl;
        const double tot = kpp_tile_scan(probs, total, n, bid, v, &excl, wtot);
        if (threadIdx.x == 0) st_dev(&tile_sum[bid], tot);
    }
    if (!last_block_done(&ctl->ticket[1][0], nblocks)) return false;
    // exclusive tile offsets: all tile totals fetched first (one round trip), then a wave scan, 64 tiles a pass, out of LDS
    for (int t = threadIdx.x; t < n_tiles; t += SCAN_T) toff[t] = ld_dev(&tile_sum[t]);
    __syncthreads();
    if (threadIdx.x < 64) {
        const int lane = threadIdx.x;
        double carry = 0.0;
        for (int t0 = 0; t0 < n_tiles; t0 += 64) {
            const int t = t0 + lane;
            const double x = t < n_tiles ? toff[t] : 0.0;
            double inc = x;
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) {
                const double o = __shfl_up(inc, off, 64);
                if (lane >= off) inc += o;
            }
            if (t < n_tiles) toff[t] = carry + (inc - x);
            carry += __shfl(inc, 63, 64);
        }
        if (lane == 0) { toff[n_tiles] = carry; s_tile = -1; pick[0] = 0; pick[1] = -1; pick[2] = 0; }
    }
    __syncthreads();
    const double s_last = toff[n_tiles];
    // the first value of tile t is its first element's p (tile-local inclusive scan) + the tile's offset
    auto first_of = [&](int t) { return ((double)(probs[(int64_t)t * SCAN_TILE] / total) + toff[t]) / s_last; };
    for (int t = threadIdx.x; t < n_tiles; t += SCAN_T)
        if (first_of(t) <= u) atomicMax(&s_tile, t);
    __syncthreads();
    const int tile = s_tile;
    if (tile < 0) {                                       // nothing <= u: index 0
        if (threadIdx.x == 0) { pick[0] = 1; pick[1] = 0; pick[2] = (first_of(0) - u > tol) ? 1 : 0; }
    } else {
        double v[SCAN_I], excl;
        (void)kpp_tile_scan(probs, total, n, tile, v, &excl, wtot);
#pragma unroll
        for (int i = 0; i < SCAN_I; ++i) ctile[threadIdx.x * SCAN_I + i] = (excl + v[i] + toff[tile]) / s_last;
        if (threadIdx.x == 0) ctile[SCAN_TILE] = tile + 1 < n_tiles ? first_of(tile + 1) : inf64();
        __syncthreads();
#pragma unroll
        for (int k = 0; k < SCAN_I; ++k) {
            const int q = k * SCAN_T + threadIdx.x;
            const int32_t j = tile * SCAN_TILE + q;
            const bool in = j < n, last = j >= n - 1;
            const double cj = in ? ctile[q] : inf64();
            const double cn = last ? inf64() : ctile[q + 1];
            if (in && cj <= u && cn > u) {
                pick[0] = 1; pick[1] = j + 1;
                pick[2] = (!last && (u - cj > tol) && (cn - u > tol)) ? 1 : 0;
            }
        }
    }
    __syncthreads();
    return true;
}

// The draw on the UN-NORMALISED weights, by one workgroup (the resident chain's method, section "k-means++ chain" of DESIGN.md):
// cdf[j] / cdf[-1] of RandomState.choice differs from sum_{i<=j} w_i / sum_i w_i by less than 2 * 2^-24 + n * 2^-52, so if u clears
// both neighbouring steps of that approximate cdf by KPP_STEP_APPROX_MARGIN (+ tol) the index is numpy's and neither the float32
// total nor any p_i = fl32(w_i / total) is needed: a scan over the n / 128 fp64 leaf sums of the reduction launch and ONE leaf,
// instead of a launch over all n weights.  pick = {index, accepted}.  Declined (1-2 % of the draws) -> the exact draw runs next.
constexpr double KPP_STEP_APPROX_MARGIN = 1.25e-7;        // > 2 * 2^-24 + n * 2^-52 up to n = 2^24
// Runs in the LAST block of the reduction launch itself (round 4 spent a launch of its own on it, ~12 us per centre): the leaf sums
// come through agent-scope loads, and the weights of the picked leaf are formed again from d_min (agent-scope loads: the entries
// this launch improved were stored at that scope) -- `probs` of that leaf may still sit in another XCD's L2.
__device__ __forceinline__ void kpp_approx_draw(const double *leaf_a, const float *dmin, const uint8_t *__restrict__ is_center,
                                                const SumPlan &pl, int32_t n, double u, double margin_rel, int32_t *pick) {
    __shared__ double w_tot[4], t_base[256];
    __shared__ int32_t s_thread, s_leaf;
    __shared__ double s_leaf_e, s_total;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n_leaves = pl.n_leaves;
    const int per = (n_leaves + 255) / 256;
    const int l0 = tid * per, l1 = l0 + per < n_leaves ? l0 + per : n_leaves;
    double mine = 0.0;
    for (int l = l0; l < l1; ++l) mine += ld_dev(&leaf_a[l]);
    double incl = mine;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const double o = __shfl_up(incl, off, 64);
        if (lane >= off) incl += o;
    }
    if (lane == 63) w_tot[wave] = incl;
    if (tid == 0) { s_thread = 0; pick[0] = -1; pick[1] = 0; }
    __syncthreads();
    double base = 0.0;
    for (int w = 0; w < wave; ++w) base += w_tot[w];
    t_base[tid] = base + (incl - mine);
    if (tid == 255) s_total = base + incl;
    __syncthreads();
    const double S = s_total, target = u * S;
    if (l0 < n_leaves && t_base[tid] <= target) atomicMax(&s_thread, tid);     // t_base[0] = 0 <= target
    __syncthreads();
    if (wave == 0) {
        // the leaf inside the picked thread's segment: the segment's leaf sums fetched by the lanes together (round 4 walked them
        // in one thread behind a data-dependent exit, a cache round trip per leaf), scanned, and counted while they stay <= target
        // -- the last leaf of the segment is never stepped over (the next segment's base is > target)
        const int st = s_thread;
        const int sl0 = st * per, sl1 = sl0 + per < n_leaves ? sl0 + per : n_leaves;
        double carry = t_base[st];
        int L = sl0;
        double e = carry;
        for (int p0 = sl0; p0 < sl1; p0 += 64) {
            const int l = p0 + lane;
            const double x = l < sl1 ? ld_dev(&leaf_a[l]) : 0.0;
            double inc = x;
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) {
                const double o = __shfl_up(inc, off, 64);
                if (lane >= off) inc += o;
            }
            const double incl = carry + inc;
            const bool ok = l + 1 < sl1 && incl <= target;
            const int cnt = __popcll(__ballot(ok));
            if (cnt > 0) { e = __shfl(incl, cnt - 1, 64); L = p0 + cnt; }
            if (cnt < 64) break;                                  // (uniform: cnt comes from a ballot)
            carry = __shfl(incl, 63, 64);
        }
        if (lane == 0) { s_leaf = L; s_leaf_e = e; }
    }
    __syncthreads();
    if (wave == 0) {
        const int L = s_leaf;
        const double leaf_e = s_leaf_e;
        const int32_t b0 = pl.leaf_start[L];
        const int len = pl.leaf_len[L];
        const int i0 = 2 * lane, i1 = 2 * lane + 1;
        auto weight = [&](int32_t i) -> double {               // kpp_sum_body's p of entry i (every d_min finite in this mode)
            const float x = ld_dev(&dmin[i]);
            return is_center[i] ? 0.0 : (double)(x * x);
        };
        const double q0 = i0 < len ? weight(b0 + i0) : 0.0;
        const double q1 = i1 < len ? weight(b0 + i1) : 0.0;
        double inc = q0 + q1;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const double o = __shfl_up(inc, off, 64);
            if (lane >= off) inc += o;
        }
        const double c1 = leaf_e + inc, c0 = leaf_e + (inc - q1);
        const bool le0 = i0 < len && c0 <= target, le1 = i1 < len && c1 <= target;
        const int count = __popcll(__ballot(le0)) + __popcll(__ballot(le1));
        // c is non-decreasing: the largest c <= target is element count-1 (or the previous leaf's end), the smallest c > target
        // is element count
        const int lo_i = count - 1, hi_i = count;
        const double lo_v = __shfl((lo_i & 1) ? c1 : c0, (lo_i >> 1) & 63, 64);
        const double hi_v = __shfl((hi_i & 1) ? c1 : c0, (hi_i >> 1) & 63, 64);
        if (lane == 0) {
            const double lower = count > 0 ? lo_v : leaf_e;
            const int32_t idx = b0 + count;
            const double margin = margin_rel * S;
            pick[0] = idx;
            pick[1] = (count < len && idx < n && S > 0.0 && S < 1e37 && (target - lower > margin) && (hi_v - target > margin)) ? 1 : 0;
        }
    }
    __syncthreads();
}

// scan + pick + commit: accept or decline the pick; on success also open the next solve
__global__ __launch_bounds__(SCAN_T) void kpp_draw_kernel(KppCtl *ctl, const float *__restrict__ probs, int32_t n,
                                                         double *cdf, double *tile_sum, int n_tiles, double u,
                                                         double tol, int32_t *centers, uint8_t *is_center,
                                                         int32_t next_pos, int32_t iter, int begin_next, double *d,
                                                         int32_t *front0) {
    if (ctl->abort_iter >= 0) return;
    __shared__ int32_t pick[3];
    if (!kpp_draw_body(ctl, ctl->total, probs, n, cdf, tile_sum, n_tiles, u, tol, blockIdx.x, gridDim.x, pick)) return;
    if (threadIdx.x == 0) {
        ctl->ticket[1][0] = 0;
        if (!pick[0] || !pick[2]) { ctl->abort_iter = iter; ctl->abort_reason = 2; return; }
        centers[next_pos] = pick[1];
        is_center[pick[1]] = 1;
        if (begin_next) kpp_begin(ctl, centers, next_pos, d, front0);
    }
}

// ---- the whole chain as ONE kernel launched over and over -------------------------------------------------------
// A solve's launches cannot be counted in advance (its sweeps end when the frontier is empty), and the host must
// not wait for every centre.  So every launch runs "the next step": it reads the state the previous launch left
// (double buffered by launch parity: all blocks read S[parity], one thread writes S[parity ^ 1]) and either
//   SOLVE: relaxes the current frontier -- or, when that is empty, the solve has converged and the SAME launch
//          applies it, sums the draw weights (kpp_sum_body) and its last block draws on the un-normalised weights
//          (kpp_approx_draw), commits the next centre and opens its solve; the last centre is only applied;
//   DRAW:  only when that draw declined (u within 1.25e-7 of a step): the exact scan, pick and commit (kpp_draw_body);
//   DONE:  nothing (launches enqueued beyond the end of the chain).
// Used once d_min is finite everywhere (no maximum pass needed); no launch is spent on an empty frontier.
struct KppState {
    int32_t mode, t, sw, stamp;        // mode 0 SOLVE (+ sum + approximate draw when converged), 3 DRAW (exact), 2 DONE; centre index; sweep of its solve; solve counter
    int32_t launches, pad;
};

template <bool WEIGHTED>
__global__ __launch_bounds__(256) void kpp_step_kernel(KppCtl *ctl, KppState *state, int lidx,
                                                      const int32_t *__restrict__ indptr,
                                                      const int32_t *__restrict__ indices,
                                                      const float *__restrict__ weights, int32_t n, double *d,
                                                      float *dmin, int32_t *argmin, int32_t *mark, int32_t *front_a,
                                                      int32_t *front_b, int32_t *centers, uint8_t *is_center,
                                                      float *probs, double *cdf, double *tile_sum, int n_tiles,
                                                      const double *__restrict__ u_dev, double tol, SumPlan pl,
                                                      int32_t it1, int32_t n_centers_total) {
    // lidx = launch index mod 6: state parity, frontier-count ring slot and queue buffer all follow the launch
    // index, so their addresses are known before the state arrives (one dependent round trip less per launch)
    const int parity = lidx & 1;
    const int cur = lidx % 3, next = (lidx + 1) % 3, clear = (lidx + 2) % 3;
    int32_t *fin = parity ? front_b : front_a, *fout = parity ? front_a : front_b;
    const int32_t cnt = ctl->fcount[cur][0];
    const KppState S = state[parity];
    KppState *out = &state[parity ^ 1];
    if (S.mode == 2 || ctl->abort_iter >= 0) {
        if (blockIdx.x == 0 && threadIdx.x == 0) { KppState x = S; x.mode = 2; *out = x; }
        return;
    }
    // commit a pick: the next centre, and the opening of its solve (the next launch reads slot `next`, queue `fout`)
    auto commit = [&](KppState &x, int32_t src) {
        centers[S.t + 1] = src;
        is_center[src] = 1;
        if (S.t + 1 < it1) {
            st_dev(&d[src], 0.0);                          // (agent scope: another block of this launch may have reset d[src])
            fout[0] = src;
            ctl->fcount[next][0] = 1;
            ctl->fcount[clear][0] = 0;
            x.mode = 0; x.t = S.t + 1; x.sw = 0; x.stamp = S.stamp + 1;
        } else {
            x.mode = 2; x.t = S.t + 1;
        }
    };
    __shared__ int32_t pick[3];
    if (S.mode == 0) {
        if (cnt > 0) {                                                    // one more sweep of this solve
            if (blockIdx.x == 0 && threadIdx.x == 0) {
                ctl->fcount[clear][0] = 0;
                KppState x = S;
                x.sw = S.sw + 1; x.launches = S.launches + 1;
                if (x.sw >= 4094) { ctl->abort_iter = S.t; ctl->abort_reason = 1; x.mode = 2; }   // stamp range
                *out = x;
            }
            kpp_push_body<WEIGHTED>(ctl, indptr, indices, weights, d, dmin, mark, fin, fout, cnt, next,
                                    S.stamp * 4096 + S.sw + 1, cnt >= 2048);
            return;
        }
        // converged
        if (S.t + 1 >= n_centers_total) {                                 // last centre: apply, no draw
            for (int32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
                const double dd = d[i];
                if (dd < inf64()) {
                    const float x = (float)dd;
                    if (x < dmin[i]) { dmin[i] = x; argmin[i] = S.t; }
                    d[i] = inf64();
                }
            }
            if (blockIdx.x == 0 && threadIdx.x == 0) { KppState x = S; x.mode = 2; x.launches = S.launches + 1; *out = x; }
            return;
        }
        const int nsum = (pl.n_leaves + SUM_LEAVES - 1) / SUM_LEAVES;
        if ((int)blockIdx.x >= nsum) return;
        float total = 0.0f;
        if (!kpp_sum_body(ctl, dmin, argmin, d, 1, S.t, is_center, nullptr, nullptr, 0, 0, probs, pl, blockIdx.x, nsum,
                          &total, cdf))                                   // (cdf[] carries the fp64 leaf sums to the draw)
            return;
        // ... and the draw, first on the un-normalised weights, by this last block
        kpp_approx_draw(cdf, dmin, is_center, pl, n, u_dev[S.t], KPP_STEP_APPROX_MARGIN + tol, pick);
        if (threadIdx.x == 0) {
            ctl->total = total;
            ctl->ticket[0][0] = 0;
            KppState x = S;
            x.launches = S.launches + 1;
            if (!(total > 0.0f)) { ctl->abort_iter = S.t; ctl->abort_reason = 3; x.mode = 2; }
            else if (pick[1]) commit(x, pick[0]);
            else x.mode = 3;                                              // u too close to a step of the approximate cdf
            *out = x;
        }
        return;
    }
    // DRAW, exact (mode 3)
    if ((int)blockIdx.x >= n_tiles) return;
    if (!kpp_draw_body(ctl, ctl->total, probs, n, cdf, tile_sum, n_tiles, u_dev[S.t], tol, blockIdx.x, n_tiles, pick))
        return;
    if (threadIdx.x == 0) {
        ctl->ticket[1][0] = 0;
        KppState x = S;
        x.launches = S.launches + 1;
        if (!pick[0] || !pick[2]) {
            ctl->abort_iter = S.t; ctl->abort_reason = 2; x.mode = 2;
        } else {
            commit(x, pick[1]);
        }
        *out = x;
    }
}

// ---- the chain as ONE resident workgroup ----------------------------------------------------------------------
// Once the cells are small (a few thousand nodes) a centre costs microseconds of work but ~6 dependent kernel launches
// of ~9 us each.  Here ONE 1024-thread workgroup stays resident and runs centre after centre, with workgroup barriers
// and LDS where the step kernel has kernel boundaries and agent-scope round trips:
//   solve   the pruned frontier solve with the tentative fp64 distances in an LDS hash table (open addressing,
//           ds_cmpst on the key, ds_min_u64 on the bit pattern of the distance): a candidate that cannot improve its
//           node (cand > d_min + tau) is never stored, so the table only holds the new centre's cell; frontier queues of
//           table slots in LDS, one "queued" bit per entry.  Nothing global is written before the solve has converged;
//   apply   d_min / argmin of the improved nodes (the only global stores), their numpy leaves marked dirty;
//   sum     numpy's float32 add.reduce incrementally: the tree of leaf sums lives in LDS, only dirty leaves are
//           re-read (the tree's shape depends on N alone, so this is numpy's result bit for bit), then the levels;
//   draw    p = w / total, per-leaf fp64 sums, scan over the leaves, pick inside one leaf with the same margin rule as
//           kpp_draw_body; commit and open the next solve.
// The same arithmetic as the step kernel (fp64 candidate = du + w, float32 w = d_min^2 without contraction, correctly
// rounded division), so centres, assignments and d_min are identical.  A cell that does not fit the table ends the
// kernel with abort reason 4 and nothing of that iteration applied; the caller runs that one centre with the step kernel.
constexpr int PG_THREADS = 1024, PG_HASH = 8192, PG_HASH_MAX = 4096, PG_QCAP = 4096, PG_TOUCH = 5120, PG_MAX_LEAVES = 768,
              PG_TILE = 256, PG_SEGCAP = 2048;
constexpr double KPP_APPROX_MARGIN = 1.25e-7;          // > 2 * 2^-24 + N * 2^-52 for every N the resident chain accepts
constexpr unsigned PG_EMPTY = 0xffffffffu, PG_FLAG = 0x80000000u, PG_MASK = 0x7fffffffu;
constexpr int32_t PG_MAX_NODES = 90000;

struct ResidentPlan {                      // numpy's reduction tree (SumPlan) + what the resident kernel adds
    const int32_t *leaf_start, *leaf_len, *node_l, *node_r, *level_off, *chunk_root;
    const uint16_t *tail_leaf;             // leaf of node v >= tail_start: tail_leaf[v - tail_start]
    int n_leaves, n_nodes, n_levels, n_chunks;
    int32_t tail_start, tail_first_leaf;   // nodes below tail_start sit in full 8192-chunks: 64 leaves of 128
};

template <bool WEIGHTED>
__global__ __launch_bounds__(PG_THREADS) void kpp_resident_kernel(KppCtl *ctl, const int32_t *__restrict__ indptr,
                                                                  const int32_t *__restrict__ indices,
                                                                  const float *__restrict__ weights, int32_t n,
                                                                  float *dmin, int32_t *argmin, int32_t *centers,
                                                                  uint8_t *is_center, const double *__restrict__ u_dev,
                                                                  double tol, ResidentPlan pl, int32_t it0, int32_t it1,
                                                                  int32_t n_centers_total, int32_t *progress,
                                                                  unsigned long long *prof) {
    // prof (diagnostic runs of the CALLER only, normally null): cycles per phase summed over the centres,
    // [0] solve [1] apply [2] leaf sums [3] locate [4] exact draws (tree + full pass + locate) [5] exact draws taken
    // [6] sweeps [7] frontier nodes
    unsigned long long pt = 0, pacc[14] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};   // [8] init [9] marks [10] fill [11] pass-rest [12] pass loads [13] pass relax
#define GEO_STAMP(k)                                                        \
    if (prof && threadIdx.x == 0) {                                         \
        const unsigned long long now_ = __builtin_amdgcn_s_memtime();       \
        pacc[k] += now_ - pt;                                               \
        pt = now_;                                                          \
    }
    __shared__ unsigned long long h_dist[PG_HASH];
    __shared__ unsigned h_key[PG_HASH];
    __shared__ unsigned short queue[2][PG_QCAP], touched[PG_TOUCH];
    __shared__ double n_du[PG_TILE];
    __shared__ int32_t n_e0[PG_TILE], n_e1[PG_TILE], s_level_off[32], s_chunk_root[64];
    // the segment list of a frontier tile (solve) and the exact per-leaf sums of p (an exact draw) never live together
    __shared__ __attribute__((aligned(8))) unsigned char s_scratch[PG_SEGCAP * 4 > (PG_MAX_LEAVES + 1) * 8 ? PG_SEGCAP * 4 : (PG_MAX_LEAVES + 1) * 8];
    unsigned *segs = reinterpret_cast<unsigned *>(s_scratch);
    double *leaf_q = reinterpret_cast<double *>(s_scratch);
    __shared__ double leaf_a[PG_MAX_LEAVES + 1], leaf_e[PG_MAX_LEAVES + 1];
    __shared__ float tree[2 * PG_MAX_LEAVES];
    __shared__ unsigned short t_l[PG_MAX_LEAVES], t_r[PG_MAX_LEAVES], dirty_list[PG_MAX_LEAVES];
    __shared__ unsigned char dirty[PG_MAX_LEAVES];
    __shared__ double wave_tot[PG_THREADS / 64];
    __shared__ float s_red[PG_THREADS / 64];
    __shared__ int32_t s_qcnt[2], s_entries, s_overflow, s_ndirty, s_nseg, s_leaf, s_pick[2];
    __shared__ float s_total;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n_leaves = pl.n_leaves, n_nodes = pl.n_nodes;
    const unsigned long long lane_lt = (1ull << lane) - 1ull;
    // ---- prologue: pruning margin (max of the finite d_min), tree plan into LDS, every leaf dirty, empty table ----
    {
        float m = 0.0f;
        for (int32_t i = tid; i < n; i += PG_THREADS) m = fmaxf(m, dmin[i]);
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) m = fmaxf(m, __shfl_xor(m, off, 64));
        if (lane == 0) s_red[wave] = m;
        for (int j = tid; j < n_nodes; j += PG_THREADS) { t_l[j] = (unsigned short)pl.node_l[j]; t_r[j] = (unsigned short)pl.node_r[j]; }
        for (int j = tid; j < n_leaves; j += PG_THREADS) dirty[j] = 1;
        if (tid <= pl.n_levels && tid < 32) s_level_off[tid] = pl.level_off[tid];
        if (tid < pl.n_chunks && tid < 64) s_chunk_root[tid] = pl.chunk_root[tid];
        for (int sl = tid; sl < PG_HASH; sl += PG_THREADS) { h_key[sl] = PG_EMPTY; h_dist[sl] = 0x7ff0000000000000ull; }
        if (tid == 0) s_entries = 0;
    }
    __syncthreads();
    float maxf = 0.0f;
#pragma unroll
    for (int w = 0; w < PG_THREADS / 64; ++w) maxf = fmaxf(maxf, s_red[w]);
    const double tau = maxf > 0.f ? 1e-6 * (double)maxf : 0.0;
    const int sub = tid & 31, half = (tid >> 5) & 1;           // 32 lanes per row segment, two segments per wave

    // Four candidates per lane (one per row segment the half wave holds), every lane of the wave takes part.  A candidate
    // is stored only if it can still improve its node.  The four table look-ups, inserts, 64-bit minima and queue marks
    // are issued side by side (one dependent LDS chain for all four); table / queue tails advance once per wave.
    auto relax4 = [&](const bool (&pred)[4], const int32_t (&v)[4], const double (&cand)[4], const float (&dv)[4], int nxt) {
        unsigned h[4], k[4];
        bool want[4], found[4], isnew[4], slow[4], qf[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            want[q] = pred[q] && (cand[q] <= (double)dv[q] + tau);     // else: cannot improve v, never stored
            h[q] = ((unsigned)v[q] * 2654435761u) >> 19;
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) k[q] = want[q] ? h_key[h[q]] : 0u;
        unsigned old[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            found[q] = want[q] && k[q] != PG_EMPTY && (k[q] & PG_MASK) == (unsigned)v[q];
            old[q] = (want[q] && k[q] == PG_EMPTY) ? atomicCAS(&h_key[h[q]], PG_EMPTY, (unsigned)v[q]) : 0u;
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            isnew[q] = want[q] && k[q] == PG_EMPTY && old[q] == PG_EMPTY;
            if (want[q] && k[q] == PG_EMPTY && old[q] != PG_EMPTY && (old[q] & PG_MASK) == (unsigned)v[q]) found[q] = true;
            found[q] = found[q] || isnew[q];
            slow[q] = want[q] && !found[q];
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {                                  // collisions (rare while the table is sparse)
            if (slow[q]) {
                for (int probe = 0; probe < 192; ++probe) {            // bounded: a long chain means the table is too full
                    h[q] = (h[q] + 1) & (PG_HASH - 1);
                    const unsigned kk = h_key[h[q]];
                    if (kk != PG_EMPTY && (kk & PG_MASK) == (unsigned)v[q]) { found[q] = true; break; }
                    if (kk == PG_EMPTY) {
                        const unsigned o2 = atomicCAS(&h_key[h[q]], PG_EMPTY, (unsigned)v[q]);
                        if (o2 == PG_EMPTY) { found[q] = isnew[q] = true; break; }
                        if ((o2 & PG_MASK) == (unsigned)v[q]) { found[q] = true; break; }
                    }
                }
                if (!found[q]) s_overflow = 1;
            }
        }
        unsigned long long od[4];
#pragma unroll
        for (int q = 0; q < 4; ++q)
            od[q] = found[q] ? atomicMin(&h_dist[h[q]], (unsigned long long)__double_as_longlong(cand[q])) : 0ull;
        unsigned k0[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const bool better = found[q] && (unsigned long long)__double_as_longlong(cand[q]) < od[q];
            k0[q] = better ? atomicOr(&h_key[h[q]], PG_FLAG) : PG_FLAG;
        }
        unsigned long long mn[4], mq[4];
        int cn = 0, cq = 0;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            qf[q] = !(k0[q] & PG_FLAG);
            mn[q] = __ballot(isnew[q]);
            mq[q] = __ballot(qf[q]);
            cn += __popcll(mn[q]);
            cq += __popcll(mq[q]);
        }
        if (cn) {                                                      // wave-uniform
            int base = 0;
            if (lane == 0) base = atomicAdd(&s_entries, cn);
            base = __builtin_amdgcn_readfirstlane(base);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                if (isnew[q]) {
                    const int at = base + __popcll(mn[q] & lane_lt);
                    if (at < PG_TOUCH) touched[at] = (unsigned short)h[q]; else s_overflow = 1;
                }
                base += __popcll(mn[q]);
            }
        }
        if (cq) {
            int base = 0;
            if (lane == 0) base = atomicAdd(&s_qcnt[nxt], cq);
            base = __builtin_amdgcn_readfirstlane(base);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                if (qf[q]) {
                    const int at = base + __popcll(mq[q] & lane_lt);
                    if (at < PG_QCAP) queue[nxt][at] = (unsigned short)h[q]; else s_overflow = 1;
                }
                base += __popcll(mq[q]);
            }
        }
    };

    // searchsorted(cdf / cdf[-1], u, 'right') over per-leaf sums src[] and per-element values val(i): exclusive scan of the
    // leaf sums, the leaf holding u * sum, the position inside it.  s_pick = {index, 1 if u clears both neighbouring cdf
    // steps by margin_rel * sum}.  All comparisons are made on cdf * sum against u * sum (no divisions).
    auto locate = [&](const double *src, double u, double margin_rel, auto val) {
        double incl = 0.0, mine = 0.0;
        if (tid < PG_MAX_LEAVES) {
            mine = tid < n_leaves ? src[tid] : 0.0;
            incl = mine;
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) {
                const double o = __shfl_up(incl, off, 64);
                if (lane >= off) incl += o;
            }
            if (lane == 63) wave_tot[wave] = incl;
        }
        if (tid == 0) s_leaf = 0;
        __syncthreads();
        if (tid < PG_MAX_LEAVES) {
            double base = 0.0;
            for (int w = 0; w < wave; ++w) base += wave_tot[w];
            if (tid < n_leaves) leaf_e[tid] = base + (incl - mine);
            if (tid == n_leaves - 1) leaf_e[n_leaves] = base + incl;
        }
        __syncthreads();
        const double S = leaf_e[n_leaves];
        const double target = u * S;
        if (tid < n_leaves && leaf_e[tid] <= target) atomicMax(&s_leaf, tid);       // leaf_e[0] = 0 <= target
        __syncthreads();
        if (wave == 0) {
            const int L = s_leaf;
            const bool full = L < pl.tail_first_leaf;
            const int32_t b0 = full ? L * PW_BLOCK : pl.leaf_start[L];
            const int len = full ? PW_BLOCK : pl.leaf_len[L];
            const int i0 = 2 * lane, i1 = 2 * lane + 1;
            const float x0 = i0 < len ? dmin[b0 + i0] : 0.0f, x1 = i1 < len ? dmin[b0 + i1] : 0.0f;
            const double q0 = i0 < len ? val(x0) : 0.0;
            const double q1 = i1 < len ? val(x1) : 0.0;
            double inc = q0 + q1;
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) {
                const double o = __shfl_up(inc, off, 64);
                if (lane >= off) inc += o;
            }
            const double c1 = leaf_e[L] + inc, c0 = leaf_e[L] + (inc - q1);
            const bool le0 = i0 < len && c0 <= target, le1 = i1 < len && c1 <= target;
            const int count = __popcll(__ballot(le0)) + __popcll(__ballot(le1));
            // c is non-decreasing: the largest c <= target is element count-1 (or the previous leaf's end), the smallest
            // c > target is element count
            const int lo_i = count - 1, hi_i = count;
            const double lo_v = __shfl((lo_i & 1) ? c1 : c0, (lo_i >> 1) & 63, 64);
            const double hi_v = __shfl((hi_i & 1) ? c1 : c0, (hi_i >> 1) & 63, 64);
            if (lane == 0) {
                const double lower = count > 0 ? lo_v : leaf_e[L];
                const int32_t idx = b0 + count;
                const double margin = margin_rel * S;
                s_pick[0] = idx;
                s_pick[1] = (count < len && idx < n && S > 0.0 && S < 1e37 && (target - lower > margin) && (hi_v - target > margin)) ? 1 : 0;
            }
        }
        __syncthreads();
    };

    if (prof && threadIdx.x == 0) pt = __builtin_amdgcn_s_memtime();
    for (int32_t t = it0; t < it1; ++t) {
        // ---- solve from centers[t]: forget the previous cell, seed the table with the source ----
        {
            const int32_t prev = s_entries < PG_TOUCH ? s_entries : PG_TOUCH;
            for (int i = tid; i < prev; i += PG_THREADS) { const unsigned sl = touched[i]; h_key[sl] = PG_EMPTY; h_dist[sl] = 0x7ff0000000000000ull; }
        }
        __syncthreads();
        if (tid == 0) {
            const unsigned src = (unsigned)centers[t];
            const unsigned h = (src * 2654435761u) >> 19;
            h_key[h] = src;
            h_dist[h] = 0ull;
            queue[0][0] = (unsigned short)h;
            touched[0] = (unsigned short)h;
            s_qcnt[0] = 1; s_qcnt[1] = 0; s_entries = 1; s_overflow = 0;
        }
        __syncthreads();
        GEO_STAMP(8)
        int cur = 0;
        int32_t sweeps = 0;
        for (;;) {
            const int32_t cnt = s_qcnt[cur];
            if (cnt == 0 || s_overflow) break;
            if (prof && tid == 0) { pacc[6] += 1; pacc[7] += cnt; }
            for (int32_t i = tid; i < cnt; i += PG_THREADS) atomicAnd(&h_key[queue[cur][i]], PG_MASK);   // dequeue marks
            if (tid == 0) { s_qcnt[cur ^ 1] = 0; s_nseg = 0; }
            __syncthreads();
            GEO_STAMP(9)
            // The frontier in tiles of PG_TILE nodes.  Thread per node: row bounds + distance into LDS and one descriptor
            // per 32-entry segment of its row into the tile's segment list (one round trip for the whole tile).  Then 32
            // lanes per segment, four segments per half wave at a time: 128 segments' entries and the d_min of their far
            // ends are in flight together (two dependent round trips per pass), whatever the degrees are.
            for (int32_t base = 0; base < cnt; base += PG_TILE) {
                const int m = cnt - base < PG_TILE ? cnt - base : PG_TILE;
                if (tid < PG_TILE) {                                   // waves 0 .. PG_TILE/64-1, whole waves
                    int nseg = 0;
                    if (tid < m) {
                        const unsigned slot = queue[cur][base + tid];
                        const int32_t u = (int32_t)(h_key[slot] & PG_MASK);
                        n_du[tid] = __longlong_as_double((long long)h_dist[slot]);
                        const int32_t e0 = indptr[u], e1 = indptr[u + 1];
                        n_e0[tid] = e0;
                        n_e1[tid] = e1;
                        nseg = (e1 - e0 + 31) >> 5;
                    }
                    int incl = nseg;                                   // wave scan of the segment counts, one tail update per wave
#pragma unroll
                    for (int off = 1; off < 64; off <<= 1) {
                        const int o = __shfl_up(incl, off, 64);
                        if (lane >= off) incl += o;
                    }
                    const int wtot = __shfl(incl, 63, 64);
                    int sbase = 0;
                    if (lane == 0 && wtot) sbase = atomicAdd(&s_nseg, wtot);
                    sbase = __builtin_amdgcn_readfirstlane(sbase) + incl - nseg;
                    for (int k = 0; k < nseg; ++k)
                        if (sbase + k < PG_SEGCAP) segs[sbase + k] = (unsigned)tid | ((unsigned)k << 8); else s_overflow = 1;
                }
                __syncthreads();
                GEO_STAMP(10)
                const int nseg_all = s_nseg < PG_SEGCAP ? s_nseg : PG_SEGCAP;
                for (int sw = wave * 8; sw < nseg_all; sw += (PG_THREADS / 64) * 8) {   // wave-uniform trip count
                    if (*(volatile int32_t *)&s_overflow) break;
                    const int s0 = sw + 4 * half;
                    int32_t v[4], en[4];
                    float w[4], dv[4];
                    double cand[4];
                    bool ok[4];
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const bool valid = s0 + q < nseg_all;
                        const unsigned desc = valid ? segs[s0 + q] : 0u;
                        const int j = desc & 255u;
                        en[q] = n_e0[j] + (int32_t)((desc >> 8) << 5) + sub;
                        ok[q] = valid && en[q] < n_e1[j];
                        cand[q] = n_du[j];
                        v[q] = ok[q] ? indices[en[q]] : 0;
                        w[q] = ok[q] ? (WEIGHTED ? weights[en[q]] : 1.0f) : 0.0f;
                    }
#pragma unroll
                    for (int q = 0; q < 4; ++q) { dv[q] = ok[q] ? dmin[v[q]] : 0.0f; cand[q] += (double)w[q]; }
                    if (prof) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); GEO_STAMP(12) }
                    relax4(ok, v, cand, dv, cur ^ 1);
                    if (prof) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); GEO_STAMP(13) }
                }
                __syncthreads();
                if (s_entries > PG_HASH_MAX) s_overflow = 1;           // same value in every thread after the barrier
                if (tid == 0) s_nseg = 0;
                __syncthreads();
                GEO_STAMP(11)
            }
            if (++sweeps >= 4094) s_overflow = 1;
            cur ^= 1;
        }
        if (s_overflow) {                                         // cell too large for the table: nothing was applied
            if (tid == 0) { ctl->abort_iter = t; ctl->abort_reason = 4; }
            break;
        }
        GEO_STAMP(0)
        // ---- apply: d_min / argmin (kmeans_optimized.py:44 + single-pass assignment), dirty leaves ----
        {
            const int32_t cnt = s_entries;
            for (int i = tid; i < cnt; i += PG_THREADS) {
                const unsigned sl = touched[i];
                const int32_t v = (int32_t)(h_key[sl] & PG_MASK);
                const float x = (float)__longlong_as_double((long long)h_dist[sl]);
                if (x < dmin[v]) {
                    dmin[v] = x;
                    argmin[v] = t;
                    const int leaf = v < pl.tail_start ? ((v >> 13) << 6) + ((v & 8191) >> 7) : (int)pl.tail_leaf[v - pl.tail_start];
                    dirty[leaf] = 1;
                }
            }
        }
        if (tid == 0) { s_ndirty = 0; *progress = t + 1; }
        __syncthreads();                                          // stores of this workgroup are visible to its later loads
        if (t + 1 >= n_centers_total) break;                      // last centre: applied, no draw
        for (int j0 = wave * 64; j0 < n_leaves; j0 += PG_THREADS) {                    // wave-uniform; list tail once per wave
            const int j = j0 + lane;
            const bool d = j < n_leaves && dirty[j];
            if (d) dirty[j] = 0;
            const unsigned long long md = __ballot(d);
            if (md) {
                int base = 0;
                if (lane == 0) base = atomicAdd(&s_ndirty, __popcll(md));
                base = __builtin_amdgcn_readfirstlane(base);
                if (d) dirty_list[base + __popcll(md & lane_lt)] = (unsigned short)j;
            }
        }
        __syncthreads();
        GEO_STAMP(1)
        // ---- leaves that changed: numpy's float32 leaf sum (8 strided accumulators: the leaf level of its add.reduce
        //      tree, kept current for exact draws) and the fp64 sum of the same weights (the approximate cdf) ----
        {
            const int j = tid & 7;
            const int nd = s_ndirty;
            for (int i = tid >> 3; i < nd; i += PG_THREADS / 8) {
                const int leaf = dirty_list[i];
                const bool full = leaf < pl.tail_first_leaf;
                const float *a = dmin + (full ? leaf * PW_BLOCK : pl.leaf_start[leaf]);
                const int len = full ? PW_BLOCK : pl.leaf_len[leaf];
                const int m8 = len - (len % 8);
                float r = 0.0f;
                double acc = 0.0;
                if (len == PW_BLOCK) {                            // all 16 loads of the lane in flight together
                    float x[16];
#pragma unroll
                    for (int q = 0; q < 16; ++q) x[q] = a[8 * q + j];
#pragma unroll
                    for (int q = 0; q < 16; ++q) x[q] = __fmul_rn(x[q], x[q]);
                    r = x[0];
                    acc = (double)x[0];
#pragma unroll
                    for (int q = 1; q < 16; ++q) { r = __fadd_rn(r, x[q]); acc += (double)x[q]; }
                } else if (len >= 8) {
                    r = __fmul_rn(a[j], a[j]);
                    acc = (double)r;
                    for (int q = 8; q < m8; q += 8) { const float wq = __fmul_rn(a[q + j], a[q + j]); r = __fadd_rn(r, wq); acc += (double)wq; }
                }
                float o = __shfl_down(r, 1, 8);
                if ((j & 1) == 0) r = __fadd_rn(r, o);
                o = __shfl_down(r, 2, 8);
                if ((j & 3) == 0) r = __fadd_rn(r, o);
                o = __shfl_down(r, 4, 8);
                acc += __shfl_down(acc, 1, 8);
                acc += __shfl_down(acc, 2, 8);
                acc += __shfl_down(acc, 4, 8);
                if (j == 0) {
                    r = __fadd_rn(r, o);
                    if (len < 8) {
                        r = 0.0f;
                        acc = 0.0;
                        for (int q = 0; q < len; ++q) { const float wq = __fmul_rn(a[q], a[q]); r = __fadd_rn(r, wq); acc += (double)wq; }
                    } else {
                        for (int q = m8; q < len; ++q) { const float wq = __fmul_rn(a[q], a[q]); r = __fadd_rn(r, wq); acc += (double)wq; }
                    }
                    tree[leaf] = r;
                    leaf_a[leaf] = acc;
                }
            }
        }
        __syncthreads();
        GEO_STAMP(2)
        // ---- draw, first on the un-normalised weights: cdf[j] / cdf[-1] of RandomState.choice differs from
        //      sum_{i<=j} w_i / sum_i w_i by less than 2 * 2^-24 + N * 2^-52 (each p_i = fl32(w_i / total) is within 2^-24
        //      of w_i / total, the fp64 cumsum within N * 2^-53): if u clears both neighbouring steps of the approximate
        //      cdf by KPP_APPROX_MARGIN the index is numpy's, and neither total nor any p_i was needed ----
        const double u = u_dev[t];
        locate(leaf_a, u, KPP_APPROX_MARGIN + tol, [](float x) { return (double)__fmul_rn(x, x); });
        GEO_STAMP(3)
        if (!s_pick[1]) {
            // ---- exact draw: numpy's float32 add.reduce (tree levels over the current leaf sums, chunk roots in order),
            //      p = float64(w / total) per leaf, the same search with the fp64-scan tolerance only ----
            for (int lv = 0; lv < pl.n_levels; ++lv) {
                for (int j = s_level_off[lv] + tid; j < s_level_off[lv + 1]; j += PG_THREADS)
                    tree[n_leaves + j] = __fadd_rn(tree[t_l[j]], tree[t_r[j]]);
                __syncthreads();
            }
            if (tid == 0) {
                float total = 0.0f;
                for (int c = 0; c < pl.n_chunks; ++c) total = __fadd_rn(total, tree[s_chunk_root[c]]);
                s_total = total;
            }
            __syncthreads();
            const float total = s_total;
            if (!(total > 0.0f)) {                                // degenerate weights: the solve IS applied, the draw is not
                if (tid == 0) { ctl->abort_iter = t; ctl->abort_reason = 3; }
                break;
            }
            {
                const int j = tid & 7;
                const int g = tid >> 3;
                // full leaves: each of the 8 lanes takes 16 consecutive elements as four 16-byte loads
                for (int leaf = g; leaf < pl.tail_first_leaf; leaf += PG_THREADS / 8) {
                    const float4 *a = reinterpret_cast<const float4 *>(dmin + (size_t)leaf * PW_BLOCK) + 4 * j;
                    float4 xa[4];
#pragma unroll
                    for (int q = 0; q < 4; ++q) xa[q] = a[q];
                    double acc = 0.0;
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        acc += (double)__fdiv_rn(__fmul_rn(xa[q].x, xa[q].x), total) + (double)__fdiv_rn(__fmul_rn(xa[q].y, xa[q].y), total);
                        acc += (double)__fdiv_rn(__fmul_rn(xa[q].z, xa[q].z), total) + (double)__fdiv_rn(__fmul_rn(xa[q].w, xa[q].w), total);
                    }
                    acc += __shfl_down(acc, 1, 8);
                    acc += __shfl_down(acc, 2, 8);
                    acc += __shfl_down(acc, 4, 8);
                    if (j == 0) leaf_q[leaf] = acc;
                }
                for (int leaf = pl.tail_first_leaf + g; leaf < n_leaves; leaf += PG_THREADS / 8) {   // last (partial) numpy chunk
                    const float *a = dmin + pl.leaf_start[leaf];
                    const int len = pl.leaf_len[leaf];
                    double acc = 0.0;
                    for (int q = j; q < len; q += 8) acc += (double)__fdiv_rn(__fmul_rn(a[q], a[q]), total);
                    acc += __shfl_down(acc, 1, 8);
                    acc += __shfl_down(acc, 2, 8);
                    acc += __shfl_down(acc, 4, 8);
                    if (j == 0) leaf_q[leaf] = acc;
                }
            }
            __syncthreads();
            locate(leaf_q, u, tol, [total](float x) { return (double)__fdiv_rn(__fmul_rn(x, x), total); });
            if (prof && tid == 0) pacc[5] += 1;
            GEO_STAMP(4)
            if (!s_pick[1]) {                                     // u within rounding reach of a cdf step: host repeats the draw
                if (tid == 0) { ctl->abort_iter = t; ctl->abort_reason = 2; }
                break;
            }
        }
        if (tid == 0) {
            centers[t + 1] = s_pick[0];
            is_center[s_pick[0]] = 1;
        }
        __syncthreads();
    }
    if (prof && threadIdx.x == 0)
        for (int k = 0; k < 14; ++k) prof[k] = pacc[k];
#undef GEO_STAMP
}

// ------------------------------------------------------------------ host: numpy's reduction tree
struct PwPlan {
    std::vector<int32_t> leaf_start, leaf_len, node_l, node_r, node_level, chunk_root;
};

int32_t pw_build(PwPlan &p, int32_t start, int32_t n, int *level) {
    if (n <= PW_BLOCK) {
        p.leaf_start.push_back(start);
        p.leaf_len.push_back(n);
        *level = 0;
        return (int32_t)p.leaf_start.size() - 1;             // leaf ids: 0 .. L-1
    }
    int32_t n2 = n / 2;
    n2 -= n2 % 8;
    int ll, lr;
    const int32_t l = pw_build(p, start, n2, &ll);
    const int32_t r = pw_build(p, start + n2, n - n2, &lr);
    *level = (ll > lr ? ll : lr) + 1;
    p.node_l.push_back(l);
    p.node_r.push_back(r);
    p.node_level.push_back(*level);
    return -(int32_t)p.node_l.size();                         // node ids: encoded -(k+1), fixed up below
}

struct DevPlan {
    int n_leaves, n_nodes, n_levels, n_chunks;
    int32_t *leaf_start, *leaf_len, *node_l, *node_r, *level_off, *chunk_root;
    float *val;
};

struct KppWs {
    KppCtl *ctl;
    double *d, *cdf, *tile_sum, *tile_off, *u_dev;
    KppState *state;
    int32_t *mark, *front[2], *part_inf;
    float *probs, *part_max;
    DevPlan plan;
    int32_t *plan_blob;
    size_t plan_ints;
    uint16_t *tail_leaf;                   // resident chain: leaf of the nodes of the last (partial) numpy chunk
    int32_t *progress;
};

size_t plan_ints_bound(int32_t n) {
    const size_t leaves = (size_t)n / 64 + (size_t)n / NP_BUFSIZE + 16;      // leaves have >= 64 elements except per chunk tails
    return 6 * leaves + 64 + 2 * CPLAN_INTS + (size_t)n / NP_BUFSIZE + 1;      // + the per-buffer plans and tickets
}

bool carve(void *ws, size_t ws_bytes, int32_t n, KppWs *o) {
    geo::Arena ar(ws, ws_bytes);
    o->ctl = ar.take<KppCtl>(4);
    o->state = ar.take<KppState>(4);
    o->u_dev = ar.take<double>((size_t)n);
    o->d = ar.take<double>((size_t)n);
    o->cdf = ar.take<double>((size_t)n);
    const size_t tiles = ((size_t)n + SCAN_TILE - 1) / SCAN_TILE;
    o->tile_sum = ar.take<double>(tiles + 1);
    o->tile_off = ar.take<double>(tiles + 1);
    o->mark = ar.take<int32_t>((size_t)n);
    o->front[0] = ar.take<int32_t>((size_t)n);
    o->front[1] = ar.take<int32_t>((size_t)n);
    o->probs = ar.take<float>((size_t)n);
    o->part_max = ar.take<float>(FINISH_GRID);
    o->part_inf = ar.take<int32_t>(FINISH_GRID);
    o->plan_ints = plan_ints_bound(n);
    o->plan_blob = ar.take<int32_t>(o->plan_ints);
    o->plan.val = ar.take<float>(o->plan_ints);
    o->tail_leaf = ar.take<uint16_t>(NP_BUFSIZE);
    o->progress = ar.take<int32_t>(4);
    return o->progress != nullptr;
}

}  // namespace

extern "C" size_t geo_kpp_workspace_bytes(int32_t n) {
    if (n <= 0) return 4096;
    const size_t tiles = ((size_t)n + SCAN_TILE - 1) / SCAN_TILE;
    return geo::align_up(4 * sizeof(KppCtl)) + geo::align_up(4 * sizeof(KppState)) + 3 * geo::align_up((size_t)n * 8) + 2 * geo::align_up((tiles + 1) * 8) +
           5 * geo::align_up((size_t)n * 4) + 2 * geo::align_up(FINISH_GRID * 4) +
           2 * geo::align_up(plan_ints_bound(n) * 4) + geo::align_up(NP_BUFSIZE * 2) + 4096;
}

extern "C" int32_t geo_kpp_resident_max_nodes(void) { return PG_MAX_NODES; }

extern "C" int geo_kpp_chain(const int32_t *indptr, const int32_t *indices, const float *weights, int32_t n,
                             int32_t *centers, uint8_t *is_center, float *dmin, int32_t *argmin, const double *u_host,
                             int32_t it0, int32_t it1, int32_t n_centers_total, int32_t sweeps_per_solve,
                             int32_t assume_finite, void *ws, size_t ws_bytes, int32_t *status_out,
                             void *stream_) {
    hipStream_t s = static_cast<hipStream_t>(stream_);
    GEO_REQUIRE(indptr && indices && centers && is_center && dmin && argmin && ws && status_out,
                "geo_kpp_chain: null pointer");
    GEO_REQUIRE((size_t)((n + SCAN_TILE - 1) / SCAN_TILE + 1) * 8 <= 64 * 1024, "geo_kpp_chain: n too large for the pick kernel");
    GEO_REQUIRE(n > 0 && 0 <= it0 && it0 <= it1 && it1 <= n_centers_total, "geo_kpp_chain: bad iteration range");
    GEO_REQUIRE(((sweeps_per_solve >= 2 && sweeps_per_solve < 4096) ||
                 ((sweeps_per_solve == 0 || sweeps_per_solve == -1) && assume_finite && n_centers_total <= n)) &&
                    it1 - it0 < 250000,
                "geo_kpp_chain: sweeps_per_solve out of range (0 = step kernel, -1 = resident workgroup: both need "
                "assume_finite and K <= n)");
    GEO_REQUIRE(sweeps_per_solve != -1 || n <= PG_MAX_NODES, "geo_kpp_chain: n=%d exceeds the resident chain's %d nodes", n,
                PG_MAX_NODES);
    GEO_REQUIRE(it1 - it0 <= 1 || u_host, "geo_kpp_chain: uniform deviates missing");
    KppWs w;
    if (!carve(ws, ws_bytes, n, &w)) {
        geo::set_error("geo_kpp_chain: workspace %zu too small", ws_bytes);
        return GEO_E_WORKSPACE;
    }
    // numpy's reduction tree for an array of n float32: chunks of NP_BUFSIZE, pairwise inside
    PwPlan pp;
    std::vector<int32_t> roots_raw;
    for (int32_t c0 = 0; c0 < n; c0 += NP_BUFSIZE) {
        int lv;
        roots_raw.push_back(pw_build(pp, c0, (n - c0 < NP_BUFSIZE) ? n - c0 : NP_BUFSIZE, &lv));
    }
    const int L = (int)pp.leaf_start.size(), M = (int)pp.node_l.size();
    int max_level = 0;
    for (int lv : pp.node_level) max_level = lv > max_level ? lv : max_level;
    // order nodes by level (children always sit on a lower level); remap ids to val[] positions
    std::vector<int32_t> order(M), newpos(M), level_off(max_level + 2, 0);
    for (int j = 0; j < M; ++j) level_off[pp.node_level[j]]++;          // counts at [level], levels start at 1
    {
        int run = 0;
        for (int lv = 1; lv <= max_level; ++lv) { const int c = level_off[lv]; level_off[lv - 1] = run; run += c; }
        level_off[max_level] = run;
    }
    {
        std::vector<int32_t> cursor(level_off.begin(), level_off.end());
        for (int j = 0; j < M; ++j) { const int lv = pp.node_level[j] - 1; newpos[j] = cursor[lv]++; order[newpos[j]] = j; }
    }
    auto fix = [&](int32_t id) { return id >= 0 ? id : L + newpos[-id - 1]; };
    std::vector<int32_t> blob;
    blob.reserve(2 * L + 2 * M + max_level + 2 + roots_raw.size());
    blob.insert(blob.end(), pp.leaf_start.begin(), pp.leaf_start.end());
    blob.insert(blob.end(), pp.leaf_len.begin(), pp.leaf_len.end());
    for (int k = 0; k < M; ++k) blob.push_back(fix(pp.node_l[order[k]]));
    for (int k = 0; k < M; ++k) blob.push_back(fix(pp.node_r[order[k]]));
    for (int lv = 0; lv <= max_level; ++lv) blob.push_back(level_off[lv]);
    for (int32_t r : roots_raw) blob.push_back(fix(r));
    // the tree of one buffer in local ids (kpp_sum_body, large n): a full buffer and the last one; then one ticket per buffer
    const size_t cplan_at = blob.size();
    for (int kind = 0; kind < 2; ++kind) {
        const int32_t len = kind == 0 ? (n < NP_BUFSIZE ? n : NP_BUFSIZE) : n - ((int32_t)roots_raw.size() - 1) * NP_BUFSIZE;
        PwPlan lp;
        int lv_root;
        const int32_t root = pw_build(lp, 0, len, &lv_root);
        const int cl = (int)lp.leaf_start.size(), cn = (int)lp.node_l.size();
        GEO_REQUIRE(cl <= CPLAN_MAX_LEAVES && cn < CPLAN_MAX_LEAVES && lv_root <= 11, "geo_kpp_chain: buffer tree too large (%d leaves)", cl);
        std::vector<int32_t> loff(lv_root + 2, 0), lpos(cn);
        for (int j = 0; j < cn; ++j) loff[lp.node_level[j]]++;
        {
            int run = 0;
            for (int lv = 1; lv <= lv_root; ++lv) { const int c = loff[lv]; loff[lv - 1] = run; run += c; }
            loff[lv_root] = run;
            std::vector<int32_t> cursor(loff.begin(), loff.end());
            for (int j = 0; j < cn; ++j) lpos[j] = cursor[lp.node_level[j] - 1]++;
        }
        auto lfix = [&](int32_t id) { return id >= 0 ? id : cl + lpos[-id - 1]; };
        std::vector<int32_t> cpv(CPLAN_INTS, 0);
        cpv[0] = cl; cpv[1] = cn; cpv[2] = lv_root; cpv[3] = lfix(root);
        for (int lv = 0; lv <= lv_root; ++lv) cpv[4 + lv] = loff[lv];
        for (int j = 0; j < cn; ++j) { cpv[16 + lpos[j]] = lfix(lp.node_l[j]); cpv[16 + CPLAN_MAX_LEAVES + lpos[j]] = lfix(lp.node_r[j]); }
        blob.insert(blob.end(), cpv.begin(), cpv.end());
    }
    const size_t ticket_at = blob.size();
    blob.insert(blob.end(), roots_raw.size(), 0);
    GEO_REQUIRE(blob.size() <= w.plan_ints && (size_t)(L + M) + roots_raw.size() <= w.plan_ints, "geo_kpp_chain: reduction plan overflow");
    GEO_HIP_CHECK(hipMemcpyAsync(w.plan_blob, blob.data(), blob.size() * sizeof(int32_t), hipMemcpyHostToDevice, s));
    DevPlan &dp = w.plan;
    dp.n_leaves = L; dp.n_nodes = M; dp.n_levels = max_level; dp.n_chunks = (int)roots_raw.size();
    dp.leaf_start = w.plan_blob; dp.leaf_len = dp.leaf_start + L; dp.node_l = dp.leaf_len + L;
    dp.node_r = dp.node_l + M; dp.level_off = dp.node_r + M; dp.chunk_root = dp.level_off + max_level + 1;

    KppCtl h0;
    memset(&h0, 0, sizeof(h0));
    h0.abort_iter = -1; h0.maxf = -1.f;
    GEO_HIP_CHECK(hipMemcpyAsync(w.ctl, &h0, sizeof(KppCtl), hipMemcpyHostToDevice, s));
    GEO_HIP_CHECK(hipMemsetAsync(w.mark, 0, (size_t)n * 4, s));

    const int g_lin = geo::grid_for(n, 256, 2048);
    // frontier sweeps: the first solves cross the whole graph, later ones only the new centre's (pruned)
    // cell, where a small grid keeps the launch itself cheap
    const int g_push_big = geo::grid_for(n, 32, 2048), g_push_small = geo::grid_for(n, 32, geo::options().kpp_grid);
    const int n_tiles = (n + SCAN_TILE - 1) / SCAN_TILE;
    const double tol = ((double)n + 16.0) * 4.440892098500626e-16;            // (n+16) * 2^-51
    const int exact_max = assume_finite ? 0 : 1;     // with every d_min finite no maximum is needed for the draw
    kpp_fill_inf_kernel<<<g_lin, 256, 0, s>>>(w.d, n);
    kpp_max_kernel<<<FINISH_GRID, 256, 0, s>>>(w.ctl, dmin, n, w.part_max, w.part_inf);
    kpp_maxfin_kernel<<<1, 64, 0, s>>>(w.ctl, w.part_max, w.part_inf, FINISH_GRID);
    if (it0 < it1 && sweeps_per_solve > 0) kpp_begin_kernel<<<1, 64, 0, s>>>(w.ctl, centers, it0, w.d, w.front[0]);
    GEO_LAUNCH_CHECK();
    SumPlan spl;
    spl.leaf_start = dp.leaf_start; spl.leaf_len = dp.leaf_len; spl.node_l = dp.node_l; spl.node_r = dp.node_r;
    spl.level_off = dp.level_off; spl.chunk_root = dp.chunk_root;
    spl.n_leaves = dp.n_leaves; spl.n_levels = dp.n_levels; spl.n_chunks = dp.n_chunks; spl.val = dp.val;
    spl.n_nodes = M;
    spl.cplan = w.plan_blob + cplan_at; spl.chunk_ticket = w.plan_blob + ticket_at; spl.chunk_val = dp.val + L + M;
    const bool have_u = n_centers_total > 1 && u_host;
    if (sweeps_per_solve <= 0 && have_u)
        GEO_HIP_CHECK(hipMemcpyAsync(w.u_dev, u_host, (size_t)(n_centers_total - 1) * sizeof(double), hipMemcpyHostToDevice, s));
    // ---- step-kernel mode for iterations [ia, ib): one kernel, launched until the chain reports DONE ----
    int32_t stamp_next = 1;
    auto run_steps = [&](int32_t ia, int32_t ib, KppState *hs_out, KppCtl *hc_out) -> int {
        kpp_begin_kernel<<<1, 64, 0, s>>>(w.ctl, centers, ia, w.d, w.front[0]);
        KppState st0;
        st0.mode = 0; st0.t = ia; st0.sw = 0; st0.stamp = stamp_next; st0.launches = 0; st0.pad = 0;
        GEO_HIP_CHECK(hipMemcpyAsync(w.state, &st0, sizeof(KppState), hipMemcpyHostToDevice, s));
        const int nsum = (dp.n_leaves + SUM_LEAVES - 1) / SUM_LEAVES;
        int g_small = 256;
        if (nsum > g_small) g_small = nsum;
        if (n_tiles > g_small) g_small = n_tiles;
        const size_t smem = (size_t)(n_tiles + 1) * sizeof(double);
        int64_t launched = 0;
        int32_t batch = ib - ia == 1 ? 12 : 64, t_now = ia;
        KppState hs = st0;
        KppCtl hc;
        for (;;) {
            const int grid = t_now < 16 && g_push_big > g_small ? g_push_big : g_small;
            for (int32_t i = 0; i < batch; ++i, ++launched) {
                const int parity = (int)(launched % 6);
                if (weights)
                    kpp_step_kernel<true><<<grid, 256, smem, s>>>(w.ctl, w.state, parity, indptr, indices, weights, n, w.d, dmin,
                                                                  argmin, w.mark, w.front[0], w.front[1], centers, is_center,
                                                                  w.probs, w.cdf, w.tile_sum, n_tiles, w.u_dev, tol, spl, ib,
                                                                  n_centers_total);
                else
                    kpp_step_kernel<false><<<grid, 256, smem, s>>>(w.ctl, w.state, parity, indptr, indices, weights, n, w.d, dmin,
                                                                   argmin, w.mark, w.front[0], w.front[1], centers, is_center,
                                                                   w.probs, w.cdf, w.tile_sum, n_tiles, w.u_dev, tol, spl, ib,
                                                                   n_centers_total);
            }
            GEO_LAUNCH_CHECK();
            GEO_HIP_CHECK(hipMemcpyAsync(&hs, w.state + (launched & 1), sizeof(KppState), hipMemcpyDeviceToHost, s));
            GEO_HIP_CHECK(hipMemcpyAsync(&hc, w.ctl, sizeof(KppCtl), hipMemcpyDeviceToHost, s));
            GEO_HIP_CHECK(hipStreamSynchronize(s));
            if (hs.mode == 2 || hc.abort_iter >= 0) break;
            // launches still needed ~ centres left x launches per centre so far (+10 %), at most 512 per round trip
            const int32_t done_centres = hs.t - ia > 0 ? hs.t - ia : 1;
            const double per_centre = (double)hs.launches / done_centres;
            const double est = (double)(ib - hs.t) * per_centre * 1.1 + 8.0;
            batch = est > 512.0 ? 512 : (int32_t)est;      // (launches enqueued behind a declined draw run empty: ~13 us each)
            t_now = hs.t;
            GEO_REQUIRE(launched < ((int64_t)1 << 31), "geo_kpp_chain: step kernel did not finish");
        }
        stamp_next = hs.stamp + 1;
        *hs_out = hs;
        *hc_out = hc;
        return GEO_OK;
    };
    if (sweeps_per_solve == -1 && it0 < it1) {
        // ---- resident mode: ONE workgroup runs the iterations in a single launch; a centre whose cell outgrows the
        //      workgroup's table (abort reason 4, nothing applied) is run by the step kernel right here, then the
        //      resident kernel takes over again ----
        GEO_REQUIRE(dp.n_leaves <= PG_MAX_LEAVES && M <= PG_MAX_LEAVES && dp.n_chunks <= 64 && dp.n_levels < 31,
                    "geo_kpp_chain: reduction tree of n=%d too large for the resident chain", n);
        ResidentPlan rp;
        rp.leaf_start = dp.leaf_start; rp.leaf_len = dp.leaf_len; rp.node_l = dp.node_l; rp.node_r = dp.node_r;
        rp.level_off = dp.level_off; rp.chunk_root = dp.chunk_root; rp.tail_leaf = w.tail_leaf;
        rp.n_leaves = dp.n_leaves; rp.n_nodes = M; rp.n_levels = dp.n_levels; rp.n_chunks = dp.n_chunks;
        rp.tail_start = (n / NP_BUFSIZE) * NP_BUFSIZE;
        rp.tail_first_leaf = (n / NP_BUFSIZE) * (NP_BUFSIZE / PW_BLOCK);
        std::vector<uint16_t> tail(NP_BUFSIZE, 0);
        for (int l = rp.tail_first_leaf; l < L; ++l)
            for (int32_t v = pp.leaf_start[l]; v < pp.leaf_start[l] + pp.leaf_len[l]; ++v) tail[v - rp.tail_start] = (uint16_t)l;
        GEO_HIP_CHECK(hipMemcpyAsync(w.tail_leaf, tail.data(), NP_BUFSIZE * sizeof(uint16_t), hipMemcpyHostToDevice, s));
        const bool profile = geo::options().kpp_profile != 0;                 // diagnostic runs only
        unsigned long long *prof = profile ? reinterpret_cast<unsigned long long *>(w.cdf) : nullptr;   // cdf[] is free meanwhile
        int32_t cur_it = it0, handed_back = 0;
        KppCtl hc;
        for (;;) {
            int32_t prog0 = cur_it;
            GEO_HIP_CHECK(hipMemcpyAsync(w.progress, &prog0, sizeof(int32_t), hipMemcpyHostToDevice, s));
            if (weights)
                kpp_resident_kernel<true><<<1, PG_THREADS, 0, s>>>(w.ctl, indptr, indices, weights, n, dmin, argmin, centers,
                                                                 is_center, w.u_dev, tol, rp, cur_it, it1, n_centers_total, w.progress, prof);
            else
                kpp_resident_kernel<false><<<1, PG_THREADS, 0, s>>>(w.ctl, indptr, indices, weights, n, dmin, argmin, centers,
                                                                  is_center, w.u_dev, tol, rp, cur_it, it1, n_centers_total, w.progress, prof);
            GEO_LAUNCH_CHECK();
            if (profile) {
                unsigned long long hp[14];
                GEO_HIP_CHECK(hipMemcpyAsync(hp, prof, sizeof(hp), hipMemcpyDeviceToHost, s));
                GEO_HIP_CHECK(hipStreamSynchronize(s));
                fprintf(stderr, "[kpp-resident] it %d..%d cycles: solve-rest %llu (init %llu marks %llu fill %llu pass-rest %llu loads %llu relax %llu) apply %llu leaf %llu locate %llu exact-draws %llu (%llu taken) | sweeps %llu frontier nodes %llu\n",
                        cur_it, it1, hp[0], hp[8], hp[9], hp[10], hp[11], hp[12], hp[13], hp[1], hp[2], hp[3], hp[4], hp[5], hp[6], hp[7]);
            }
            GEO_HIP_CHECK(hipMemcpyAsync(&hc, w.ctl, sizeof(KppCtl), hipMemcpyDeviceToHost, s));
            GEO_HIP_CHECK(hipStreamSynchronize(s));
            if (hc.abort_iter < 0 || hc.abort_reason != 4) break;
            const int32_t t = hc.abort_iter;
            ++handed_back;
            KppCtl hr = hc;                                  // the margin (maxf) of the call stays
            hr.abort_iter = -1; hr.abort_reason = 0;
            memset(hr.fcount, 0, sizeof(hr.fcount));
            memset(hr.ticket, 0, sizeof(hr.ticket));
            GEO_HIP_CHECK(hipMemcpyAsync(w.ctl, &hr, sizeof(KppCtl), hipMemcpyHostToDevice, s));
            KppState hs;
            if (int rc = run_steps(t, t + 1, &hs, &hc)) return rc;
            if (hc.abort_iter >= 0) break;                 // the step kernel declined too (reasons 1-3): the caller's turn
            cur_it = t + 1;
            if (cur_it >= it1) break;
        }
        status_out[0] = hc.abort_iter;
        status_out[1] = hc.abort_reason;
        status_out[2] = 0;                           // resident mode is only entered with d_min finite everywhere
        status_out[3] = handed_back;                 // centres run by the step kernel
        return GEO_OK;
    }
    if (sweeps_per_solve == 0 && it0 < it1) {
        KppState hs;
        KppCtl hc;
        if (int rc = run_steps(it0, it1, &hs, &hc)) return rc;
        status_out[0] = hc.abort_iter;
        status_out[1] = hc.abort_reason;
        status_out[2] = hc.n_inf;
        status_out[3] = (int32_t)(hs.launches);
        return GEO_OK;
    }
    for (int32_t t = it0; t < it1; ++t) {
        const int32_t stamp_solve = (t - it0) + 1;
        int last_next = 0;
        const int g_push = t < 16 ? g_push_big : g_push_small;
        for (int sw = 0; sw < sweeps_per_solve; ++sw) {
            const int cur = sw % 3, next = (sw + 1) % 3, clear = (sw + 2) % 3;
            if (weights)
                kpp_push_kernel<true><<<g_push, 256, 0, s>>>(w.ctl, indptr, indices, weights, w.d, dmin, w.mark,
                                                             w.front[sw & 1], w.front[(sw + 1) & 1], cur, next, clear,
                                                             stamp_solve, sw);
            else
                kpp_push_kernel<false><<<g_push, 256, 0, s>>>(w.ctl, indptr, indices, weights, w.d, dmin, w.mark,
                                                              w.front[sw & 1], w.front[(sw + 1) & 1], cur, next, clear,
                                                              stamp_solve, sw);
            last_next = next;
        }
        const bool draw = t + 1 < n_centers_total;
        const int fuse = (!exact_max && draw) ? 1 : 0;
        if (!fuse)
            kpp_finish_kernel<<<geo::grid_for(n, 256, 256), 256, 0, s>>>(w.ctl, w.d, dmin, argmin, n, last_next, t);
        if (draw) {
            if (exact_max) kpp_max_kernel<<<FINISH_GRID, 256, 0, s>>>(w.ctl, dmin, n, w.part_max, w.part_inf);
            kpp_sum_kernel<<<(dp.n_leaves + SUM_LEAVES - 1) / SUM_LEAVES, 256, 0, s>>>(
                w.ctl, dmin, argmin, w.d, fuse, last_next, t, is_center, w.part_max, w.part_inf, FINISH_GRID, exact_max,
                w.probs, spl);
            kpp_draw_kernel<<<n_tiles, SCAN_T, (size_t)(n_tiles + 1) * sizeof(double), s>>>(
                w.ctl, w.probs, n, w.cdf, w.tile_sum, n_tiles, u_host[t], tol, centers, is_center, t + 1, t,
                t + 1 < it1 ? 1 : 0, w.d, w.front[0]);
        }
        GEO_LAUNCH_CHECK();
    }
    KppCtl h1;
    GEO_HIP_CHECK(hipMemcpyAsync(&h1, w.ctl, sizeof(KppCtl), hipMemcpyDeviceToHost, s));
    GEO_HIP_CHECK(hipStreamSynchronize(s));
    status_out[0] = h1.abort_iter;
    status_out[1] = h1.abort_reason;
    status_out[2] = h1.n_inf;                 // unreachable entries of d_min at the last maximum pass
    status_out[3] = h1.max_sw;                // sweeps the longest solve of this call used
    return GEO_OK;
}

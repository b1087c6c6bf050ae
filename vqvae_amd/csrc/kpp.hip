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

#include <vector>

namespace {

constexpr int NP_BUFSIZE = 8192;      // numpy's default ufunc buffer size (elements per reduction chunk)
constexpr int PW_BLOCK = 128;         // numpy pairwise-sum leaf size
constexpr int SCAN_T = 256, SCAN_I = 8, SCAN_TILE = SCAN_T * SCAN_I;
constexpr int FINISH_GRID = 256;

struct KppCtl {
    int32_t abort_iter, abort_reason;      // -1 / 0 while healthy; reason 1 solve, 2 margin, 3 degenerate
    int32_t n_touched;                     // nodes whose distance was lowered by the current solve
    int32_t found, pick_idx, pick_ok;
    int32_t fcount[3];                     // frontier sizes, ring over sweeps
    float total;
    float maxf;                            // max finite d_min (-1: none), basis of the pruning margin
    int32_t n_inf;                         // unreachable (inf) entries of d_min at the last max pass
    double s_last;
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
// single stamp per node: solve_base = stamp_solve * 8192; within a solve, sweep sw uses base = solve_base + 2*sw
// for "distance lowered" and base + 1 for "queued for the next sweep"
__device__ __forceinline__ void kpp_begin(KppCtl *ctl, const int32_t *centers, int32_t pos, double *d, int32_t *mark,
                                          int32_t *touched, int32_t *front0, int32_t stamp_solve) {
    const int32_t src = centers[pos];
    d[src] = 0.0;
    mark[src] = stamp_solve * 8192;
    touched[0] = src;
    front0[0] = src;
    ctl->n_touched = 1;
    ctl->fcount[0] = 1;
    ctl->fcount[1] = 0;
    ctl->fcount[2] = 0;
}

__global__ void kpp_begin_kernel(KppCtl *ctl, const int32_t *__restrict__ centers, int32_t pos, double *d,
                                 int32_t *mark, int32_t *touched, int32_t *front0, int32_t stamp_solve) {
    if (ctl->abort_iter >= 0 || threadIdx.x != 0 || blockIdx.x != 0) return;
    kpp_begin(ctl, centers, pos, d, mark, touched, front0, stamp_solve);
}

// Expand one frontier node u (32 lanes share its adjacency row).  `du` must be a fresh (L2) read.
template <bool WEIGHTED, typename AppendTouched, typename AppendFront>
__device__ __forceinline__ void kpp_expand(const int32_t *__restrict__ indptr, const int32_t *__restrict__ indices,
                                           const float *__restrict__ weights, unsigned long long *dbits,
                                           const float *__restrict__ dmin, int32_t *mark, int32_t u, double du, double tau,
                                           int sub, int32_t solve_base, int32_t base, AppendTouched add_touched,
                                           AppendFront add_front) {
    const int32_t e0 = indptr[u], e1 = indptr[u + 1];
    for (int32_t e = e0 + sub; e < e1; e += 32) {
        const int32_t v = indices[e];
        const double cand = du + (WEIGHTED ? (double)weights[e] : 1.0);
        const unsigned long long cb = (unsigned long long)__double_as_longlong(cand);
        if (cb >= dbits[v]) continue;                   // cheap pre-test (monotone: values only decrease)
        const unsigned long long old = atomicMin(&dbits[v], cb);
        if (cb < old) {
            const bool open = cand <= (double)dmin[v] + tau;          // not pruned: goes to the next frontier
            const int32_t before = atomicMax(&mark[v], open ? base + 1 : base);
            if (before < solve_base) add_touched(v);                  // first time lowered in this solve
            if (open && before < base + 1) add_front(v);              // first time queued in this sweep
        }
    }
}

template <bool WEIGHTED>
__global__ __launch_bounds__(256) void kpp_push_kernel(KppCtl *ctl, const int32_t *__restrict__ indptr,
                                                      const int32_t *__restrict__ indices,
                                                      const float *__restrict__ weights, double *d,
                                                      const float *__restrict__ dmin, int32_t *mark,
                                                      int32_t *touched, const int32_t *__restrict__ fin,
                                                      int32_t *__restrict__ fout, int cur, int next, int clear,
                                                      int32_t stamp_solve, int32_t sw) {
    if (ctl->abort_iter >= 0) return;
    if (blockIdx.x == 0 && threadIdx.x == 0) ctl->fcount[clear] = 0;
    const int32_t cnt = ctl->fcount[cur];
    if (cnt == 0) return;
    const double tau = ctl->maxf > 0.f ? 1e-6 * (double)ctl->maxf : 0.0;
    const int sub = threadIdx.x & 31;                       // 32 lanes share one frontier node (mean degree ~31)
    const int grp = (blockIdx.x * blockDim.x + threadIdx.x) >> 5;
    const int ngrp = (gridDim.x * blockDim.x) >> 5;
    const int32_t solve_base = stamp_solve * 8192, base = solve_base + 2 * sw;
    unsigned long long *dbits = reinterpret_cast<unsigned long long *>(d);
    for (int32_t i = grp; i < cnt; i += ngrp) {
        const int32_t u = fin[i];
        const double du = d[u];                             // fresh: written before this launch
        if (du > (double)dmin[u] + tau) continue;           // pruned: cannot improve anything behind it
        kpp_expand<WEIGHTED>(indptr, indices, weights, dbits, dmin, mark, u, du, tau, sub, solve_base, base,
                             [&](int32_t v) { touched[atomicAdd(&ctl->n_touched, 1)] = v; },
                             [&](int32_t v) { fout[atomicAdd(&ctl->fcount[next], 1)] = v; });
    }
}

// The same solve inside ONE workgroup: frontier queues in LDS, sweeps separated by __syncthreads instead of
// kernel launches.  Used once the cells are small (after the warm-up centres); a frontier that outgrows the
// queue aborts the iteration (reason 4), which the caller redoes with the multi-launch sweeps above.
constexpr int MICRO_QCAP = 8192;
template <bool WEIGHTED>
__global__ __launch_bounds__(1024) void kpp_micro_kernel(KppCtl *ctl, const int32_t *__restrict__ indptr,
                                                        const int32_t *__restrict__ indices,
                                                        const float *__restrict__ weights, double *d,
                                                        const float *__restrict__ dmin, int32_t *mark,
                                                        int32_t *touched, const int32_t *__restrict__ centers,
                                                        int32_t pos, int32_t stamp_solve, int32_t max_sweeps) {
    if (ctl->abort_iter >= 0) return;
    __shared__ int32_t q[2][MICRO_QCAP];
    __shared__ int32_t qcnt[2], n_touched, overflow;
    const double tau = ctl->maxf > 0.f ? 1e-6 * (double)ctl->maxf : 0.0;
    const int sub = threadIdx.x & 31, grp = threadIdx.x >> 5, ngrp = blockDim.x >> 5;
    const int32_t solve_base = stamp_solve * 8192;
    unsigned long long *dbits = reinterpret_cast<unsigned long long *>(d);
    if (threadIdx.x == 0) {
        const int32_t src = centers[pos];
        d[src] = 0.0;
        mark[src] = solve_base;
        touched[0] = src;
        q[0][0] = src;
        qcnt[0] = 1; qcnt[1] = 0; n_touched = 1; overflow = 0;
    }
    __syncthreads();
    int sw = 0;
    for (; sw < max_sweeps; ++sw) {
        const int cur = sw & 1, nxt = cur ^ 1;
        const int32_t cnt = qcnt[cur];
        if (cnt == 0 || overflow) break;
        const int32_t base = solve_base + 2 * sw;
        for (int32_t i = grp; i < cnt; i += ngrp) {
            const int32_t u = q[cur][i];
            // L2 read: another wave of this workgroup may have lowered d[u] during the previous sweep and this
            // CU's L1 is not refreshed inside a kernel
            const double du = __longlong_as_double((long long)__hip_atomic_load(&dbits[u], __ATOMIC_RELAXED,
                                                                               __HIP_MEMORY_SCOPE_AGENT));
            if (du > (double)dmin[u] + tau) continue;
            kpp_expand<WEIGHTED>(indptr, indices, weights, dbits, dmin, mark, u, du, tau, sub, solve_base, base,
                                 [&](int32_t v) { touched[atomicAdd(&n_touched, 1)] = v; },
                                 [&](int32_t v) {
                                     const int32_t p = atomicAdd(&qcnt[nxt], 1);
                                     if (p < MICRO_QCAP) q[nxt][p] = v; else overflow = 1;
                                 });
        }
        __syncthreads();
        if (threadIdx.x == 0) qcnt[cur] = 0;
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        ctl->n_touched = n_touched;
        ctl->fcount[0] = 0; ctl->fcount[1] = 0; ctl->fcount[2] = 0;
        if (overflow) { ctl->abort_iter = pos; ctl->abort_reason = 4; }
        else if (qcnt[sw & 1] != 0) { ctl->abort_iter = pos; ctl->abort_reason = 1; }
    }
}

// d_min / argmin update over the touched nodes (kmeans_optimized.py:44 + the single-pass assignment);
// resets their distances for the next solve.  Aborts (nothing applied) when the frontier is not empty.
__global__ __launch_bounds__(256) void kpp_finish_kernel(KppCtl *ctl, double *d, float *__restrict__ dmin,
                                                        int32_t *__restrict__ argmin, const int32_t *__restrict__ touched,
                                                        int last_next, int32_t pos) {
    if (ctl->abort_iter >= 0) return;
    if (ctl->fcount[last_next] != 0) {
        if (blockIdx.x == 0 && threadIdx.x == 0) { ctl->abort_iter = pos; ctl->abort_reason = 1; }
        return;                                             // every block sees the same final count
    }
    const int32_t nt = ctl->n_touched;
    for (int32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < nt; i += gridDim.x * blockDim.x) {
        const int32_t v = touched[i];
        const float x = (float)d[v];
        if (x < dmin[v]) {
            dmin[v] = x;
            argmin[v] = pos;
        }
        d[v] = inf64();
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

// probs = d_safe**2 with probs[centres] = 0 (kmeans_optimized.py:47-57), written for the cdf pass, and stage 1 of
// numpy's float32 add.reduce over it (see file header): leaf sums.  8 lanes own the 8 strided accumulators
// r[0..7] of one <=128-element leaf; they are combined as ((r0+r1)+(r2+r3))+((r4+r5)+(r6+r7)).
// `exact_max`: the per-block maxima of kpp_max_kernel are reduced here (needed when d_min still holds inf
// entries, which are replaced by 2*max_finite); otherwise every entry is finite and no maximum is needed.
// `fuse_finish`: the d_min / argmin update of the solve just finished (and the reset of its distances) is done
// here, element by element, instead of by kpp_finish_kernel -- every node is visited exactly once by the leaves.
__global__ __launch_bounds__(256) void kpp_leaf_kernel(KppCtl *ctl, float *__restrict__ dmin,
                                                      int32_t *__restrict__ argmin, double *__restrict__ d,
                                                      int fuse_finish, int last_next, int32_t pos,
                                                      const uint8_t *__restrict__ is_center,
                                                      const float *__restrict__ part_max,
                                                      const int32_t *__restrict__ part_inf, int n_part, int exact_max,
                                                      float *__restrict__ probs,
                                                      const int32_t *__restrict__ leaf_start,
                                                      const int32_t *__restrict__ leaf_len, int n_leaves,
                                                      float *__restrict__ val) {
    if (ctl->abort_iter >= 0) return;
    if (fuse_finish && ctl->fcount[last_next] != 0) {          // the solve did not converge: apply nothing
        if (blockIdx.x == 0 && threadIdx.x == 0) { ctl->abort_iter = pos; ctl->abort_reason = 1; }
        return;
    }
    __shared__ float smax;
    __shared__ int32_t sinf;
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
        if (blockIdx.x == 0 && threadIdx.x == 0) { ctl->maxf = maxf; ctl->n_inf = sinf; }   // margin of the next solve
    }
    const float sub = maxf * 2.0f;
    const int j = threadIdx.x & 7;
    const int leaf = (blockIdx.x * blockDim.x + threadIdx.x) >> 3;
    const bool live = leaf < n_leaves;
    const int32_t i0 = live ? leaf_start[leaf] : 0;
    const int len = live ? leaf_len[leaf] : 0;
    const int m8 = len - (len % 8);
    auto prob_at = [&](int32_t i) {
        float x = dmin[i];
        if (fuse_finish) {
            const double dd = d[i];
            if (dd < inf64()) {                            // touched by the solve of centre `pos`
                const float xd = (float)dd;
                if (xd < x) { x = xd; dmin[i] = xd; argmin[i] = pos; }
                d[i] = inf64();
            }
        }
        const float safe = any_finite ? (x < inf32() ? x : sub) : 1.0f;
        const float p = is_center[i] ? 0.0f : safe * safe;
        probs[i] = p;
        return p;
    };
    float r = 0.0f;
    if (len >= 8) {
        r = prob_at(i0 + j);
        for (int i = 8; i < m8; i += 8) r += prob_at(i0 + i + j);
    }
    // lanes (0,1) (2,3) (4,5) (6,7) -> lanes 0,2,4,6 ; then (0,2) (4,6) -> 0,4 ; then (0,4) -> 0
    float o = __shfl_down(r, 1, 8);
    if ((j & 1) == 0) r += o;
    o = __shfl_down(r, 2, 8);
    if ((j & 3) == 0) r += o;
    o = __shfl_down(r, 4, 8);
    if (j == 0) {
        r += o;
        if (len < 8) {
            r = 0.0f;
            for (int i = 0; i < len; ++i) r += prob_at(i0 + i);
        } else {
            for (int i = m8; i < len; ++i) r += prob_at(i0 + i);
        }
        if (live) val[leaf] = r;
    }
}

// stage 2 (one block): the halving tree above the leaves, then the chunk roots accumulated in order.
__global__ __launch_bounds__(1024) void kpp_tree_kernel(KppCtl *ctl, int n_leaves, const int32_t *__restrict__ node_l,
                                                       const int32_t *__restrict__ node_r,
                                                       const int32_t *__restrict__ level_off, int n_levels,
                                                       const int32_t *__restrict__ chunk_root, int n_chunks,
                                                       float *__restrict__ val, int32_t iter) {
    if (ctl->abort_iter >= 0) return;
    for (int lv = 0; lv < n_levels; ++lv) {
        for (int j = level_off[lv] + threadIdx.x; j < level_off[lv + 1]; j += blockDim.x)
            val[n_leaves + j] = val[node_l[j]] + val[node_r[j]];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        float total = 0.0f;
        for (int c = 0; c < n_chunks; ++c) total += val[chunk_root[c]];
        ctl->total = total;
        ctl->found = 0;
        if (!(total > 0.0f)) { ctl->abort_iter = iter; ctl->abort_reason = 3; }
    }
}

// tile-local inclusive fp64 scan of p = float64(probs / total); tile totals
__global__ __launch_bounds__(SCAN_T) void kpp_scan_tiles_kernel(const KppCtl *ctl, const float *__restrict__ probs,
                                                               int32_t n, double *__restrict__ cdf,
                                                               double *__restrict__ tile_sum) {
    if (ctl->abort_iter >= 0) return;
    __shared__ double wtot[SCAN_T / 64];
    const float total = ctl->total;
    const int64_t base = (int64_t)blockIdx.x * SCAN_TILE + (int64_t)threadIdx.x * SCAN_I;
    double v[SCAN_I];
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
    const double excl = wbase + inc - run;
#pragma unroll
    for (int i = 0; i < SCAN_I; ++i)
        if (base + i < n) cdf[base + i] = excl + v[i];
    if (threadIdx.x == 0) tile_sum[blockIdx.x] = tot;
}

// idx = searchsorted(cdf / cdf[-1], u, side='right') with a safety margin around u.  Every block first
// rebuilds the exclusive tile offsets (sequential order, so all blocks agree bit for bit) in LDS.
__global__ __launch_bounds__(256) void kpp_pick_kernel(KppCtl *ctl, const double *__restrict__ cdf,
                                                      const double *__restrict__ tile_sum, int n_tiles, int32_t n,
                                                      double u, double tol) {
    if (ctl->abort_iter >= 0) return;
    extern __shared__ __attribute__((aligned(16))) double toff[];          // [n_tiles + 1]
    if (threadIdx.x == 0) {
        double run = 0.0;
        for (int t = 0; t < n_tiles; ++t) { toff[t] = run; run += tile_sum[t]; }
        toff[n_tiles] = run;
    }
    __syncthreads();
    const double s_last = toff[n_tiles];
    for (int32_t j = blockIdx.x * blockDim.x + threadIdx.x; j < n; j += gridDim.x * blockDim.x) {
        const double cj = (cdf[j] + toff[j / SCAN_TILE]) / s_last;
        if (j == 0 && cj > u) {                       // nothing <= u: index 0
            ctl->found = 1; ctl->pick_idx = 0; ctl->pick_ok = (cj - u > tol) ? 1 : 0;
        }
        if (cj <= u) {
            const bool last = j == n - 1;
            const double cn = last ? inf64() : (cdf[j + 1] + toff[(j + 1) / SCAN_TILE]) / s_last;
            if (cn > u) {
                ctl->found = 1; ctl->pick_idx = j + 1;
                ctl->pick_ok = (!last && (u - cj > tol) && (cn - u > tol)) ? 1 : 0;
            }
        }
    }
}

// accept or decline the pick; on success also open the next solve (when it runs with the multi-launch sweeps)
__global__ void kpp_commit_kernel(KppCtl *ctl, int32_t *centers, uint8_t *is_center, int32_t next_pos, int32_t iter,
                                  int begin_next, double *d, int32_t *mark, int32_t *touched, int32_t *front0,
                                  int32_t next_stamp) {
    if (ctl->abort_iter >= 0) return;
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    if (!ctl->found || !ctl->pick_ok) { ctl->abort_iter = iter; ctl->abort_reason = 2; return; }
    centers[next_pos] = ctl->pick_idx;
    is_center[ctl->pick_idx] = 1;
    if (begin_next) kpp_begin(ctl, centers, next_pos, d, mark, touched, front0, next_stamp);
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
    double *d, *cdf, *tile_sum, *tile_off;
    int32_t *mark, *touched, *front[2], *part_inf;
    float *probs, *part_max;
    DevPlan plan;
    int32_t *plan_blob;
    size_t plan_ints;
};

size_t plan_ints_bound(int32_t n) {
    const size_t leaves = (size_t)n / 64 + (size_t)n / NP_BUFSIZE + 16;      // leaves have >= 64 elements except per chunk tails
    return 6 * leaves + 64;
}

bool carve(void *ws, size_t ws_bytes, int32_t n, KppWs *o) {
    geo::Arena ar(ws, ws_bytes);
    o->ctl = ar.take<KppCtl>(4);
    o->d = ar.take<double>((size_t)n);
    o->cdf = ar.take<double>((size_t)n);
    const size_t tiles = ((size_t)n + SCAN_TILE - 1) / SCAN_TILE;
    o->tile_sum = ar.take<double>(tiles + 1);
    o->tile_off = ar.take<double>(tiles + 1);
    o->mark = ar.take<int32_t>((size_t)n);
    o->touched = ar.take<int32_t>((size_t)n);
    o->front[0] = ar.take<int32_t>((size_t)n);
    o->front[1] = ar.take<int32_t>((size_t)n);
    o->probs = ar.take<float>((size_t)n);
    o->part_max = ar.take<float>(FINISH_GRID);
    o->part_inf = ar.take<int32_t>(FINISH_GRID);
    o->plan_ints = plan_ints_bound(n);
    o->plan_blob = ar.take<int32_t>(o->plan_ints);
    o->plan.val = ar.take<float>(o->plan_ints);
    return o->plan.val != nullptr;
}

}  // namespace

extern "C" size_t geo_kpp_workspace_bytes(int32_t n) {
    if (n <= 0) return 4096;
    const size_t tiles = ((size_t)n + SCAN_TILE - 1) / SCAN_TILE;
    return geo::align_up(4 * sizeof(KppCtl)) + 2 * geo::align_up((size_t)n * 8) + 2 * geo::align_up((tiles + 1) * 8) +
           6 * geo::align_up((size_t)n * 4) + 2 * geo::align_up(FINISH_GRID * 4) +
           2 * geo::align_up(plan_ints_bound(n) * 4) + 4096;
}

extern "C" int geo_kpp_chain(const int32_t *indptr, const int32_t *indices, const float *weights, int32_t n,
                             int32_t *centers, uint8_t *is_center, float *dmin, int32_t *argmin, const double *u_host,
                             int32_t it0, int32_t it1, int32_t n_centers_total, int32_t sweeps_per_solve,
                             int32_t micro, int32_t assume_finite, void *ws, size_t ws_bytes, int32_t *status_out,
                             void *stream_) {
    hipStream_t s = static_cast<hipStream_t>(stream_);
    GEO_REQUIRE(indptr && indices && centers && is_center && dmin && argmin && ws && status_out,
                "geo_kpp_chain: null pointer");
    GEO_REQUIRE((size_t)((n + SCAN_TILE - 1) / SCAN_TILE + 1) * 8 <= 64 * 1024, "geo_kpp_chain: n too large for the pick kernel");
    GEO_REQUIRE(n > 0 && 0 <= it0 && it0 <= it1 && it1 <= n_centers_total, "geo_kpp_chain: bad iteration range");
    GEO_REQUIRE(sweeps_per_solve >= 2 && sweeps_per_solve < 4096 && it1 - it0 < 250000, "geo_kpp_chain: sweeps_per_solve out of range");
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
    GEO_REQUIRE(blob.size() <= w.plan_ints && (size_t)(L + M) <= w.plan_ints, "geo_kpp_chain: reduction plan overflow");
    GEO_HIP_CHECK(hipMemcpyAsync(w.plan_blob, blob.data(), blob.size() * sizeof(int32_t), hipMemcpyHostToDevice, s));
    DevPlan &dp = w.plan;
    dp.n_leaves = L; dp.n_nodes = M; dp.n_levels = max_level; dp.n_chunks = (int)roots_raw.size();
    dp.leaf_start = w.plan_blob; dp.leaf_len = dp.leaf_start + L; dp.node_l = dp.leaf_len + L;
    dp.node_r = dp.node_l + M; dp.level_off = dp.node_r + M; dp.chunk_root = dp.level_off + max_level + 1;

    KppCtl h0;
    h0.abort_iter = -1; h0.abort_reason = 0; h0.n_touched = 0; h0.found = 0; h0.pick_idx = -1; h0.pick_ok = 0;
    h0.fcount[0] = h0.fcount[1] = h0.fcount[2] = 0;
    h0.total = 0.f; h0.maxf = -1.f; h0.n_inf = 0; h0.s_last = 0.0;
    GEO_HIP_CHECK(hipMemcpyAsync(w.ctl, &h0, sizeof(KppCtl), hipMemcpyHostToDevice, s));
    GEO_HIP_CHECK(hipMemsetAsync(w.mark, 0, (size_t)n * 4, s));

    const int g_lin = geo::grid_for(n, 256, 2048);
    // frontier sweeps: the first solves cross the whole graph, later ones only the new centre's (pruned)
    // cell, where a small grid keeps the launch itself cheap
    const int g_push_big = geo::grid_for(n, 32, 2048), g_push_small = geo::grid_for(n, 32, 256);
    const int n_tiles = (n + SCAN_TILE - 1) / SCAN_TILE;
    const double tol = ((double)n + 16.0) * 4.440892098500626e-16;            // (n+16) * 2^-51
    const int exact_max = assume_finite ? 0 : 1;     // with every d_min finite no maximum is needed for the draw
    kpp_fill_inf_kernel<<<g_lin, 256, 0, s>>>(w.d, n);
    kpp_max_kernel<<<FINISH_GRID, 256, 0, s>>>(w.ctl, dmin, n, w.part_max, w.part_inf);
    kpp_maxfin_kernel<<<1, 64, 0, s>>>(w.ctl, w.part_max, w.part_inf, FINISH_GRID);
    if (!micro && it0 < it1)
        kpp_begin_kernel<<<1, 64, 0, s>>>(w.ctl, centers, it0, w.d, w.mark, w.touched, w.front[0], 1);
    GEO_LAUNCH_CHECK();
    for (int32_t t = it0; t < it1; ++t) {
        const int32_t stamp_solve = (t - it0) + 1;
        int last_next = 0;
        if (micro) {
            if (weights)
                kpp_micro_kernel<true><<<1, 1024, 0, s>>>(w.ctl, indptr, indices, weights, w.d, dmin, w.mark, w.touched,
                                                          centers, t, stamp_solve, 4000);
            else
                kpp_micro_kernel<false><<<1, 1024, 0, s>>>(w.ctl, indptr, indices, weights, w.d, dmin, w.mark, w.touched,
                                                           centers, t, stamp_solve, 4000);
        } else {
            const int g_push = t < 16 ? g_push_big : g_push_small;
            for (int sw = 0; sw < sweeps_per_solve; ++sw) {
                const int cur = sw % 3, next = (sw + 1) % 3, clear = (sw + 2) % 3;
                if (weights)
                    kpp_push_kernel<true><<<g_push, 256, 0, s>>>(w.ctl, indptr, indices, weights, w.d, dmin, w.mark, w.touched,
                                                                 w.front[sw & 1], w.front[(sw + 1) & 1], cur, next, clear,
                                                                 stamp_solve, sw);
                else
                    kpp_push_kernel<false><<<g_push, 256, 0, s>>>(w.ctl, indptr, indices, weights, w.d, dmin, w.mark, w.touched,
                                                                  w.front[sw & 1], w.front[(sw + 1) & 1], cur, next, clear,
                                                                  stamp_solve, sw);
                last_next = next;
            }
        }
        const int fuse = (!exact_max && t + 1 < n_centers_total) ? 1 : 0;
        if (!fuse)
            kpp_finish_kernel<<<geo::grid_for(n, 256, 256), 256, 0, s>>>(w.ctl, w.d, dmin, argmin, w.touched, last_next, t);
        if (t + 1 < n_centers_total) {
            if (exact_max) kpp_max_kernel<<<FINISH_GRID, 256, 0, s>>>(w.ctl, dmin, n, w.part_max, w.part_inf);
            kpp_leaf_kernel<<<(dp.n_leaves * 8 + 255) / 256, 256, 0, s>>>(w.ctl, dmin, argmin, w.d, fuse, last_next, t,
                                                                          is_center, w.part_max, w.part_inf,
                                                                          FINISH_GRID, exact_max, w.probs, dp.leaf_start,
                                                                          dp.leaf_len, dp.n_leaves, dp.val);
            kpp_tree_kernel<<<1, 1024, 0, s>>>(w.ctl, dp.n_leaves, dp.node_l, dp.node_r, dp.level_off, dp.n_levels,
                                               dp.chunk_root, dp.n_chunks, dp.val, t);
            kpp_scan_tiles_kernel<<<n_tiles, SCAN_T, 0, s>>>(w.ctl, w.probs, n, w.cdf, w.tile_sum);
            kpp_pick_kernel<<<g_lin, 256, (size_t)(n_tiles + 1) * sizeof(double), s>>>(w.ctl, w.cdf, w.tile_sum, n_tiles,
                                                                                       n, u_host[t], tol);
            kpp_commit_kernel<<<1, 64, 0, s>>>(w.ctl, centers, is_center, t + 1, t, (!micro && t + 1 < it1) ? 1 : 0, w.d,
                                               w.mark, w.touched, w.front[0], stamp_solve + 1);
        }
        GEO_LAUNCH_CHECK();
    }
    KppCtl h1;
    GEO_HIP_CHECK(hipMemcpyAsync(&h1, w.ctl, sizeof(KppCtl), hipMemcpyDeviceToHost, s));
    GEO_HIP_CHECK(hipStreamSynchronize(s));
    status_out[0] = h1.abort_iter;
    status_out[1] = h1.abort_reason;
    status_out[2] = h1.n_inf;                 // unreachable entries of d_min at the last maximum pass
    return GEO_OK;
}

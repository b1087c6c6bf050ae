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
    int32_t cur_src;
    int32_t found, pick_idx, pick_ok;
    float total;
    double s_last;
};

__device__ __forceinline__ double inf64() { return __longlong_as_double(0x7ff0000000000000LL); }
__device__ __forceinline__ float inf32() { return __int_as_float(0x7f800000); }

__global__ __launch_bounds__(256) void kpp_init_kernel(KppCtl *ctl, const int32_t *__restrict__ centers, int32_t pos,
                                                      double *__restrict__ d, int32_t n, int32_t *flags) {
    if (ctl->abort_iter >= 0) return;
    const int32_t source = centers[pos];
    for (int32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x)
        d[i] = (i == source) ? 0.0 : inf64();
    if (blockIdx.x == 0 && threadIdx.x < 3) flags[threadIdx.x] = 0;
}

template <bool WEIGHTED>
__global__ __launch_bounds__(256) void kpp_sweep_kernel(const KppCtl *ctl, const int32_t *__restrict__ indptr,
                                                       const int32_t *__restrict__ indices,
                                                       const float *__restrict__ weights, int32_t n, double *d,
                                                       int32_t *flags, int prev, int cur, int next, int first) {
    if (ctl->abort_iter >= 0) return;
    if (blockIdx.x == 0 && threadIdx.x == 0) flags[next] = 0;
    if (!first && flags[prev] == 0) return;
    if (geo::sweep_single_body<WEIGHTED>(indptr, indices, weights, n, d)) flags[cur] = 1;
}

__global__ void kpp_verdict_kernel(KppCtl *ctl, const int32_t *flags, int last_cur, int32_t iter) {
    if (ctl->abort_iter >= 0) return;
    if (flags[last_cur] != 0) { ctl->abort_iter = iter; ctl->abort_reason = 1; }
}

// d_min / argmin update (kmeans_optimized.py:44 + the single-pass assignment) and per-block maxima
// of the finite d_min for the inf -> 2*max_finite rule (:47-50).
__global__ __launch_bounds__(256) void kpp_finish_kernel(const KppCtl *ctl, const double *__restrict__ d, int32_t n,
                                                        float *__restrict__ dmin, int32_t *__restrict__ argmin,
                                                        int32_t pos, float *__restrict__ part_max) {
    if (ctl->abort_iter >= 0) return;
    __shared__ float smax[4];
    float m = -1.0f;                                   // distances are >= 0: -1 means "no finite value seen"
    for (int32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const float x = (float)d[i];
        float cur = dmin[i];
        if (x < cur) {
            cur = x;
            dmin[i] = x;
            argmin[i] = pos;
        }
        if (cur < inf32()) m = fmaxf(m, cur);
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) m = fmaxf(m, __shfl_xor(m, off, 64));
    if ((threadIdx.x & 63) == 0) smax[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) part_max[blockIdx.x] = fmaxf(fmaxf(smax[0], smax[1]), fmaxf(smax[2], smax[3]));
}

__global__ __launch_bounds__(256) void kpp_probs_kernel(const KppCtl *ctl, const float *__restrict__ dmin,
                                                       const uint8_t *__restrict__ is_center, int32_t n,
                                                       const float *__restrict__ part_max, int n_part,
                                                       float *__restrict__ probs) {
    if (ctl->abort_iter >= 0) return;
    __shared__ float smax;
    if (threadIdx.x < 64) {
        float m = -1.0f;
        for (int i = threadIdx.x; i < n_part; i += 64) m = fmaxf(m, part_max[i]);
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) m = fmaxf(m, __shfl_xor(m, off, 64));
        if (threadIdx.x == 0) smax = m;
    }
    __syncthreads();
    const float maxf = smax;
    const bool any_finite = maxf >= 0.0f;
    const float sub = maxf * 2.0f;
    for (int32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const float x = dmin[i];
        const float safe = any_finite ? (x < inf32() ? x : sub) : 1.0f;
        probs[i] = is_center[i] ? 0.0f : safe * safe;
    }
}

// numpy float32 add.reduce of probs (see file header).  One block.
__global__ __launch_bounds__(1024) void kpp_total_kernel(KppCtl *ctl, const float *__restrict__ a,
                                                        const int32_t *__restrict__ leaf_start,
                                                        const int32_t *__restrict__ leaf_len, int n_leaves,
                                                        const int32_t *__restrict__ node_l,
                                                        const int32_t *__restrict__ node_r,
                                                        const int32_t *__restrict__ level_off, int n_levels,
                                                        const int32_t *__restrict__ chunk_root, int n_chunks,
                                                        float *__restrict__ val, int32_t iter) {
    if (ctl->abort_iter >= 0) return;
    for (int leaf = threadIdx.x; leaf < n_leaves; leaf += blockDim.x) {
        const float *x = a + leaf_start[leaf];
        const int len = leaf_len[leaf];
        float res;
        if (len < 8) {
            res = 0.0f;
            for (int i = 0; i < len; ++i) res += x[i];
        } else {
            float r0 = x[0], r1 = x[1], r2 = x[2], r3 = x[3], r4 = x[4], r5 = x[5], r6 = x[6], r7 = x[7];
            int i = 8;
            const int m = len - (len % 8);
            for (; i < m; i += 8) {
                r0 += x[i]; r1 += x[i + 1]; r2 += x[i + 2]; r3 += x[i + 3];
                r4 += x[i + 4]; r5 += x[i + 5]; r6 += x[i + 6]; r7 += x[i + 7];
            }
            res = ((r0 + r1) + (r2 + r3)) + ((r4 + r5) + (r6 + r7));
            for (; i < len; ++i) res += x[i];
        }
        val[leaf] = res;
    }
    __syncthreads();
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

__global__ void kpp_scan_offsets_kernel(KppCtl *ctl, const double *__restrict__ tile_sum, int n_tiles,
                                        double *__restrict__ tile_off) {
    if (ctl->abort_iter >= 0) return;
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        double run = 0.0;
        for (int t = 0; t < n_tiles; ++t) { tile_off[t] = run; run += tile_sum[t]; }
        ctl->s_last = run;
    }
}

// idx = searchsorted(cdf / cdf[-1], u, side='right') with a safety margin around u
__global__ __launch_bounds__(256) void kpp_pick_kernel(KppCtl *ctl, const double *__restrict__ cdf,
                                                      const double *__restrict__ tile_off, int32_t n, double u,
                                                      double tol) {
    if (ctl->abort_iter >= 0) return;
    const double s_last = ctl->s_last;
    for (int32_t j = blockIdx.x * blockDim.x + threadIdx.x; j < n; j += gridDim.x * blockDim.x) {
        const double cj = (cdf[j] + tile_off[j / SCAN_TILE]) / s_last;
        if (j == 0 && cj > u) {                       // nothing <= u: index 0
            ctl->found = 1; ctl->pick_idx = 0; ctl->pick_ok = (cj - u > tol) ? 1 : 0;
        }
        if (cj <= u) {
            const bool last = j == n - 1;
            const double cn = last ? inf64() : (cdf[j + 1] + tile_off[(j + 1) / SCAN_TILE]) / s_last;
            if (cn > u) {
                ctl->found = 1; ctl->pick_idx = j + 1;
                ctl->pick_ok = (!last && (u - cj > tol) && (cn - u > tol)) ? 1 : 0;
            }
        }
    }
}

__global__ void kpp_commit_kernel(KppCtl *ctl, int32_t *centers, uint8_t *is_center, int32_t next_pos, int32_t iter) {
    if (ctl->abort_iter >= 0) return;
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    if (!ctl->found || !ctl->pick_ok) { ctl->abort_iter = iter; ctl->abort_reason = 2; return; }
    centers[next_pos] = ctl->pick_idx;
    is_center[ctl->pick_idx] = 1;
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
    int32_t *flags;
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
    o->flags = ar.take<int32_t>(16);
    o->probs = ar.take<float>((size_t)n);
    o->part_max = ar.take<float>(FINISH_GRID);
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
           geo::align_up(64) + geo::align_up((size_t)n * 4) + geo::align_up(FINISH_GRID * 4) +
           2 * geo::align_up(plan_ints_bound(n) * 4) + 4096;
}

extern "C" int geo_kpp_chain(const int32_t *indptr, const int32_t *indices, const float *weights, int32_t n,
                             int32_t *centers, uint8_t *is_center, float *dmin, int32_t *argmin, const double *u_host,
                             int32_t it0, int32_t it1, int32_t n_centers_total, int32_t sweeps_per_solve, void *ws,
                             size_t ws_bytes, int32_t *status_out, void *stream_) {
    hipStream_t s = static_cast<hipStream_t>(stream_);
    GEO_REQUIRE(indptr && indices && centers && is_center && dmin && argmin && ws && status_out,
                "geo_kpp_chain: null pointer");
    GEO_REQUIRE(n > 0 && 0 <= it0 && it0 <= it1 && it1 <= n_centers_total, "geo_kpp_chain: bad iteration range");
    GEO_REQUIRE(sweeps_per_solve >= 2 && sweeps_per_solve <= 4096, "geo_kpp_chain: sweeps_per_solve out of range");
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
    h0.abort_iter = -1; h0.abort_reason = 0; h0.cur_src = -1; h0.found = 0; h0.pick_idx = -1; h0.pick_ok = 0;
    h0.total = 0.f; h0.s_last = 0.0;
    GEO_HIP_CHECK(hipMemcpyAsync(w.ctl, &h0, sizeof(KppCtl), hipMemcpyHostToDevice, s));

    const int g_lin = geo::grid_for(n, 256, 2048);
    const int g_sweep = geo::grid_for(n, 16, 4096);
    const int n_tiles = (n + SCAN_TILE - 1) / SCAN_TILE;
    const double tol = ((double)n + 16.0) * 4.440892098500626e-16;            // (n+16) * 2^-51
    for (int32_t t = it0; t < it1; ++t) {
        kpp_init_kernel<<<g_lin, 256, 0, s>>>(w.ctl, centers, t, w.d, n, w.flags);
        int last_cur = 0;
        for (int sw = 0; sw < sweeps_per_solve; ++sw) {
            const int cur = sw % 3, prev = (sw + 2) % 3, next = (sw + 1) % 3;
            if (weights)
                kpp_sweep_kernel<true><<<g_sweep, 256, 0, s>>>(w.ctl, indptr, indices, weights, n, w.d, w.flags, prev, cur, next, sw == 0);
            else
                kpp_sweep_kernel<false><<<g_sweep, 256, 0, s>>>(w.ctl, indptr, indices, weights, n, w.d, w.flags, prev, cur, next, sw == 0);
            last_cur = cur;
        }
        kpp_verdict_kernel<<<1, 1, 0, s>>>(w.ctl, w.flags, last_cur, t);
        kpp_finish_kernel<<<FINISH_GRID, 256, 0, s>>>(w.ctl, w.d, n, dmin, argmin, t, w.part_max);
        if (t + 1 < n_centers_total) {
            kpp_probs_kernel<<<g_lin, 256, 0, s>>>(w.ctl, dmin, is_center, n, w.part_max, FINISH_GRID, w.probs);
            kpp_total_kernel<<<1, 1024, 0, s>>>(w.ctl, w.probs, dp.leaf_start, dp.leaf_len, dp.n_leaves, dp.node_l, dp.node_r,
                                                dp.level_off, dp.n_levels, dp.chunk_root, dp.n_chunks, dp.val, t);
            kpp_scan_tiles_kernel<<<n_tiles, SCAN_T, 0, s>>>(w.ctl, w.probs, n, w.cdf, w.tile_sum);
            kpp_scan_offsets_kernel<<<1, 64, 0, s>>>(w.ctl, w.tile_sum, n_tiles, w.tile_off);
            kpp_pick_kernel<<<g_lin, 256, 0, s>>>(w.ctl, w.cdf, w.tile_off, n, u_host[t], tol);
            kpp_commit_kernel<<<1, 64, 0, s>>>(w.ctl, centers, is_center, t + 1, t);
        }
        GEO_LAUNCH_CHECK();
    }
    KppCtl h1;
    GEO_HIP_CHECK(hipMemcpyAsync(&h1, w.ctl, sizeof(KppCtl), hipMemcpyDeviceToHost, s));
    GEO_HIP_CHECK(hipStreamSynchronize(s));
    status_out[0] = h1.abort_iter;
    status_out[1] = h1.abort_reason;
    return GEO_OK;
}

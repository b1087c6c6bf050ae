// medoid.hip -- medoid update / re-assignment over a resident all-pairs geodesic matrix (gfx950).
//
// The reference stops after the k-means++ seeding and one assignment (src/geo/kmeans_optimized.py:141-183); its notes
// list the medoid update as the missing step ("iterative medoid update (Voronoi/PAM) over D").
// With 288 GB of HBM the full N x N float32 matrix of a 60 000-latent set (14.4 GB) stays resident: the K-source solve
// fills it 512 rows at a time, and one Voronoi iteration is
//   cost[i]   = sum over the members j of i's cluster of D[i][j]^power        (cluster_cost_kernel)
//   medoid[c] = the member of c with the smallest cost, lowest index on ties   (host side: two scatter-min passes)
//   assign[j] = argmin_c D[medoid[c]][j], first medoid on ties                 (rows_argmin_kernel)
// There is no reference implementation: parity is against the numpy restatement in oracle/kmedoids.py
// (voronoi_iteration), see DESIGN.md "Extensions".
#include "geo_common.h"

namespace {

// One wave per row i.  members = order[offsets[c] .. offsets[c+1]) are the nodes of i's cluster c (ascending node
// index inside a cluster); lane l takes members l, l + 64, ... in that order, fp64 accumulation of the float32 entries
// raised to `power` (1 or 2; the squares are exact in fp64), then the xor butterfly 32, 16, ..., 1: a fixed summation
// tree, the same for every run.
__global__ __launch_bounds__(256) void cluster_cost_kernel(const float *__restrict__ D, int64_t ld,
                                                          const int32_t *__restrict__ assign,
                                                          const int32_t *__restrict__ order,
                                                          const int32_t *__restrict__ offsets, int32_t n, int32_t power,
                                                          double *__restrict__ cost) {
    const int lane = threadIdx.x & 63;
    const int64_t i = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= n) return;
    const int32_t c = assign[i];
    const int32_t m0 = offsets[c], m1 = offsets[c + 1];
    const float *row = D + i * ld;
    double acc = 0.0;
    for (int32_t m = m0 + lane; m < m1; m += 64) {
        const double x = (double)row[order[m]];
        acc += power == 2 ? x * x : x;
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) acc += __shfl_xor(acc, off, 64);
    if (lane == 0) cost[i] = acc;
}

// Thread per column j: the first row (in the order given) with the smallest D[rows[m]][j]; coalesced across j.
__global__ __launch_bounds__(256) void rows_argmin_kernel(const float *__restrict__ D, int64_t ld,
                                                         const int32_t *__restrict__ rows, int32_t n_rows, int32_t n,
                                                         float *__restrict__ dmin, int32_t *__restrict__ argmin) {
    const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    float best = __int_as_float(0x7f800000);
    int32_t arg = 0;
    for (int32_t m = 0; m < n_rows; ++m) {
        const float v = D[(int64_t)rows[m] * ld + j];
        if (v < best) { best = v; arg = m; }                    // strict: the first index wins ties (np.argmin)
    }
    if (dmin) dmin[j] = best;
    if (argmin) argmin[j] = arg;
}

// One wave per new point v: the k attachment edges (nbr[v][u], len[v][u]) join v to the graph; its geodesic distance to
// medoid m is min_u len[v][u] + Dt[nbr[v][u]][m] (Dt = the medoids' distance rows transposed: [node][medoid], so the 64
// lanes read 64 consecutive medoids of one node).  Smallest distance and FIRST medoid attaining it; nbr < 0 = no edge.
__global__ __launch_bounds__(256) void attach_argmin_kernel(const float *__restrict__ Dt, int64_t ld, int32_t K,
                                                           const int32_t *__restrict__ nbr, const float *__restrict__ len,
                                                           int32_t k, int64_t n_new, float *__restrict__ dist_out,
                                                           int32_t *__restrict__ arg_out) {
    const int lane = threadIdx.x & 63;
    const int64_t v = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (v >= n_new) return;
    float best = __int_as_float(0x7f800000);
    int32_t arg = 0x7fffffff;
    for (int32_t m0 = 0; m0 < K; m0 += 64) {
        const int32_t m = m0 + lane;
        float dm = __int_as_float(0x7f800000);
        if (m < K) {
            for (int32_t u = 0; u < k; ++u) {
                const int32_t node = nbr[v * k + u];           // wave-uniform
                if (node < 0) continue;
                const float cand = len[v * k + u] + Dt[(int64_t)node * ld + m];
                dm = cand < dm ? cand : dm;
            }
        }
        if (dm < best) { best = dm; arg = m; }                 // ascending m per lane: strict < keeps the first
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {                  // (distance, medoid index) lexicographic minimum
        const float ob = __shfl_xor(best, off, 64);
        const int32_t oa = __shfl_xor(arg, off, 64);
        if (ob < best || (ob == best && oa < arg)) { best = ob; arg = oa; }
    }
    if (lane == 0) {
        dist_out[v] = best;
        arg_out[v] = arg == 0x7fffffff ? 0 : arg;              // no finite path: np.argmin of an all-inf column is 0
    }
}

// PAM's SWAP evaluation in the FastPAM1 form: for a candidate x (a row of D) the change of the total cost when medoid i is
// replaced by x is, with c = D[x][j]^power and c1(j) <= c2(j) the costs of j's nearest / second-nearest medoid,
//   dTD(i, x) = sum_j min(c - c1, 0)                                          (x takes j over from its medoid)
//             + sum_{j: nearest(j) = i} [ min(c, c2) - c1 - min(c - c1, 0) ]  (j loses medoid i: goes to x or to its second)
// The bracket equals c2 - c1 unless c < c2, so with base[i] = sum_{nearest(j) = i} (c2 - c1) (one small pass per swap
// evaluation, geo_pam_swap_deltas's `base` argument) it is base[i] + sum_{nearest(j) = i, c < c2} (max(c, c1) - c2): only the
// few nodes that are closer to x than to their second medoid contribute per-medoid terms.
// One workgroup per PAIR of candidate rows, streamed in their natural order (coalesced: the matrix is read exactly once per
// pass, n^2 * 4 bytes, HBM-bound; nearest / d1 / d2 are 12 bytes per node shared by both rows and by all workgroups: they stay
// in the L2).  The first sum is kept in registers (lane-strided order, wave butterfly, waves in order); the rare per-medoid
// terms go to the wave's own accumulators in LDS (ds_add_f64; program order inside a wave), combined in wave order; one thread
// per row picks the best medoid (first on ties).
// (Earlier forms: cluster-sorted member order with a fixed summation tree per cluster -- the gather into the row re-fetched
// every 128-byte line ~30 times from HBM, 54 ms per pass at 60 000 x 60 000; natural order with an LDS atomic per (row, node)
// -- bound by the LDS atomic rate, 9.5 ms.)
constexpr int PAM_ROWS = 2;

template <bool PER_WAVE>
__global__ __launch_bounds__(256) void pam_swap_kernel(const float *__restrict__ D, int64_t ld, const int32_t *__restrict__ nearest,
                                                      const float *__restrict__ d1v, const float *__restrict__ d2v,
                                                      const double *__restrict__ base, const uint8_t *__restrict__ is_medoid,
                                                      int32_t n, int32_t K, int32_t power,
                                                      double *__restrict__ best_delta, int32_t *__restrict__ best_medoid) {
    extern __shared__ double sh[];                             // B terms: PER_WAVE [4 waves][PAM_ROWS][K], else [PAM_ROWS][K]
    __shared__ double s_all[4][PAM_ROWS];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t x0 = (int64_t)blockIdx.x * PAM_ROWS;
    const int copies = PER_WAVE ? 4 : 1;
    for (int i = threadIdx.x; i < copies * PAM_ROWS * K; i += 256) sh[i] = 0.0;
    __syncthreads();
    double *Bw = sh + (PER_WAVE ? (size_t)wave * PAM_ROWS * K : 0);
    const float *row[PAM_ROWS];
    bool live[PAM_ROWS];
#pragma unroll
    for (int r = 0; r < PAM_ROWS; ++r) {
        live[r] = x0 + r < n && !is_medoid[x0 + r];            // a medoid is no candidate
        row[r] = D + (x0 + (live[r] ? r : 0)) * ld;
    }
    double all[PAM_ROWS] = {0.0, 0.0};
    // a lane takes 4 consecutive nodes per step (16-byte loads of the rows and of d1 / d2 / nearest; four steps in flight)
    const bool vec = (ld & 3) == 0 && (n & 3) == 0;
    const int32_t steps = (n + 255) / 256, per = (steps + 3) / 4;
    const int32_t s0 = wave * per, s1 = (s0 + per < steps) ? s0 + per : steps;
    auto relax = [&](int r, int32_t j, float dj, float e1f, float e2f, int32_t cl) {
        const double e1 = (double)e1f, e2 = (double)e2f, d = (double)dj;
        const double c1 = power == 2 ? e1 * e1 : e1, c2 = power == 2 ? e2 * e2 : e2, c = power == 2 ? d * d : d;
        all[r] += fmin(c - c1, 0.0);
        if (c < c2) atomicAdd(&Bw[(size_t)r * K + cl], fmax(c, c1) - c2);      // rare: x is nearer than j's second medoid
    };
    if (vec) {
        constexpr int U = 4;                                   // steps in flight per lane
        for (int32_t st = s0; st < s1; st += U) {
            float4 q[U][PAM_ROWS], e1[U], e2[U];
            int4 cl[U];
            bool ok[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int32_t j = ((st + u) * 64 + lane) * 4;
                ok[u] = st + u < s1 && j < n;
                if (ok[u]) {
#pragma unroll
                    for (int r = 0; r < PAM_ROWS; ++r) q[u][r] = *reinterpret_cast<const float4 *>(row[r] + j);
                    e1[u] = *reinterpret_cast<const float4 *>(d1v + j);
                    e2[u] = *reinterpret_cast<const float4 *>(d2v + j);
                    cl[u] = *reinterpret_cast<const int4 *>(nearest + j);
                }
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                if (!ok[u]) continue;
                const int32_t j = ((st + u) * 64 + lane) * 4;
#pragma unroll
                for (int r = 0; r < PAM_ROWS; ++r) {
                    if (!live[r]) continue;                    // block-uniform
                    relax(r, j, q[u][r].x, e1[u].x, e2[u].x, cl[u].x);
                    relax(r, j + 1, q[u][r].y, e1[u].y, e2[u].y, cl[u].y);
                    relax(r, j + 2, q[u][r].z, e1[u].z, e2[u].z, cl[u].z);
                    relax(r, j + 3, q[u][r].w, e1[u].w, e2[u].w, cl[u].w);
                }
            }
        }
    } else {
        for (int32_t st = s0; st < s1; ++st)
            for (int t = 0; t < 4; ++t) {
                const int32_t j = (st * 4 + t) * 64 + lane;
                if (j >= n) continue;
#pragma unroll
                for (int r = 0; r < PAM_ROWS; ++r)
                    if (live[r]) relax(r, j, row[r][j], d1v[j], d2v[j], nearest[j]);
            }
    }
#pragma unroll
    for (int r = 0; r < PAM_ROWS; ++r) {
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) all[r] += __shfl_xor(all[r], off, 64);
        if (lane == 0) s_all[wave][r] = all[r];
    }
    __syncthreads();
    if (PER_WAVE)
        for (int i = threadIdx.x; i < PAM_ROWS * K; i += 256)  // waves in order
            sh[i] = ((sh[i] + sh[(size_t)PAM_ROWS * K + i]) + sh[(size_t)2 * PAM_ROWS * K + i]) + sh[(size_t)3 * PAM_ROWS * K + i];
    __syncthreads();
    if (threadIdx.x < PAM_ROWS && x0 + threadIdx.x < n) {
        const int r = threadIdx.x;
        double best = __longlong_as_double(0x7ff0000000000000LL);
        int32_t arg = 0;
        if (live[r]) {
            const double shared_term = ((s_all[0][r] + s_all[1][r]) + s_all[2][r]) + s_all[3][r];
            for (int32_t i = 0; i < K; ++i) {
                const double dtd = shared_term + (base[i] + sh[(size_t)r * K + i]);
                if (dtd < best) { best = dtd; arg = i; }       // strict: the first medoid wins ties
            }
        }
        best_delta[x0 + r] = best;
        best_medoid[x0 + r] = arg;
    }
}

}  // namespace

extern "C" int geo_pam_swap_deltas(const float *D, int64_t ld, const int32_t *nearest, const float *d1, const float *d2,
                                   const double *base, const uint8_t *is_medoid, int32_t n, int32_t K, int32_t power,
                                   double *best_delta_out, int32_t *best_medoid_out, void *stream_) {
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    GEO_REQUIRE(D && nearest && d1 && d2 && base && is_medoid && best_delta_out && best_medoid_out, "geo_pam_swap_deltas: null pointer");
    GEO_REQUIRE(n > 0 && K > 1 && K <= 3584 && ld >= n && (power == 1 || power == 2), "geo_pam_swap_deltas: bad n=%d K=%d ld=%lld power=%d",
                n, K, (long long)ld, power);
    const unsigned grid = (unsigned)((n + PAM_ROWS - 1) / PAM_ROWS);
    if (K <= 896)            // 8 K doubles of LDS per workgroup (<= 56 KB): accumulators per wave, combined in wave order
        pam_swap_kernel<true><<<grid, 256, (size_t)4 * PAM_ROWS * K * sizeof(double), stream>>>(D, ld, nearest, d1, d2, base, is_medoid, n, K, power,
                                                                                              best_delta_out, best_medoid_out);
    else                     // one set of accumulators for the workgroup (sums then depend on the waves' interleaving: ~1e-16 relative)
        pam_swap_kernel<false><<<grid, 256, (size_t)PAM_ROWS * K * sizeof(double), stream>>>(D, ld, nearest, d1, d2, base, is_medoid, n, K, power,
                                                                                           best_delta_out, best_medoid_out);
    GEO_LAUNCH_CHECK();
    return GEO_OK;
}

extern "C" int geo_attach_argmin(const float *Dt, int64_t ld, int32_t K, const int32_t *nbr, const float *len, int32_t k,
                                 int64_t n_new, float *dist_out, int32_t *arg_out, void *stream_) {
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    GEO_REQUIRE(Dt && nbr && len && dist_out && arg_out, "geo_attach_argmin: null pointer");
    GEO_REQUIRE(K > 0 && k > 0 && ld >= K && n_new >= 0, "geo_attach_argmin: bad K=%d k=%d ld=%lld", K, k, (long long)ld);
    if (n_new == 0) return GEO_OK;
    attach_argmin_kernel<<<(unsigned)((n_new + 3) / 4), 256, 0, stream>>>(Dt, ld, K, nbr, len, k, n_new, dist_out, arg_out);
    GEO_LAUNCH_CHECK();
    return GEO_OK;
}

extern "C" int geo_cluster_costs(const float *D, int64_t ld, const int32_t *assign, const int32_t *order,
                                 const int32_t *offsets, int32_t n, int32_t power, double *cost_out, void *stream_) {
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    GEO_REQUIRE(D && assign && order && offsets && cost_out, "geo_cluster_costs: null pointer");
    GEO_REQUIRE(n > 0 && ld >= n && (power == 1 || power == 2), "geo_cluster_costs: bad n=%d ld=%lld power=%d", n,
                (long long)ld, power);
    cluster_cost_kernel<<<(unsigned)((n + 3) / 4), 256, 0, stream>>>(D, ld, assign, order, offsets, n, power, cost_out);
    GEO_LAUNCH_CHECK();
    return GEO_OK;
}

extern "C" int geo_rows_argmin(const float *D, int64_t ld, const int32_t *rows, int32_t n_rows, int32_t n,
                               float *dmin_out, int32_t *argmin_out, void *stream_) {
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    GEO_REQUIRE(D && rows && (dmin_out || argmin_out), "geo_rows_argmin: null pointer");
    GEO_REQUIRE(n > 0 && n_rows > 0 && ld >= n, "geo_rows_argmin: bad n=%d rows=%d ld=%lld", n, n_rows, (long long)ld);
    rows_argmin_kernel<<<(unsigned)((n + 255) / 256), 256, 0, stream>>>(D, ld, rows, n_rows, n, dmin_out, argmin_out);
    GEO_LAUNCH_CHECK();
    return GEO_OK;
}

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
// replaced by x is
//   dTD(i, x) = sum_j min(c(x,j) - c1(j), 0)  +  sum_{j: nearest(j) = i} [ min(c(x,j), c2(j)) - c1(j) - min(c(x,j) - c1(j), 0) ]
// with c = D^power, c1 / c2 = cost to the nearest / second-nearest medoid -- all K medoids from ONE pass over row x.
// One workgroup per candidate row, streamed in its natural order (coalesced: the matrix is read exactly once per pass,
// n^2 * 4 bytes, HBM-bound; nearest / c1 / c2 are 1.2 MB shared by all rows and stay in the L2).  Each wave takes a quarter of
// the row and adds its terms to ITS OWN per-medoid accumulators in LDS (ds_add_f64; program order inside a wave), the four
// waves' accumulators are combined in wave order, thread 0 picks the best medoid (first on ties).
// (Two earlier forms walked the row in cluster-sorted member order to sum each cluster with a fixed tree: the gather into the
// row re-fetched every 128-byte line ~30 times from HBM -- 54 ms per pass at 60 000 x 60 000 instead of 4.)
template <bool PER_WAVE>
__global__ __launch_bounds__(256) void pam_swap_kernel(const float *__restrict__ D, int64_t ld, const int32_t *__restrict__ nearest,
                                                      const double *__restrict__ c1v, const double *__restrict__ c2v,
                                                      const uint8_t *__restrict__ is_medoid, int32_t n, int32_t K, int32_t power,
                                                      double *__restrict__ best_delta, int32_t *__restrict__ best_medoid) {
    extern __shared__ double sh[];                             // PER_WAVE: [4][2][K]; else [2][K] shared by the waves
    const int64_t x = blockIdx.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (is_medoid[x]) {                                        // (block-uniform) a medoid is no candidate
        if (threadIdx.x == 0) { best_delta[x] = __longlong_as_double(0x7ff0000000000000LL); best_medoid[x] = 0; }
        return;
    }
    const int copies = PER_WAVE ? 4 : 1;
    for (int i = threadIdx.x; i < copies * 2 * K; i += 256) sh[i] = 0.0;
    __syncthreads();
    double *Aw = sh + (PER_WAVE ? (size_t)wave * 2 * K : 0), *Bw = Aw + K;
    const float *row = D + x * ld;
    const int32_t blocks = (n + 63) / 64, per = (blocks + 3) / 4;
    const int32_t b0 = wave * per, b1 = (b0 + per < blocks) ? b0 + per : blocks;
    for (int32_t blk = b0; blk < b1; ++blk) {
        const int32_t j = blk * 64 + lane;
        if (j < n) {
            const double d = (double)row[j];
            const double c = power == 2 ? d * d : d;
            const double c1 = c1v[j], c2 = c2v[j];
            const int32_t cl = nearest[j];
            const double a = fmin(c - c1, 0.0);
            const double b = fmin(c, c2) - c1 - a;
            if (a != 0.0) atomicAdd(&Aw[cl], a);
            if (b != 0.0) atomicAdd(&Bw[cl], b);
        }
    }
    __syncthreads();
    if (PER_WAVE)
        for (int i = threadIdx.x; i < K; i += 256) {          // waves in order
            sh[i] = ((sh[i] + sh[2 * K + i]) + sh[4 * K + i]) + sh[6 * K + i];
            sh[K + i] = ((sh[K + i] + sh[3 * K + i]) + sh[5 * K + i]) + sh[7 * K + i];
        }
    __syncthreads();
    if (threadIdx.x == 0) {
        double all = 0.0;
        for (int32_t i = 0; i < K; ++i) all += sh[i];          // medoid order
        double best = __longlong_as_double(0x7ff0000000000000LL);
        int32_t arg = 0;
        for (int32_t i = 0; i < K; ++i) {
            const double dtd = all + sh[K + i];
            if (dtd < best) { best = dtd; arg = i; }           // strict: the first medoid wins ties
        }
        best_delta[x] = best;
        best_medoid[x] = arg;
    }
}

}  // namespace

extern "C" int geo_pam_swap_deltas(const float *D, int64_t ld, const int32_t *nearest, const double *c1, const double *c2,
                                   const uint8_t *is_medoid, int32_t n, int32_t K, int32_t power, double *best_delta_out,
                                   int32_t *best_medoid_out, void *stream_) {
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    GEO_REQUIRE(D && nearest && c1 && c2 && is_medoid && best_delta_out && best_medoid_out, "geo_pam_swap_deltas: null pointer");
    GEO_REQUIRE(n > 0 && K > 0 && K <= 3584 && ld >= n && (power == 1 || power == 2), "geo_pam_swap_deltas: bad n=%d K=%d ld=%lld power=%d",
                n, K, (long long)ld, power);
    if (K <= 896)            // 8 K doubles of LDS per workgroup (<= 56 KB): accumulators per wave, combined in wave order
        pam_swap_kernel<true><<<(unsigned)n, 256, 8 * (size_t)K * sizeof(double), stream>>>(D, ld, nearest, c1, c2, is_medoid, n, K, power,
                                                                                          best_delta_out, best_medoid_out);
    else                     // one pair of accumulators for the workgroup (sums then depend on the waves' interleaving: ~1e-16 relative)
        pam_swap_kernel<false><<<(unsigned)n, 256, 2 * (size_t)K * sizeof(double), stream>>>(D, ld, nearest, c1, c2, is_medoid, n, K, power,
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

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
// One workgroup per candidate; a wave takes whole clusters (members = order[offsets[i] .. offsets[i+1]), ascending node
// index; c1 / c2 are given in that member order, so they stream), lane-strided fp64 sums + xor butterfly = a fixed summation
// tree; the cluster terms are then combined in cluster order by thread 0.  Output per candidate: its best medoid to replace
// (first index on ties) and the change.  The matrix is read exactly once per pass: N^2 x 4 bytes, HBM-bound.
__global__ __launch_bounds__(256) void pam_swap_kernel(const float *__restrict__ D, int64_t ld, const int32_t *__restrict__ order,
                                                      const int32_t *__restrict__ offsets, const double *__restrict__ c1m,
                                                      const double *__restrict__ c2m, const uint8_t *__restrict__ is_medoid,
                                                      int32_t n, int32_t K, int32_t power, double *__restrict__ best_delta,
                                                      int32_t *__restrict__ best_medoid) {
    extern __shared__ double sh[];                             // A[K] (shared term per cluster), B[K] (own-cluster term)
    double *A = sh, *B = sh + K;
    const int64_t x = blockIdx.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (is_medoid[x]) {                                        // (block-uniform) a medoid is no candidate
        if (threadIdx.x == 0) { best_delta[x] = __longlong_as_double(0x7ff0000000000000LL); best_medoid[x] = 0; }
        return;
    }
    const float *row = D + x * ld;
    for (int32_t i = wave; i < K; i += 4) {
        const int32_t m0 = offsets[i], m1 = offsets[i + 1];
        double a = 0.0, b = 0.0;
        for (int32_t m = m0 + lane; m < m1; m += 64) {
            const double d = (double)row[order[m]];
            const double c = power == 2 ? d * d : d;
            const double c1 = c1m[m], c2 = c2m[m];
            const double shared = fmin(c - c1, 0.0);
            a += shared;
            b += fmin(c, c2) - c1 - shared;
        }
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) { a += __shfl_xor(a, off, 64); b += __shfl_xor(b, off, 64); }
        if (lane == 0) { A[i] = a; B[i] = b; }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        double all = 0.0;
        for (int32_t i = 0; i < K; ++i) all += A[i];           // cluster order
        double best = __longlong_as_double(0x7ff0000000000000LL);
        int32_t arg = 0;
        for (int32_t i = 0; i < K; ++i) {
            const double dtd = all + B[i];
            if (dtd < best) { best = dtd; arg = i; }           // strict: the first medoid wins ties
        }
        best_delta[x] = best;
        best_medoid[x] = arg;
    }
}

// The same evaluation with the row streamed in MEMBER order (members sorted by cluster): every wave takes a quarter of the
// member list in blocks of 64 consecutive members -- order / cluster / c1 / c2 coalesced, one gather into the row per member,
// several blocks in flight -- and reduces each block by a segmented inclusive scan over the lanes (a cluster is a contiguous
// run of lanes; Hillis-Steele, fixed order); the last lane of every run adds the run's sums to the wave's own per-cluster
// accumulators in LDS (runs of one block hit distinct clusters; blocks follow each other in program order), and the four
// waves' accumulators are combined in wave order.  Deterministic like the kernel above, ~10x its speed at K = 512 (that
// one walks 128 clusters per wave one after the other: latency-bound at 0.25 TB/s of matrix bytes).
__global__ __launch_bounds__(256) void pam_swap_stream_kernel(const float *__restrict__ D, int64_t ld, const int32_t *__restrict__ order,
                                                             const int32_t *__restrict__ member_cluster,
                                                             const double *__restrict__ c1m, const double *__restrict__ c2m,
                                                             const uint8_t *__restrict__ is_medoid, int32_t n, int32_t K,
                                                             int32_t power, double *__restrict__ best_delta,
                                                             int32_t *__restrict__ best_medoid) {
    extern __shared__ double sh[];                             // [4 waves][2][K]
    const int64_t x = blockIdx.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (is_medoid[x]) {
        if (threadIdx.x == 0) { best_delta[x] = __longlong_as_double(0x7ff0000000000000LL); best_medoid[x] = 0; }
        return;
    }
    for (int i = threadIdx.x; i < 8 * K; i += 256) sh[i] = 0.0;
    __syncthreads();
    double *Aw = sh + (size_t)wave * 2 * K, *Bw = Aw + K;
    const float *row = D + x * ld;
    const int32_t blocks = (n + 63) / 64, per = (blocks + 3) / 4;
    const int32_t b0 = wave * per, b1 = (b0 + per < blocks) ? b0 + per : blocks;
    for (int32_t blk = b0; blk < b1; ++blk) {
        const int32_t m = blk * 64 + lane;
        const bool valid = m < n;
        int32_t cl = K;                                       // sentinel: its own run, never stored
        double a = 0.0, b = 0.0;
        if (valid) {
            cl = member_cluster[m];
            const double d = (double)row[order[m]];
            const double c = power == 2 ? d * d : d;
            const double c1 = c1m[m], c2 = c2m[m];
            a = fmin(c - c1, 0.0);
            b = fmin(c, c2) - c1 - a;
        }
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {              // segmented inclusive scan: a run = equal cluster ids
            const double a2 = __shfl_up(a, off, 64), b2 = __shfl_up(b, off, 64);
            const int32_t c2l = __shfl_up(cl, off, 64);
            if (lane >= off && c2l == cl) { a += a2; b += b2; }
        }
        const int32_t nxt = __shfl_down(cl, 1, 64);
        if (valid && (lane == 63 || nxt != cl)) { Aw[cl] += a; Bw[cl] += b; }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < K; i += 256) {              // waves in order
        sh[i] = ((sh[i] + sh[2 * K + i]) + sh[4 * K + i]) + sh[6 * K + i];
        sh[K + i] = ((sh[K + i] + sh[3 * K + i]) + sh[5 * K + i]) + sh[7 * K + i];
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        double all = 0.0;
        for (int32_t i = 0; i < K; ++i) all += sh[i];
        double best = __longlong_as_double(0x7ff0000000000000LL);
        int32_t arg = 0;
        for (int32_t i = 0; i < K; ++i) {
            const double dtd = all + sh[K + i];
            if (dtd < best) { best = dtd; arg = i; }
        }
        best_delta[x] = best;
        best_medoid[x] = arg;
    }
}

}  // namespace

extern "C" int geo_pam_swap_deltas(const float *D, int64_t ld, const int32_t *order, const int32_t *offsets, const int32_t *member_cluster,
                                   const double *c1_members, const double *c2_members, const uint8_t *is_medoid, int32_t n, int32_t K,
                                   int32_t power, double *best_delta_out, int32_t *best_medoid_out, void *stream_) {
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    GEO_REQUIRE(D && order && offsets && member_cluster && c1_members && c2_members && is_medoid && best_delta_out && best_medoid_out,
                "geo_pam_swap_deltas: null pointer");
    GEO_REQUIRE(n > 0 && K > 0 && K <= 4096 && ld >= n && (power == 1 || power == 2), "geo_pam_swap_deltas: bad n=%d K=%d ld=%lld power=%d",
                n, K, (long long)ld, power);
    if (K <= 896)            // 8 K doubles of LDS per workgroup (<= 56 KB): the streaming kernel
        pam_swap_stream_kernel<<<(unsigned)n, 256, 8 * (size_t)K * sizeof(double), stream>>>(D, ld, order, member_cluster, c1_members, c2_members,
                                                                                           is_medoid, n, K, power, best_delta_out, best_medoid_out);
    else
        pam_swap_kernel<<<(unsigned)n, 256, 2 * (size_t)K * sizeof(double), stream>>>(D, ld, order, offsets, c1_members, c2_members, is_medoid,
                                                                                    n, K, power, best_delta_out, best_medoid_out);
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

// knn.hip -- exact fp64-ranked k-nearest-neighbour search (gfx950).
//
// Replaces sklearn NearestNeighbors(...).kneighbors as called from the reference's
// src/geo/knn_graph_optimized.py:40-42.  The ranking key is the fp64 squared distance, formed
// either by the expansion |x|^2 - 2 x.y + |y|^2 clamped at 0 (sklearn's brute-force path,
// _argkmin.pyx.tp:492-502, used for d > 15) or directly as sum (x-y)^2 (kd-tree path, d <= 15);
// both are fma chains over the dimensions in ascending order, ties are ordered by index.
//
// Mapping: one wave owns Q query rows; every lane owns ONE candidate row per step (64 candidates
// per step, coordinates converted f32->f64 once and reused for the Q queries).  Query coordinates
// are wave-uniform and stream through the scalar cache (fp64 copy in the workspace), so the
// inner loop is v_fma_f64 with one SGPR-pair operand -- no LDS traffic.  The k best of a query
// live one per lane (lane i = i-th smallest), so a hit is inserted with one ballot, one popcount
// and a one-lane shift of the whole wave: no per-lane divergence.  Hits are rare after the first
// steps (k * ln(N/k) per query), the loop is bound by the fp64 FMA rate.
#include "geo_common.h"

namespace {

constexpr int KNN_WAVES = 4;

__device__ __forceinline__ double inf64() { return __longlong_as_double(0x7ff0000000000000LL); }

__device__ __forceinline__ double readlane_f64(double x, int lane) {
    const long long b = __double_as_longlong(x);
    const int lo = __builtin_amdgcn_readlane((int)(b & 0xffffffffLL), lane);
    const int hi = __builtin_amdgcn_readlane((int)(b >> 32), lane);
    return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}

// zp32 [n][dp] f32 zero padded, zq64 [n][dp] f64 zero padded, nrm [n] fp64 fma-chain squared norms
__global__ __launch_bounds__(256) void knn_prep_kernel(const float *__restrict__ z, int64_t n, int d, int dp,
                                                      float *__restrict__ zp32, double *__restrict__ zq64,
                                                      double *__restrict__ nrm) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        double s = 0.0;
        for (int c = 0; c < dp; ++c) {
            const float x = c < d ? z[i * d + c] : 0.0f;
            zp32[i * dp + c] = x;
            zq64[i * dp + c] = (double)x;
            s = fma((double)x, (double)x, s);
        }
        nrm[i] = s;
    }
}

template <int DCH, int Q, bool EXPANSION>
__global__ __launch_bounds__(256) void knn_kernel(const float *__restrict__ zp32, const double *__restrict__ zq64,
                                                 const double *__restrict__ nrm, int64_t n, int nch, int kq,
                                                 int64_t row0, int64_t row1, int32_t *__restrict__ idx_out,
                                                 double *__restrict__ d2_out) {
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int64_t q0 = row0 + ((int64_t)blockIdx.x * KNN_WAVES + wave) * Q;
    if (q0 >= row1) return;
    const int dp = nch * DCH;

    double lv[Q];      // lane i: i-th smallest squared distance seen so far for query q
    int32_t li[Q];     //         and its corpus index
    double tau[Q];     // wave-uniform: current kq-th smallest
#pragma unroll
    for (int q = 0; q < Q; ++q) { lv[q] = inf64(); li[q] = -1; tau[q] = inf64(); }

    for (int64_t c0 = 0; c0 < n; c0 += 64) {
        const int64_t j = c0 + lane;
        const bool valid = j < n;
        const int64_t jc = valid ? j : n - 1;
        double acc[Q];
#pragma unroll
        for (int q = 0; q < Q; ++q) acc[q] = 0.0;
        for (int ch = 0; ch < nch; ++ch) {
            double cj[DCH];
            const float4 *src = reinterpret_cast<const float4 *>(zp32 + jc * dp + ch * DCH);
#pragma unroll
            for (int v = 0; v < DCH / 4; ++v) {
                const float4 f = src[v];
                cj[4 * v] = (double)f.x; cj[4 * v + 1] = (double)f.y;
                cj[4 * v + 2] = (double)f.z; cj[4 * v + 3] = (double)f.w;
            }
#pragma unroll
            for (int q = 0; q < Q; ++q) {
                int64_t qr = q0 + q;
                if (qr >= row1) qr = row1 - 1;
                const double *__restrict__ qv = zq64 + qr * dp + ch * DCH;     // wave-uniform address
#pragma unroll
                for (int k = 0; k < DCH; ++k) {
                    if (EXPANSION) {
                        acc[q] = fma(qv[k], cj[k], acc[q]);
                    } else {
                        const double t = qv[k] - cj[k];
                        acc[q] = fma(t, t, acc[q]);
                    }
                }
            }
        }
        const double nj = EXPANSION ? nrm[jc] : 0.0;
#pragma unroll
        for (int q = 0; q < Q; ++q) {
            double d2 = acc[q];
            if (EXPANSION) {
                int64_t qr = q0 + q;
                if (qr >= row1) qr = row1 - 1;
                d2 = (nrm[qr] + (-2.0 * d2)) + nj;
                if (!(d2 > 0.0)) d2 = 0.0;
            }
            unsigned long long hits = __ballot(valid && d2 < tau[q]);
            while (hits) {
                const int src_lane = __builtin_ctzll(hits);
                hits &= hits - 1;
                const double val = readlane_f64(d2, src_lane);
                if (!(val < tau[q])) continue;             // tau shrank since the ballot
                const int32_t id = (int32_t)(c0 + src_lane);
                // stable insertion behind equal keys: candidates arrive in ascending index order
                const int pos = __builtin_popcountll(__ballot(lv[q] <= val));
                const double up_v = __shfl_up(lv[q], 1, 64);
                const int32_t up_i = __shfl_up(li[q], 1, 64);
                if (lane > pos) { lv[q] = up_v; li[q] = up_i; }
                if (lane == pos) { lv[q] = val; li[q] = id; }
                tau[q] = readlane_f64(lv[q], kq - 1);
            }
        }
    }
#pragma unroll
    for (int q = 0; q < Q; ++q) {
        const int64_t qr = q0 + q;
        if (qr < row1 && lane < kq) {
            idx_out[(qr - row0) * kq + lane] = li[q];
            d2_out[(qr - row0) * kq + lane] = lv[q];
        }
    }
}

struct KnnPlan { int dch, nch, dp; };

KnnPlan plan_for(int d) {
    KnnPlan p;
    if (d <= 8) { p.dch = 8; p.nch = 1; }
    else if (d <= 16) { p.dch = 16; p.nch = 1; }
    else { p.dch = 32; p.nch = (d + 31) / 32; }
    p.dp = p.dch * p.nch;
    return p;
}

template <int DCH, int Q>
int launch_knn(bool expansion, const float *zp32, const double *zq64, const double *nrm, int64_t n, int nch, int kq,
               int64_t row0, int64_t row1, int32_t *idx_out, double *d2_out, hipStream_t s) {
    const int64_t rows = row1 - row0;
    const int64_t waves = (rows + Q - 1) / Q;
    const unsigned grid = (unsigned)((waves + KNN_WAVES - 1) / KNN_WAVES);
    if (expansion)
        knn_kernel<DCH, Q, true><<<grid, 256, 0, s>>>(zp32, zq64, nrm, n, nch, kq, row0, row1, idx_out, d2_out);
    else
        knn_kernel<DCH, Q, false><<<grid, 256, 0, s>>>(zp32, zq64, nrm, n, nch, kq, row0, row1, idx_out, d2_out);
    GEO_LAUNCH_CHECK();
    return GEO_OK;
}

}  // namespace

extern "C" size_t geo_knn_workspace_bytes(int64_t n, int32_t d) {
    if (n <= 0 || d <= 0) return 1024;
    const KnnPlan p = plan_for(d);
    return geo::align_up((size_t)n * p.dp * sizeof(float)) + geo::align_up((size_t)n * p.dp * sizeof(double)) +
           geo::align_up((size_t)n * sizeof(double)) + 1024;
}

extern "C" int geo_knn_topk(const float *z, int64_t n, int32_t d, int32_t n_neighbors, int32_t form, int64_t row0,
                            int64_t row1, int32_t *idx_out, double *d2_out, void *ws, size_t ws_bytes,
                            void *stream_) {
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    GEO_REQUIRE(z && idx_out && d2_out && ws, "geo_knn_topk: null pointer");
    GEO_REQUIRE(n > 0 && n < (int64_t)1 << 31, "geo_knn_topk: n=%lld out of range", (long long)n);
    GEO_REQUIRE(d > 0 && d <= 128, "geo_knn_topk: d=%d not in [1,128]", d);
    GEO_REQUIRE(n_neighbors > 0 && n_neighbors <= 64 && n_neighbors <= n,
                "geo_knn_topk: n_neighbors=%d not in [1, min(64, n)]", n_neighbors);
    GEO_REQUIRE(0 <= row0 && row0 <= row1 && row1 <= n, "geo_knn_topk: bad row range");
    if (row0 == row1) return GEO_OK;
    const KnnPlan p = plan_for(d);
    geo::Arena ar(ws, ws_bytes);
    float *zp32 = ar.take<float>((size_t)n * p.dp);
    double *zq64 = ar.take<double>((size_t)n * p.dp);
    double *nrm = ar.take<double>((size_t)n);
    if (!zp32 || !zq64 || !nrm) {
        geo::set_error("geo_knn_topk: workspace %zu too small", ws_bytes);
        return GEO_E_WORKSPACE;
    }
    knn_prep_kernel<<<geo::grid_for(n, 256, 4096), 256, 0, stream>>>(z, n, d, p.dp, zp32, zq64, nrm);
    GEO_LAUNCH_CHECK();
    const bool ex = form != 0;
    const bool small = (row1 - row0) < 16384;     // fewer queries per wave keeps the chip busy on small inputs
    int rc;
    if (p.dch == 8)
        rc = small ? launch_knn<8, 4>(ex, zp32, zq64, nrm, n, p.nch, n_neighbors, row0, row1, idx_out, d2_out, stream)
                   : launch_knn<8, 16>(ex, zp32, zq64, nrm, n, p.nch, n_neighbors, row0, row1, idx_out, d2_out, stream);
    else if (p.dch == 16)
        rc = small ? launch_knn<16, 4>(ex, zp32, zq64, nrm, n, p.nch, n_neighbors, row0, row1, idx_out, d2_out, stream)
                   : launch_knn<16, 16>(ex, zp32, zq64, nrm, n, p.nch, n_neighbors, row0, row1, idx_out, d2_out, stream);
    else
        rc = small ? launch_knn<32, 4>(ex, zp32, zq64, nrm, n, p.nch, n_neighbors, row0, row1, idx_out, d2_out, stream)
                   : launch_knn<32, 16>(ex, zp32, zq64, nrm, n, p.nch, n_neighbors, row0, row1, idx_out, d2_out, stream);
    return rc;
}

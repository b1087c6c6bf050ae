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

#include <cstdlib>

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

template <int DCH, int Q, bool EXPANSION, int LR = 1>
__global__ __launch_bounds__(256) void knn_kernel(const float *__restrict__ zp32, const double *__restrict__ zq64,
                                                 const double *__restrict__ nrm, const double *__restrict__ nrm_q,
                                                 int64_t n, int nch, int kq, int64_t row0, int64_t row1,
                                                 int32_t *__restrict__ idx_out, double *__restrict__ d2_out) {
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int64_t q0 = row0 + ((int64_t)blockIdx.x * KNN_WAVES + wave) * Q;
    if (q0 >= row1) return;
    const int dp = nch * DCH;

    // list position p = 64 r + lane holds the p-th smallest squared distance seen so far for query q (LR registers
    // per lane: lists of up to 64 LR entries) and its corpus index
    double lv[Q][LR];
    int32_t li[Q][LR];
    double tau[Q];     // wave-uniform: current kq-th smallest
    const int tau_reg = (kq - 1) >> 6, tau_lane = (kq - 1) & 63;
#pragma unroll
    for (int q = 0; q < Q; ++q) {
        tau[q] = inf64();
#pragma unroll
        for (int r = 0; r < LR; ++r) { lv[q][r] = inf64(); li[q][r] = -1; }
    }

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
                d2 = (nrm_q[qr] + (-2.0 * d2)) + nj;
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
                int pos = 0;
#pragma unroll
                for (int r = 0; r < LR; ++r) pos += __builtin_popcountll(__ballot(lv[q][r] <= val));
#pragma unroll
                for (int r = LR - 1; r >= 0; --r) {            // top register first: it takes lane 63 of the one below
                    double up_v = __shfl_up(lv[q][r], 1, 64);
                    int32_t up_i = __shfl_up(li[q][r], 1, 64);
                    if (r > 0) {
                        const double carry_v = readlane_f64(lv[q][r - 1], 63);
                        const int32_t carry_i = __builtin_amdgcn_readlane(li[q][r - 1], 63);
                        if (lane == 0) { up_v = carry_v; up_i = carry_i; }
                    }
                    const int g = 64 * r + lane;
                    if (g > pos) { lv[q][r] = up_v; li[q][r] = up_i; }
                    if (g == pos) { lv[q][r] = val; li[q][r] = id; }
                }
#pragma unroll
                for (int r = 0; r < LR; ++r)
                    if (r == tau_reg) tau[q] = readlane_f64(lv[q][r], tau_lane);
            }
        }
    }
#pragma unroll
    for (int q = 0; q < Q; ++q) {
        const int64_t qr = q0 + q;
#pragma unroll
        for (int r = 0; r < LR; ++r) {
            const int g = 64 * r + lane;
            if (qr < row1 && g < kq) {
                idx_out[(qr - row0) * kq + g] = li[q][r];
                d2_out[(qr - row0) * kq + g] = lv[q][r];
            }
        }
    }
}

// ---- exact search behind a float32 matrix-core filter (large corpora) -----------------------------------------------
// 1. tau[i] = the kq-th smallest exact distance from query i to every FILTER_STRIDE-th corpus row: an upper bound of
//    the kq-th smallest over the whole corpus (knn_kernel on the subset).
// 2. knn_scan_kernel: approximate squared distances of ALL pairs on the matrix cores (v_mfma_f32_32x32x2_f32, expansion
//    form with the fp64 norms); pair (i, j) is kept when  approx <= tau[i] + eps * (|x_i|^2 + |x_j|^2),  eps far above
//    the rounding of a d-term float32 dot product -- every true neighbour is kept, plus ~FILTER_STRIDE * kq others.
// 3. knn_refine_kernel: the kept candidates are re-evaluated with the same fp64 fma chains as knn_kernel and ranked
//    by (distance, index): the result is the one knn_kernel gives.  A query whose list overflows FILTER_CAP (heavily
//    duplicated points) is reported and the caller falls back to knn_kernel.
constexpr int FILTER_STRIDE = 16, FILTER_CAP = 1024;
constexpr int64_t TWO_LEVEL_MIN = 200000;      // from here on the thresholds themselves come from a filtered pass (below)
typedef float f32x16v __attribute__((ext_vector_type(16)));

__global__ __launch_bounds__(256) void knn_subset_kernel(const float *__restrict__ zp32, const double *__restrict__ nrm,
                                                        int64_t n, int dp, int stride, int64_t m,
                                                        float *__restrict__ zs32, double *__restrict__ nrm_s) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < m * dp; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = i / dp;
        const int c = (int)(i % dp);
        zs32[i] = zp32[r * stride * dp + c];
        if (c == 0) nrm_s[r] = nrm[r * stride];
    }
}

// u[i] = tau[i] - |x_i|^2 (1 - eps): the scan keeps (i, j) when  |x_j|^2 (1 - eps) - 2 x_i.x_j <= u[i]
__global__ __launch_bounds__(256) void knn_tau_kernel(const double *__restrict__ d2_sub, const double *__restrict__ nrm,
                                                     int64_t row0, int64_t rows, int kq, double eps,
                                                     double *__restrict__ u, int32_t *__restrict__ cnt) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < rows; i += (int64_t)gridDim.x * blockDim.x) {
        const double nq = nrm[row0 + i];
        u[i] = d2_sub[i * kq + (kq - 1)] + eps * nq - nq;
        cnt[i] = 0;
    }
}

template <int DP, int SCAN_CT>
__global__ __launch_bounds__(256) void knn_scan_kernel(const float *__restrict__ zp32, const double *__restrict__ nrm,
                                                      const double *__restrict__ u, int64_t n, int64_t row0, int64_t rows,
                                                      double eps, int splits, int32_t *__restrict__ cnt,
                                                      int32_t *__restrict__ list) {
    // The whole test runs on the matrix cores: with A = [-2 x_i, 1, -u_i] and B = [x_j, |x_j|^2 (1 - eps), 1] the
    // product is  |x_j|^2 (1 - eps) - 2 x_i.x_j - u_i,  negative exactly for the pairs to keep (u rounded up, the norm
    // rounded down to float32; eps covers the float32 roundings of the accumulation).
    constexpr int KI = DP / 2 + 1;                             // MFMA steps (2-deep each), the last one carries the test
    __shared__ float Bs[2][SCAN_CT / 32][KI][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int64_t qbase = (int64_t)(blockIdx.x / splits) * 128 + wave * 32;   // this wave's 32 queries (local row numbers)
    const int split = blockIdx.x % splits;                                  // ... against one slice of the corpus
    // A operand: lane holds A[m = r][k = 2i + h]
    float a[KI];
    {
        const bool live = qbase + r < rows;
        const int64_t q = live ? qbase + r : rows - 1;
#pragma unroll
        for (int i = 0; i < KI - 1; ++i) a[i] = -2.0f * zp32[(row0 + q) * DP + 2 * i + h];
        float ur = -3.0e38f;                                   // padded queries keep nothing
        if (live) {
            const double ud = u[q];
            ur = (float)ud;
            if ((double)ur < ud) ur = nextafterf(ur, 3.4e38f);
        }
        a[KI - 1] = h == 0 ? 1.0f : -ur;
    }
    const int64_t all_tiles = (n + SCAN_CT - 1) / SCAN_CT, per_split = (all_tiles + splits - 1) / splits;
    const int64_t tile_lo = per_split * split;
    const int64_t n_tiles = tile_lo + per_split < all_tiles ? tile_lo + per_split : all_tiles;
    if (tile_lo >= n_tiles) return;
    // staging in two halves: the global loads of the next tile are issued before the MFMAs of the current one and
    // written to the other LDS buffer after them
    constexpr int PER = SCAN_CT * DP / 256;                    // floats per thread: (candidate, slice of its row)
    const int sc = (threadIdx.x * PER) / DP, sk0 = (threadIdx.x * PER) % DP;
    float vals[PER];
    float nval = 3.0e38f;
    auto fetch = [&](int64_t tile) {
        const int64_t c0 = tile * SCAN_CT;
        const int64_t j = c0 + sc < n ? c0 + sc : n - 1;
#pragma unroll
        for (int k = 0; k < PER; k += 4) {
            const float4 f = *reinterpret_cast<const float4 *>(zp32 + j * DP + sk0 + k);
            vals[k] = f.x; vals[k + 1] = f.y; vals[k + 2] = f.z; vals[k + 3] = f.w;
        }
        if (threadIdx.x < SCAN_CT) {
            const int64_t jj = c0 + threadIdx.x;
            nval = 3.0e38f;                                    // beyond the corpus: never kept
            if (jj < n) {
                const double nd = nrm[jj] * (1.0 - eps);
                nval = (float)nd;
                if ((double)nval > nd) nval = nextafterf(nval, -3.4e38f);
            }
        }
    };
    auto put = [&](int buf) {
#pragma unroll
        for (int k = 0; k < PER; ++k) {
            const int kk = sk0 + k;
            Bs[buf][sc >> 5][kk >> 1][(kk & 1) * 32 + (sc & 31)] = vals[k];
        }
        if (threadIdx.x < SCAN_CT) {
            Bs[buf][threadIdx.x >> 5][KI - 1][threadIdx.x & 31] = nval;
            Bs[buf][threadIdx.x >> 5][KI - 1][32 + (threadIdx.x & 31)] = 1.0f;
        }
    };
    constexpr int EQ_CAP = 128;
    __shared__ unsigned long long eb[4][EQ_CAP];
    int qn = 0;                                                // pairs queued by this wave (wave-uniform)
    auto flush = [&]() {                                       // wave-uniform call
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_wave_barrier();
        const int m = qn < EQ_CAP ? qn : EQ_CAP;
        for (int i = lane; i < m; i += 64) {
            const unsigned long long e = eb[wave][i];
            const int64_t row = (int64_t)(e >> 32);
            const int32_t slot = atomicAdd(&cnt[row], 1);
            if (slot < FILTER_CAP) list[row * FILTER_CAP + slot] = (int32_t)(e & 0xffffffffu);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_wave_barrier();
        qn = 0;
    };
    // Kept pairs go to a per-wave LDS queue first (the returning global atomic that reserves a list slot costs
    // microseconds, too much inside the MFMA stream; the queue is flushed at tile ends).  The signs of a lane's 32
    // accumulator entries (two column tiles) are packed into one word; lanes then take their set bits one per round and
    // get queue positions from the round's ballot: no LDS atomics, a handful of branches per 2 048 pairs.
    auto keep_pairs = [&](const f32x16v &acc0, const f32x16v &acc1, int64_t cand0) {
        unsigned any = 0;
#pragma unroll
        for (int q = 0; q < 16; ++q) any |= __float_as_uint(acc0[q]) | __float_as_uint(acc1[q]);
        if (!__ballot((any & 0x80000000u) != 0)) return;       // wave-uniform: nothing to keep in these 2 048 pairs
        unsigned bits = 0;                                     // bit q: acc0[q] < 0, bit 16 + q: acc1[q] < 0
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            bits |= (__float_as_uint(acc0[q]) >> 31) << q;
            bits |= (__float_as_uint(acc1[q]) >> 31) << (16 + q);
        }
        for (;;) {
            const unsigned long long hit = __ballot(bits != 0);
            if (!hit) break;
            if (bits != 0) {
                const int b = __builtin_ctz(bits);
                bits &= bits - 1;
                const int q = b & 15;
                const int pos = qn + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(hit >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)hit, 0u));
                const int64_t row = qbase + (q & 3) + 8 * (q >> 2) + 4 * h;
                const int32_t cand = (int32_t)(cand0 + (b >> 4) * 32 + r);
                if (pos < EQ_CAP) {
                    eb[wave][pos] = ((unsigned long long)row << 32) | (unsigned)cand;
                } else {                                       // queue full (masses of duplicates): straight to the list
                    const int32_t slot = atomicAdd(&cnt[row], 1);
                    if (slot < FILTER_CAP) list[row * FILTER_CAP + slot] = cand;
                }
            }
            qn += __builtin_popcountll(hit);
        }
    };
    fetch(tile_lo);
    put((int)(tile_lo & 1));
    __syncthreads();
    for (int64_t tile = tile_lo; tile < n_tiles; ++tile) {
        const int buf = (int)(tile & 1);
        if (tile + 1 < n_tiles) fetch(tile + 1);
#pragma unroll
        for (int t = 0; t < SCAN_CT / 32; t += 2) {           // two column tiles in flight: independent accumulators
            f32x16v acc0, acc1;
#pragma unroll
            for (int i = 0; i < 16; ++i) { acc0[i] = 0.f; acc1[i] = 0.f; }
#pragma unroll
            for (int i = 0; i < KI; ++i) {
                acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], Bs[buf][t][i][lane], acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], Bs[buf][t + 1][i][lane], acc1, 0, 0, 0);
            }
            keep_pairs(acc0, acc1, tile * SCAN_CT + t * 32);       // a negative entry = a pair to keep
        }
        if (qn >= EQ_CAP / 2) flush();
        if (tile + 1 < n_tiles) put(buf ^ 1);
        __syncthreads();
    }
    flush();
}

// ---- the same filter on the bf16 matrix cores ---------------------------------------------------------------------------
// x = hi + lo with hi = bf16(x), lo = bf16(x - hi): 16 significant bits.  x_i.x_j ~ hi_i.hi_j + hi_i.lo_j + lo_i.hi_j on
// v_mfma_f32_32x32x16_bf16 (three 32-cycle instructions per 32 x 32 pairs and 16 dimensions, against DP/2+1 64-cycle
// float32 ones); the dropped lo.lo term and the parts' own roundings are below 2^-16 |x_i||x_j|, which the caller's eps
// (relative to |x_i|^2 + |x_j|^2) covers.  The accumulator starts at -u_i; a lane's 16 entries belong to one candidate j, so
// the test is acc < -|x_j|^2 (1 - eps): the lane's minimum against one value (round 4; before, |x_j|^2 entered through a fourth
// MFMA and the sign was the test, as in knn_scan_kernel).  The kept pairs take the same queue.
typedef __bf16 bf16x8k __attribute__((ext_vector_type(8)));

__device__ __forceinline__ unsigned short bf16_rne(float x) {
    unsigned u = __float_as_uint(x);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (unsigned short)(u >> 16);
}

// zb [n][2][dpb] bf16 bit patterns: the two parts of every coordinate (dpb = dp rounded up to 16, zero padded)
__global__ __launch_bounds__(256) void knn_split_kernel(const float *__restrict__ zp32, const double *__restrict__ nrm,
                                                       int64_t n, int64_t n_pad, int dp, int dpb, double eps,
                                                       unsigned short *__restrict__ zb, float *__restrict__ negn) {
    // rows n .. n_pad - 1 (the scan reads whole tiles of 256 candidates): zeros, and a threshold nothing falls below
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_pad * dpb; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = i / dpb;
        const int c = (int)(i % dpb);
        const float x = (c < dp && r < n) ? zp32[r * dp + c] : 0.0f;
        const unsigned short hi = bf16_rne(x);
        const float rest = x - __uint_as_float((unsigned)hi << 16);      // exact
        zb[(r * 2 + 0) * dpb + c] = hi;
        zb[(r * 2 + 1) * dpb + c] = bf16_rne(rest);
        if (c == 0) {
            // -|x_j|^2 (1 - eps), the magnitude rounded DOWN to float32 (the scan keeps a pair when acc < this)
            float nval = 3.0e38f;
            if (r < n) {
                const double nd = nrm[r] * (1.0 - eps);
                nval = (float)nd;
                if ((double)nval > nd) nval = nextafterf(nval, -3.4e38f);
            }
            negn[r] = -nval;
        }
    }
}

template <int DPB, int SCAN_CT>
__global__ __launch_bounds__(256) void knn_scan_bf16_kernel(const unsigned short *__restrict__ zb,
                                                           const unsigned short *__restrict__ zbq,
                                                           const float *__restrict__ negn, const double *__restrict__ u,
                                                           int64_t n, int64_t row0, int64_t rows, int splits,
                                                           int32_t *__restrict__ cnt, int32_t *__restrict__ list) {
    constexpr int KST = DPB / 16;                              // MFMA k-steps
    // bytes per candidate in LDS: the hi row, the lo row, no padding -- the 16-byte pieces of a row are XOR-swizzled with the row
    // number instead, so that 16 consecutive candidates' reads of one piece spread over all banks (round 3 padded the rows to
    // 80 bytes at d <= 16: 47 KB per workgroup, three per CU; 38 KB fit four, and the scan is bound by the waves' serial latency)
    constexpr int ROWB = 2 * DPB * 2, PR = ROWB / 16;          // PR = 4 / 8 / 16 pieces per row
    auto swz = [](int row) { return (row / (16 / PR)) & (PR - 1); };
    __shared__ __attribute__((aligned(16))) unsigned char Ts[2][SCAN_CT * ROWB];
    // -|x_j|^2 (1 - eps), |.| rounded down to float32: a lane's accumulator entries all belong to ONE candidate (its column), so
    // "distance below the threshold" is acc < -n_j with acc = -u_i - 2 x_i.x_j -- the minimum of the lane's 16 entries against one
    // value (8 v_min3 + 1 compare per 1 024 pairs).  Round 3 added n_j inside the accumulator through one more MFMA (A = ones,
    // B = n_j in three bf16 parts): a quarter of the kernel's matrix-core time, and 16 ORs for the sign test on top.
    __shared__ float Nf[2][SCAN_CT];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int64_t qbase = (int64_t)(blockIdx.x / splits) * 128 + wave * 32;   // this wave's 32 queries (local row numbers)
    const int split = blockIdx.x % splits;                                  // ... against one slice of the corpus
    const int sw = swz(r);                                     // swizzle of the candidate rows this lane reads (row = 32 t + r)
    // A operand: lane holds A[m = r][k = 16 ks + 8 h .. + 7] = -2 x (both parts: the scaling is exact)
    bf16x8k ahi[KST], alo[KST];
    float uq[16];                                              // u of the 16 rows this lane's accumulator entries belong to
    {
        const bool live = qbase + r < rows;
        const int64_t q = live ? qbase + r : rows - 1;
#pragma unroll
        for (int ks = 0; ks < KST; ++ks)
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float xh = __uint_as_float((unsigned)zbq[((row0 + q) * 2 + 0) * DPB + ks * 16 + h * 8 + e] << 16);
                const float xl = __uint_as_float((unsigned)zbq[((row0 + q) * 2 + 1) * DPB + ks * 16 + h * 8 + e] << 16);
                ahi[ks][e] = (__bf16)(-2.0f * xh);
                alo[ks][e] = (__bf16)(-2.0f * xl);
            }
#pragma unroll
        for (int qq = 0; qq < 16; ++qq) {
            const int64_t row = qbase + (qq & 3) + 8 * (qq >> 2) + 4 * h;
            float ur = -3.0e38f;                               // padded queries keep nothing
            if (row < rows) {
                const double ud = u[row];
                ur = (float)ud;
                if ((double)ur < ud) ur = nextafterf(ur, 3.4e38f);
            }
            uq[qq] = -ur;                                      // the accumulator starts at -u (the MFMA's C operand)
        }
    }
    f32x16v negu;
#pragma unroll
    for (int qq = 0; qq < 16; ++qq) negu[qq] = uq[qq];
    const int64_t all_tiles = (n + SCAN_CT - 1) / SCAN_CT, per_split = (all_tiles + splits - 1) / splits;
    const int64_t tile_lo = per_split * split;
    const int64_t n_tiles = tile_lo + per_split < all_tiles ? tile_lo + per_split : all_tiles;
    if (tile_lo >= n_tiles) return;
    // staging: a tile's candidates are one contiguous run of SCAN_CT x 4 DPB bytes in zb (rows padded to whole tiles by
    // knn_split_kernel, so no index is clamped): piece k * 256 + tid of 16 bytes per thread and k, from a wave-uniform base
    constexpr int PIECES = SCAN_CT * PR;                       // 16-byte pieces per tile
    constexpr int PER = PIECES / 256;
    static_assert(PIECES % 256 == 0, "tile pieces divide among the threads");
    unsigned vals[PER][4];                                     // (scalars: an array of uint4 stays in scratch memory)
    float nval = 0.f;
    auto fetch = [&](int64_t tile) {
        const uint4 *tp = reinterpret_cast<const uint4 *>(zb + tile * SCAN_CT * 2 * DPB);
#pragma unroll
        for (int k = 0; k < PER; ++k) {
            const uint4 f = tp[k * 256 + threadIdx.x];
            vals[k][0] = f.x; vals[k][1] = f.y; vals[k][2] = f.z; vals[k][3] = f.w;
        }
        if (threadIdx.x < SCAN_CT) nval = negn[tile * SCAN_CT + threadIdx.x];
    };
    auto put = [&](int buf) {
#pragma unroll
        for (int k = 0; k < PER; ++k) {
            const int piece = k * 256 + threadIdx.x;
            const int sc = piece / PR, off = piece % PR;
            *reinterpret_cast<uint4 *>(&Ts[buf][sc * ROWB + (off ^ swz(sc)) * 16]) = make_uint4(vals[k][0], vals[k][1], vals[k][2], vals[k][3]);
        }
        if (threadIdx.x < SCAN_CT) Nf[buf][threadIdx.x] = nval;
    };
    constexpr int EQ_CAP = 128;
    __shared__ unsigned long long eb[4][EQ_CAP];
    int qn = 0;                                                // pairs queued by this wave (wave-uniform)
    auto flush = [&]() {                                       // wave-uniform call
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_wave_barrier();
        const int m = qn < EQ_CAP ? qn : EQ_CAP;
        for (int i = lane; i < m; i += 64) {
            const unsigned long long e = eb[wave][i];
            const int64_t row = (int64_t)(e >> 32);
            const int32_t slot = atomicAdd(&cnt[row], 1);
            if (slot < FILTER_CAP) list[row * FILTER_CAP + slot] = (int32_t)(e & 0xffffffffu);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_wave_barrier();
        qn = 0;
    };
    // Kept pairs go to a per-wave LDS queue first (the returning global atomic that reserves a list slot costs
    // microseconds, too much inside the MFMA stream; the queue is flushed at tile ends).  One 32 x 32 tile at a time: the lane's
    // minimum against -n_j decides wave-wide whether anything is kept (8 v_min3 + a compare; ~70 % of the tiles end here).
    // Otherwise the signs of acc - (-n_j) of a lane's 16 entries are packed into one word (the sign of a float32 difference is
    // the sign of the exact difference, denormals included); lanes then take their set bits one per round and get queue
    // positions from the round's ballot: no LDS atomics, a handful of branches per tile with something to keep.
    const int32_t rowb = (int32_t)qbase + 4 * h;               // local row of accumulator entry q: rowb + (q & 3) + 8 (q >> 2)
    auto keep_tile = [&](const f32x16v &acc, float negn, int32_t cand) {
        // (the two halves of the lane's entries separately: a tile that keeps something builds the sign word of one half only)
        float ma = __builtin_fminf(__builtin_fminf(acc[0], acc[1]), acc[2]);
        float mb = __builtin_fminf(__builtin_fminf(acc[8], acc[9]), acc[10]);
        ma = __builtin_fminf(__builtin_fminf(ma, acc[3]), acc[4]);
        mb = __builtin_fminf(__builtin_fminf(mb, acc[11]), acc[12]);
        ma = __builtin_fminf(__builtin_fminf(ma, acc[5]), acc[6]);
        mb = __builtin_fminf(__builtin_fminf(mb, acc[13]), acc[14]);
        ma = __builtin_fminf(ma, acc[7]);
        mb = __builtin_fminf(mb, acc[15]);
        if (!__ballot(__builtin_fminf(ma, mb) < negn)) return;  // wave-uniform: nothing to keep in these 1 024 pairs
        unsigned bits = 0;                                     // bit 15 - q: acc[q] < -n
        if (__ballot(ma < negn)) {
            unsigned w = 0;
#pragma unroll
            for (int q = 0; q < 8; ++q) w = __builtin_amdgcn_alignbit(w, __float_as_uint(acc[q] - negn), 31);   // (w << 1) | sign
            bits = w << 8;
        }
        if (__ballot(mb < negn)) {
            unsigned w = 0;
#pragma unroll
            for (int q = 8; q < 16; ++q) w = __builtin_amdgcn_alignbit(w, __float_as_uint(acc[q] - negn), 31);
            bits |= w;
        }
        for (;;) {
            const unsigned long long hit = __ballot(bits != 0);
            if (!hit) break;
            if (bits != 0) {
                const int b = __builtin_ctz(bits);
                bits &= bits - 1;
                const int q = 15 - b;
                const int pos = qn + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(hit >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)hit, 0u));
                const int32_t row = rowb + (q & 3) + 8 * (q >> 2);
                if (pos < EQ_CAP) {
                    eb[wave][pos] = ((unsigned long long)(unsigned)row << 32) | (unsigned)cand;
                } else {                                       // queue full (masses of duplicates): straight to the list
                    const int32_t slot = atomicAdd(&cnt[row], 1);
                    if (slot < FILTER_CAP) list[(int64_t)row * FILTER_CAP + slot] = cand;
                }
            }
            qn += __builtin_popcountll(hit);
        }
    };
    fetch(tile_lo);
    put((int)(tile_lo & 1));
    __syncthreads();
    for (int64_t tile = tile_lo; tile < n_tiles; ++tile) {
        const int buf = (int)(tile & 1);
        if (tile + 1 < n_tiles) fetch(tile + 1);
        // One 32-candidate column tile per step, software-pipelined inside the wave: the fragments of tile t + 1 are requested
        // from LDS, the three MFMAs of tile t issued, and only then the keep test of tile t - 1 runs -- vector work and branches
        // of one tile under the matrix-core time of the next, instead of MFMAs -> wait -> test -> branch in sequence.
        constexpr int NT = SCAN_CT / 32;
        bf16x8k yh[2][KST], yl[2][KST];
        f32x16v accs[2];
        float negns[2];
        auto frags = [&](int t2, int slot) {
            const unsigned char *b = &Ts[buf][(t2 * 32 + r) * ROWB];
#pragma unroll
            for (int ks = 0; ks < KST; ++ks) {
                const int ph = ((2 * ks + h) ^ sw) * 16, pl = ((PR / 2 + 2 * ks + h) ^ sw) * 16;   // this lane's hi / lo piece
                yh[slot][ks] = *reinterpret_cast<const bf16x8k *>(b + ph);
                yl[slot][ks] = *reinterpret_cast<const bf16x8k *>(b + pl);
            }
            negns[slot] = Nf[buf][t2 * 32 + r];
        };
        const int32_t cand_t0 = (int32_t)(tile * SCAN_CT) + r;           // this lane's candidate in column tile 0
        frags(0, 0);
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const int cur = t & 1;
            const float negn_prev = negns[cur ^ 1];                    // (tile t - 1's, before the slot is refilled)
            if (t + 1 < NT) frags(t + 1, cur ^ 1);
            f32x16v acc = negu;
#pragma unroll
            for (int ks = 0; ks < KST; ++ks) {
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ahi[ks], yh[cur][ks], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ahi[ks], yl[cur][ks], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(alo[ks], yh[cur][ks], acc, 0, 0, 0);
            }
            accs[cur] = acc;
            __builtin_amdgcn_sched_barrier(0);
            if (t > 0) keep_tile(accs[cur ^ 1], negn_prev, cand_t0 + (t - 1) * 32);   // an entry below -n_j = a pair to keep
        }
        keep_tile(accs[(NT - 1) & 1], negns[(NT - 1) & 1], cand_t0 + (NT - 1) * 32);
        if (qn >= EQ_CAP / 2) flush();
        if (tile + 1 < n_tiles) put(buf ^ 1);
        __syncthreads();
    }
    flush();
}

// one wave per query: exact fp64 distances of its kept candidates (same fma chains as knn_kernel) into LDS, then the
// kq smallest in (distance, index) order -- the lists are a few hundred entries, unordered
template <int DCH, bool EXPANSION>
__global__ __launch_bounds__(256) void knn_refine_kernel(const float *__restrict__ zp32, const double *__restrict__ zq64,
                                                        const double *__restrict__ nrm, const double *__restrict__ nrm_q,
                                                        int nch, int kq, int64_t row0,
                                                        int64_t rows, const int32_t *__restrict__ cnt,
                                                        const int32_t *__restrict__ list, int32_t *__restrict__ idx_out,
                                                        double *__restrict__ d2_out, int32_t *__restrict__ overflow) {
    __shared__ double sv[KNN_WAVES][FILTER_CAP];
    // (the index of entry e is list[row][e]: no copy of it in LDS -- 35 KB per workgroup instead of 51, four per CU)
    __shared__ double cv[KNN_WAVES][64];                       // the entries at or below the kq-th distance, compacted
    __shared__ int32_t ci[KNN_WAVES][64];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int64_t row = (int64_t)blockIdx.x * KNN_WAVES + wave;
    if (row >= rows) return;
    const int dp = nch * DCH;
    const int32_t m_all = cnt[row];
    if (m_all > FILTER_CAP) { if (lane == 0) atomicAdd(overflow, 1); return; }
    const int64_t qr = row0 + row;
    // U candidates per lane at a time, all of their row loads in flight before the first fma: the rows are random 64- to 256-byte
    // gathers from a corpus far larger than the L2, and one candidate per lane and trip (round 3) paid that latency six times
    // over per query (27 ms of the 172 ms stage at one million latents).  Per candidate the fma chain is unchanged.
    constexpr int U = DCH <= 16 ? 4 : 2;
    for (int32_t c0 = 0; c0 < m_all; c0 += 64 * U) {
        int32_t j[U];
        bool valid[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            valid[u] = c0 + u * 64 + lane < m_all;
            j[u] = valid[u] ? list[row * FILTER_CAP + c0 + u * 64 + lane] : 0;
        }
        double acc[U];
#pragma unroll
        for (int u = 0; u < U; ++u) acc[u] = 0.0;
        for (int ch = 0; ch < nch; ++ch) {
            float4 f[U][DCH / 4];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const float4 *src = reinterpret_cast<const float4 *>(zp32 + (int64_t)j[u] * dp + ch * DCH);
#pragma unroll
                for (int v = 0; v < DCH / 4; ++v) f[u][v] = src[v];
            }
            const double *__restrict__ qv = zq64 + qr * dp + ch * DCH;     // wave-uniform address
#pragma unroll
            for (int u = 0; u < U; ++u)
#pragma unroll
                for (int v = 0; v < DCH / 4; ++v) {
                    const double c4[4] = {(double)f[u][v].x, (double)f[u][v].y, (double)f[u][v].z, (double)f[u][v].w};
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        if (EXPANSION) {
                            acc[u] = fma(qv[4 * v + k], c4[k], acc[u]);
                        } else {
                            const double t = qv[4 * v + k] - c4[k];
                            acc[u] = fma(t, t, acc[u]);
                        }
                    }
                }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            double d2 = acc[u];
            if (EXPANSION) {
                d2 = (nrm_q[qr] + (-2.0 * d2)) + nrm[j[u]];
                if (!(d2 > 0.0)) d2 = 0.0;
            }
            if (valid[u]) sv[wave][c0 + u * 64 + lane] = d2;
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");     // the wave's own LDS writes, read back below
    __builtin_amdgcn_wave_barrier();
    // Selection.  First the value of the kq-th smallest distance by quickselect over the wave's entries: a pivot from the window
    // (lo, hi) (lo: fewer than kq entries <= lo; hi: at least kq entries <= hi), one counting pass, ~2 ln(m) rounds of ~35
    // instructions.  Then the entries <= that value -- kq of them unless distances tie at the threshold -- are compacted (cv / ci)
    // and ordered by (distance, index) with a rank count, one entry per lane.  (Round 3 extracted the kq minima one by one: kq
    // rounds of an LDS rescan + a three-value butterfly, ~1 900 instructions per query, the bulk of this kernel's time.)
    // More than 64 entries at or below the threshold (masses of duplicates): the extraction loop below, as before.
    if (m_all >= kq) {
        double lo = -1.0, hi = inf64();                        // distances are >= 0 and finite
        int rot = 0;
        for (int guard = 0; guard < 2048; ++guard) {
            // pivot: the first entry inside (lo, hi) of the first lane (from a rotating start) that has one
            double cand = inf64();
            for (int32_t e = lane; e < m_all; e += 64) {
                const double v = sv[wave][e];
                if (v > lo && v < hi && !(cand < inf64())) cand = v;
            }
            unsigned long long has = __ballot(cand < inf64());
            if (!has) break;                                   // nothing strictly inside: the threshold is hi
            has = (has >> rot) | (rot ? has << (64 - rot) : 0ull);
            const int src = (__ffsll((long long)has) - 1 + rot) & 63;
            rot = (rot + 23) & 63;
            const double pivot = __shfl(cand, src, 64);
            int c = 0;
            for (int32_t e = lane; e < m_all; e += 64) c += sv[wave][e] <= pivot ? 1 : 0;
#pragma unroll
            for (int off = 32; off >= 1; off >>= 1) c += __shfl_xor(c, off, 64);
            if (c >= kq) hi = pivot; else lo = pivot;
        }
        // hi == inf: fewer than kq finite entries (cannot happen); otherwise compact the entries <= hi
        int c_le = 0;
        for (int32_t e = lane; e < m_all; e += 64) c_le += sv[wave][e] <= hi ? 1 : 0;
        int incl = c_le;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const int o = __shfl_up(incl, off, 64);
            if (lane >= off) incl += o;
        }
        const int total = __shfl(incl, 63, 64);
        if (hi < inf64() && total <= 64) {
            int at = incl - c_le;
            for (int32_t e = lane; e < m_all; e += 64) {
                const double v = sv[wave][e];
                if (v <= hi) { cv[wave][at] = v; ci[wave][at] = list[row * FILTER_CAP + e]; ++at; }
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            __builtin_amdgcn_wave_barrier();
            // rank of this lane's entry among the `total` compacted ones in (distance, index) order
            const double myv = lane < total ? cv[wave][lane] : inf64();
            const int32_t myi = lane < total ? ci[wave][lane] : 0x7fffffff;
            int rank = 0;
            for (int t2 = 0; t2 < total; ++t2) {
                const double ov = cv[wave][t2];
                const int32_t oi = ci[wave][t2];
                rank += (ov < myv || (ov == myv && oi < myi)) ? 1 : 0;
            }
            if (lane < total && rank < kq) {
                idx_out[row * kq + rank] = myi;
                d2_out[row * kq + rank] = myv;
            }
            return;
        }
    }
    // kq rounds of "extract the minimum in (distance, index) order": every lane scans its own strided entries,
    // a butterfly finds the wave's minimum, its owner retires it
    for (int round = 0; round < kq; ++round) {
        double bv = inf64();
        int32_t bi = 0x7fffffff, bpos = -1;
        for (int32_t e = lane; e < m_all; e += 64) {
            const double v = sv[wave][e];
            const int32_t id = list[row * FILTER_CAP + e];
            if (v < bv || (v == bv && id < bi)) { bv = v; bi = id; bpos = e; }
        }
        // the wave's smallest distance first (one 8-byte value through the butterfly), then its owner: almost always one lane;
        // equal distances in several lanes (duplicate latents) -> the lowest index among them
        double mv = bv;
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) mv = fmin(mv, __shfl_xor(mv, off, 64));
        if (!(mv < inf64())) break;                            // fewer candidates than kq (cannot happen: subset >= kq rows)
        const unsigned long long tied = __ballot(bv == mv);
        int owner = __ffsll((long long)tied) - 1;
        if (tied & (tied - 1)) {                               // wave-uniform
            int32_t mi = bv == mv ? bi : 0x7fffffff;
#pragma unroll
            for (int off = 32; off >= 1; off >>= 1) mi = min(mi, __shfl_xor(mi, off, 64));
            owner = __ffsll((long long)__ballot(bv == mv && bi == mi)) - 1;
        }
        bi = __shfl(bi, owner, 64);
        bpos = __shfl(bpos, owner, 64);
        if (lane == owner) sv[wave][bpos] = inf64();           // retired (real distances are finite)
        if (lane == 0) {
            idx_out[row * kq + round] = bi;
            d2_out[row * kq + round] = mv;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_wave_barrier();
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

template <int DCH, int Q, int LR = 1>
int launch_knn(bool expansion, const float *zp32, const double *zq64, const double *nrm, const double *nrm_q, int64_t n,
               int nch, int kq, int64_t row0, int64_t row1, int32_t *idx_out, double *d2_out, hipStream_t s) {
    const int64_t rows = row1 - row0;
    const int64_t waves = (rows + Q - 1) / Q;
    const unsigned grid = (unsigned)((waves + KNN_WAVES - 1) / KNN_WAVES);
    if (expansion)
        knn_kernel<DCH, Q, true, LR><<<grid, 256, 0, s>>>(zp32, zq64, nrm, nrm_q, n, nch, kq, row0, row1, idx_out, d2_out);
    else
        knn_kernel<DCH, Q, false, LR><<<grid, 256, 0, s>>>(zp32, zq64, nrm, nrm_q, n, nch, kq, row0, row1, idx_out, d2_out);
    GEO_LAUNCH_CHECK();
    return GEO_OK;
}

// lists longer than one wave (64 < kq <= 256): 2 or 4 list registers per lane, 4 queries per wave
template <int DCH>
int launch_knn_wide(bool expansion, const float *zp32, const double *zq64, const double *nrm, int64_t n, int nch, int kq,
                    int64_t row0, int64_t row1, int32_t *idx_out, double *d2_out, hipStream_t s) {
    if (kq <= 128) return launch_knn<DCH, 4, 2>(expansion, zp32, zq64, nrm, nrm, n, nch, kq, row0, row1, idx_out, d2_out, s);
    return launch_knn<DCH, 4, 4>(expansion, zp32, zq64, nrm, nrm, n, nch, kq, row0, row1, idx_out, d2_out, s);
}

bool filter_applies(int64_t n, int dp) {
    if (geo::options().knn_filter == 0) return false;
    // worth it from ~40 000 rows at 16 dimensions; the exact scan costs in proportion to the dimension
    const int64_t n_min = 640000 / dp > 16384 ? 640000 / dp : 16384;
    return n >= n_min && (dp == 8 || dp == 16 || dp == 32 || dp == 64);
}

}  // namespace

extern "C" size_t geo_knn_workspace_bytes(int64_t n, int32_t d) {
    if (n <= 0 || d <= 0) return 1024;
    const KnnPlan p = plan_for(d);
    size_t b = geo::align_up((size_t)n * p.dp * sizeof(float)) + geo::align_up((size_t)n * p.dp * sizeof(double)) +
               geo::align_up((size_t)n * sizeof(double)) + 1024;
    if (filter_applies(n, p.dp)) {
        const size_t m = ((size_t)n + FILTER_STRIDE - 1) / FILTER_STRIDE;
        const size_t dpb = p.dp < 16 ? 16 : p.dp;
        const size_t n_pad = ((size_t)n + 255) / 256 * 256, m_pad = (m + 255) / 256 * 256;   // the scan reads whole tiles
        b += geo::align_up(m * p.dp * sizeof(float)) + geo::align_up(m * sizeof(double)) + geo::align_up((size_t)n * 8) +
             geo::align_up((size_t)n * 4) + geo::align_up((size_t)n * FILTER_CAP * 4) + 256 +
             geo::align_up(n_pad * 2 * dpb * sizeof(unsigned short)) + geo::align_up(n_pad * sizeof(float));
        const size_t m0 = (m + FILTER_STRIDE - 1) / FILTER_STRIDE;
        b += geo::align_up(m0 * p.dp * sizeof(float)) + geo::align_up(m0 * sizeof(double)) +
             geo::align_up(m_pad * 2 * dpb * sizeof(unsigned short)) + geo::align_up(m_pad * sizeof(float));   // two-level thresholds
    }
    return b;
}

extern "C" int geo_knn_topk(const float *z, int64_t n, int32_t d, int32_t n_neighbors, int32_t form, int64_t row0,
                            int64_t row1, int32_t *idx_out, double *d2_out, void *ws, size_t ws_bytes,
                            void *stream_) {
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    GEO_REQUIRE(z && idx_out && d2_out && ws, "geo_knn_topk: null pointer");
    GEO_REQUIRE(n > 0 && n < (int64_t)1 << 31, "geo_knn_topk: n=%lld out of range", (long long)n);
    GEO_REQUIRE(d > 0 && d <= 128, "geo_knn_topk: d=%d not in [1,128]", d);
    GEO_REQUIRE(n_neighbors > 0 && n_neighbors <= GEO_KNN_MAX_NEIGHBORS && n_neighbors <= n,
                "geo_knn_topk: n_neighbors=%d not in [1, min(%d, n)]", n_neighbors, GEO_KNN_MAX_NEIGHBORS);
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
    const int64_t rows = row1 - row0;
    if (n_neighbors > 64) {                       // no float32 filter here: its candidate lists are sized for kq <= 64
        if (p.dch == 8) return launch_knn_wide<8>(ex, zp32, zq64, nrm, n, p.nch, n_neighbors, row0, row1, idx_out, d2_out, stream);
        if (p.dch == 16) return launch_knn_wide<16>(ex, zp32, zq64, nrm, n, p.nch, n_neighbors, row0, row1, idx_out, d2_out, stream);
        return launch_knn_wide<32>(ex, zp32, zq64, nrm, n, p.nch, n_neighbors, row0, row1, idx_out, d2_out, stream);
    }
    if (filter_applies(n, p.dp) && (n + FILTER_STRIDE - 1) / FILTER_STRIDE >= n_neighbors) {
        const int64_t m = (n + FILTER_STRIDE - 1) / FILTER_STRIDE;
        float *zs32 = ar.take<float>((size_t)m * p.dp);
        double *nrm_s = ar.take<double>((size_t)m);
        double *u = ar.take<double>((size_t)n);
        int32_t *cnt = ar.take<int32_t>((size_t)n);
        int32_t *list = ar.take<int32_t>((size_t)n * FILTER_CAP);
        int32_t *overflow = ar.take<int32_t>(16);
        const int dpb = p.dp < 16 ? 16 : p.dp;
        const bool bf16_scan = geo::options().knn_filter != 2;         // 2: the float32 matrix-core scan (comparison runs)
        const int64_t n_pad = (n + 255) / 256 * 256, m_pad = (m + 255) / 256 * 256;      // the scan reads whole tiles
        unsigned short *zb = ar.take<unsigned short>((size_t)n_pad * 2 * dpb);
        float *negn = ar.take<float>((size_t)n_pad);
        const size_t m0w = ((size_t)m + FILTER_STRIDE - 1) / FILTER_STRIDE;
        float *zs0 = ar.take<float>(m0w * p.dp);
        double *nrm_s0 = ar.take<double>(m0w);
        unsigned short *zb_s = ar.take<unsigned short>((size_t)m_pad * 2 * dpb);
        float *negn_s = ar.take<float>((size_t)m_pad);
        if (!zs0 || !nrm_s0 || !zb_s || !negn_s) zs0 = nullptr;   // (a caller with a smaller workspace: one level)
        if (!zs32 || !nrm_s || !u || !cnt || !list || !overflow || !zb || !negn) {
            geo::set_error("geo_knn_topk: workspace %zu too small", ws_bytes);
            return GEO_E_WORKSPACE;
        }
        // a d-term float32 dot product errs by < (d + 2) 2^-24 |x||y|, |x||y| <= (|x|^2 + |y|^2) / 2, and the float32
        // test adds a few more roundings of the same size: 16x margin
        // (the bf16 scan keeps 16 bits of every coordinate: its products err by < 2^-16 (|x|^2 + |y|^2) on top, 4x margin)
        const double eps = 16.0 * (p.dp + 2) * 5.9604644775390625e-08 + (bf16_scan ? 4.0 * 1.52587890625e-05 : 0.0);
        GEO_HIP_CHECK(hipMemsetAsync(overflow, 0, 4, stream));
        knn_subset_kernel<<<geo::grid_for(m * p.dp, 256, 4096), 256, 0, stream>>>(zp32, nrm, n, p.dp, FILTER_STRIDE, m, zs32, nrm_s);
        GEO_LAUNCH_CHECK();
        if (bf16_scan) {
            knn_split_kernel<<<geo::grid_for(n_pad * dpb, 256, 4096), 256, 0, stream>>>(zp32, nrm, n, n_pad, p.dp, dpb, eps, zb, negn);
            GEO_LAUNCH_CHECK();
        }
        const unsigned rgrid = (unsigned)((rows + KNN_WAVES - 1) / KNN_WAVES);
        auto splits_for = [&](int64_t corpus) {                // corpus slices per 128-query block: enough workgroups, tiles to spare
            const int64_t qb = (rows + 127) / 128;
            int64_t sp = 16384 / qb;
            if (sp > 64) sp = 64;
            if (sp > (corpus + 255) / 256) sp = (corpus + 255) / 256;
            return (int)(sp > 1 ? sp : 1);
        };
#define GEO_EXACT_SUBSET(ZS, NRMS, MS)                                                                              \
    (p.dch == 8 ? launch_knn<8, 16>(ex, ZS, zq64, NRMS, nrm, MS, p.nch, n_neighbors, row0, row1, idx_out, d2_out, stream)    \
     : p.dch == 16 ? launch_knn<16, 16>(ex, ZS, zq64, NRMS, nrm, MS, p.nch, n_neighbors, row0, row1, idx_out, d2_out, stream) \
                   : launch_knn<32, 16>(ex, ZS, zq64, NRMS, nrm, MS, p.nch, n_neighbors, row0, row1, idx_out, d2_out, stream))
#define GEO_REFINE(DCHV, EXV, ZC, NRMC) \
    knn_refine_kernel<DCHV, EXV><<<rgrid, 256, 0, stream>>>(ZC, zq64, NRMC, nrm, p.nch, n_neighbors, row0, rows, cnt, list, \
                                                          idx_out, d2_out, overflow)
#define GEO_REFINE_ANY(ZC, NRMC)                                                                                   \
    do {                                                                                                           \
        if (p.dch == 8) { if (ex) GEO_REFINE(8, true, ZC, NRMC); else GEO_REFINE(8, false, ZC, NRMC); }              \
        else if (p.dch == 16) { if (ex) GEO_REFINE(16, true, ZC, NRMC); else GEO_REFINE(16, false, ZC, NRMC); }      \
        else { if (ex) GEO_REFINE(32, true, ZC, NRMC); else GEO_REFINE(32, false, ZC, NRMC); }                       \
    } while (0)
        int rc = 0;
        bool tau_done = false;
        // Large inputs: the exact pass over every 16th row is itself N^2 / 16 fp64 work (100 ms of 255 at a million latents).
        // Two levels instead: thresholds from every 256th row (exact), with them the SAME filter + refinement over the
        // every-16th-row subset gives the exact kq-th distance to that subset -- the threshold the one-level scheme uses.
        const int64_t m0 = (m + FILTER_STRIDE - 1) / FILTER_STRIDE;
        if (bf16_scan && zs0 && n >= TWO_LEVEL_MIN && m0 >= n_neighbors) {
            knn_subset_kernel<<<geo::grid_for(m0 * p.dp, 256, 4096), 256, 0, stream>>>(zs32, nrm_s, m, p.dp, FILTER_STRIDE, m0, zs0, nrm_s0);
            GEO_LAUNCH_CHECK();
            rc = GEO_EXACT_SUBSET(zs0, nrm_s0, m0);
            if (rc) return rc;
            knn_tau_kernel<<<geo::grid_for(rows, 256, 4096), 256, 0, stream>>>(d2_out, nrm, row0, rows, n_neighbors, eps, u, cnt);
            GEO_LAUNCH_CHECK();
            knn_split_kernel<<<geo::grid_for(m_pad * dpb, 256, 4096), 256, 0, stream>>>(zs32, nrm_s, m, m_pad, p.dp, dpb, eps, zb_s, negn_s);
            GEO_LAUNCH_CHECK();
            const int sp1 = splits_for(m);
            const unsigned g1 = (unsigned)((rows + 127) / 128) * (unsigned)sp1;
            if (dpb == 16) knn_scan_bf16_kernel<16, 256><<<g1, 256, 0, stream>>>(zb_s, zb, negn_s, u, m, row0, rows, sp1, cnt, list);
            else if (dpb == 32) knn_scan_bf16_kernel<32, 256><<<g1, 256, 0, stream>>>(zb_s, zb, negn_s, u, m, row0, rows, sp1, cnt, list);
            else knn_scan_bf16_kernel<64, 128><<<g1, 256, 0, stream>>>(zb_s, zb, negn_s, u, m, row0, rows, sp1, cnt, list);
            GEO_LAUNCH_CHECK();
            GEO_REFINE_ANY(zs32, nrm_s);                       // kq smallest exact distances to the subset -> d2_out
            GEO_LAUNCH_CHECK();
            int32_t h_over0 = 0;
            GEO_HIP_CHECK(hipMemcpyAsync(&h_over0, overflow, 4, hipMemcpyDeviceToHost, stream));
            GEO_HIP_CHECK(hipStreamSynchronize(stream));
            tau_done = h_over0 == 0;
            if (!tau_done) GEO_HIP_CHECK(hipMemsetAsync(overflow, 0, 4, stream));
        }
        if (!tau_done) {
            rc = GEO_EXACT_SUBSET(zs32, nrm_s, m);
            if (rc) return rc;
        }
        knn_tau_kernel<<<geo::grid_for(rows, 256, 4096), 256, 0, stream>>>(d2_out, nrm, row0, rows, n_neighbors, eps, u, cnt);
        GEO_LAUNCH_CHECK();
        const int splits = splits_for(n);
        const unsigned sgrid = (unsigned)((rows + 127) / 128) * (unsigned)splits;
        if (bf16_scan) {
            if (dpb == 16) knn_scan_bf16_kernel<16, 256><<<sgrid, 256, 0, stream>>>(zb, zb, negn, u, n, row0, rows, splits, cnt, list);
            else if (dpb == 32) knn_scan_bf16_kernel<32, 256><<<sgrid, 256, 0, stream>>>(zb, zb, negn, u, n, row0, rows, splits, cnt, list);
            else knn_scan_bf16_kernel<64, 128><<<sgrid, 256, 0, stream>>>(zb, zb, negn, u, n, row0, rows, splits, cnt, list);
        } else
        if (p.dp == 8) knn_scan_kernel<8, 256><<<sgrid, 256, 0, stream>>>(zp32, nrm, u, n, row0, rows, eps, splits, cnt, list);
        else if (p.dp == 16) knn_scan_kernel<16, 256><<<sgrid, 256, 0, stream>>>(zp32, nrm, u, n, row0, rows, eps, splits, cnt, list);
        else if (p.dp == 32) knn_scan_kernel<32, 256><<<sgrid, 256, 0, stream>>>(zp32, nrm, u, n, row0, rows, eps, splits, cnt, list);
        else knn_scan_kernel<64, 128><<<sgrid, 256, 0, stream>>>(zp32, nrm, u, n, row0, rows, eps, splits, cnt, list);
        GEO_LAUNCH_CHECK();
        GEO_REFINE_ANY(zp32, nrm);
#undef GEO_REFINE_ANY
#undef GEO_REFINE
#undef GEO_EXACT_SUBSET
        GEO_LAUNCH_CHECK();
        int32_t h_over = 0;
        GEO_HIP_CHECK(hipMemcpyAsync(&h_over, overflow, 4, hipMemcpyDeviceToHost, stream));
        GEO_HIP_CHECK(hipStreamSynchronize(stream));
        if (h_over == 0) return GEO_OK;
        // some query kept more than FILTER_CAP candidates (masses of near-duplicates): exact scan for everything
    }
    const bool small = rows < 16384;              // fewer queries per wave keeps the chip busy on small inputs
    int rc;
    if (p.dch == 8)
        rc = small ? launch_knn<8, 4>(ex, zp32, zq64, nrm, nrm, n, p.nch, n_neighbors, row0, row1, idx_out, d2_out, stream)
                   : launch_knn<8, 16>(ex, zp32, zq64, nrm, nrm, n, p.nch, n_neighbors, row0, row1, idx_out, d2_out, stream);
    else if (p.dch == 16)
        rc = small ? launch_knn<16, 4>(ex, zp32, zq64, nrm, nrm, n, p.nch, n_neighbors, row0, row1, idx_out, d2_out, stream)
                   : launch_knn<16, 16>(ex, zp32, zq64, nrm, nrm, n, p.nch, n_neighbors, row0, row1, idx_out, d2_out, stream);
    else
        rc = small ? launch_knn<32, 4>(ex, zp32, zq64, nrm, nrm, n, p.nch, n_neighbors, row0, row1, idx_out, d2_out, stream)
                   : launch_knn<32, 16>(ex, zp32, zq64, nrm, nrm, n, p.nch, n_neighbors, row0, row1, idx_out, d2_out, stream);
    return rc;
}

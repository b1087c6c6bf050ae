// jvp.hip -- decoder pull-back edge lengths by forward-mode tangent propagation (gfx950).
//
// Replaces torch.autograd.functional.jvp through SpatialDecoder in the reference
// (src/geo/riemannian_metric.py:12-35,37-66 over src/models/spatial_vae.py:47-81):
//     len[e] = 0.5 * ( |J(z_i) dz| + |J(z_j) dz| ),  dz = z_j - z_i,  J = d sigmoid(decoder(z)) / dz
// with the decoder applied to a 1x1 latent "image":
//     conv1x1(d->c0) -> ConvT(c0->c1,k4,s2,p1): 1x1 -> 2x2 -> norm -> ReLU
//                    -> ConvT(c1->c2,k4,s2,p1): 2x2 -> 4x4 -> norm -> ReLU
//                    -> ConvT(c2->co,k4,s2,p=3|1): 4x4 -> 4x4 | 8x8 -> sigmoid.
// The tangent is pushed through next to the primal (one pass instead of autograd's forward +
// double backward).  Edges are processed in chunks of `batch_size` consecutive edges; each
// (chunk, endpoint side) is one BatchNorm batch, exactly the batches riemannian_metric.py:50-58
// feeds to the decoder, so train-mode batch statistics see the same samples.
//
// Kernels (activations are laid out [slot][pixel][channel], slot = padded sample position):
//   front  (VALU)  pre1 = z . M01 + b01, M01 = conv_in o ConvT1 composed once in fp64 (no
//                  nonlinearity sits between them); per-tile partial BN sums in fp64.
//   mid    (MFMA)  v_mfma_f32_32x32x2_f32 GEMM for ConvT2 as a block-sparse product over
//                  (input pixel -> output pixel) blocks; prologue = norm1 + ReLU on primal and
//                  tangent while staging A into LDS; epilogue = bias, store, partial BN sums.
//   back   (VALU)  norm2 + ReLU, ConvT3, sigmoid', squared norm per sample.
#include "geo_common.h"

#include <cmath>
#include <cstdlib>
#include <type_traits>
#include <vector>

namespace {

constexpr int TS = 32;          // samples per tile (= one 32-row MFMA slice for primal, one for tangent)
constexpr int NC = 128;         // output columns per mid-kernel workgroup (4 waves x 32)
constexpr int MAX_BLOCKS = 4;   // input pixels feeding one output pixel (2x2 input)
constexpr int FRONT_MAX_N1 = 1024;   // columns of the first pre-activation (4 pixels x c1 <= 256 channels)
constexpr int BACK_TS = 8;      // samples per back-kernel workgroup

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned short u16x8 __attribute__((ext_vector_type(8)));

struct Shape {
    int d, c0, c1, c2, co, s_out, pad3, p_out;   // p_out = co * s_out * s_out
    int n1, n2;                                   // 4*c1, 16*c2
    int opix_per_chunk, n_chunks;
};

struct ChunkTable {
    int nblk[16];
    int ipix[16][MAX_BLOCKS];
    int opix[16][16];
};

// prologue constants per (stat row, channel): y = (x - mu) * sc + beta ; t' = sc * (t - mt - (x - mu) * c5)
struct NormConst { float mu, sc, beta, mt, c5; };

// ---------------------------------------------------------------------------------- weight prep
// M01[k][n], n = px*c1 + co, px = oy*2+ox: composition of conv_in and ConvT1 (taps ky=oy+1,kx=ox+1)
__global__ __launch_bounds__(256) void compose_front_kernel(const float *__restrict__ w_in, const float *__restrict__ b_in,
                                                           const float *__restrict__ w1, const float *__restrict__ b1,
                                                           int d, int c0, int c1, float *__restrict__ M01,
                                                           float *__restrict__ b01) {
    const int n1 = 4 * c1;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < (d + 1) * n1; i += gridDim.x * blockDim.x) {
        const int k = i / n1, n = i % n1;
        const int px = n / c1, co = n % c1;
        const int ky = (px >> 1) + 1, kx = (px & 1) + 1;
        double s = 0.0;
        for (int c8 = 0; c8 < c0; c8 += 8) {               // 16 strided loads in flight, then the fma chain in ci order
            float m1[8], a[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int ci = c8 + j < c0 ? c8 + j : c0 - 1;
                m1[j] = w1[(((size_t)ci * c1 + co) * 4 + ky) * 4 + kx];
                a[j] = k < d ? w_in[(size_t)ci * d + k] : b_in[ci];
            }
#pragma unroll
            for (int j = 0; j < 8; ++j)
                if (c8 + j < c0) s = fma((double)a[j], (double)m1[j], s);
        }
        if (k < d) M01[(size_t)k * n1 + n] = (float)s;
        else b01[n] = (float)(s + (double)b1[co]);
    }
}

// B2p[chunk][blk][k = ci][col], col = local output pixel * c2 + co; zero where the tap is out of range
__global__ __launch_bounds__(256) void pack_mid_kernel(const float *__restrict__ w2, int c1, int c2, ChunkTable tab,
                                                      int n_chunks, int opix_per_chunk, float *__restrict__ B2p) {
    const size_t per_blk = (size_t)c1 * NC;
    const size_t total = (size_t)n_chunks * MAX_BLOCKS * per_blk;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int col = (int)(i % NC);
        const int ci = (int)((i / NC) % c1);
        const int blk = (int)((i / per_blk) % MAX_BLOCKS);
        const int ch = (int)(i / (per_blk * MAX_BLOCKS));
        float v = 0.0f;
        const int lo = col / c2, co = col % c2;
        if (blk < tab.nblk[ch] && lo < opix_per_chunk) {
            const int op = tab.opix[ch][lo], ip = tab.ipix[ch][blk];
            const int ky = (op >> 2) + 1 - 2 * (ip >> 1), kx = (op & 3) + 1 - 2 * (ip & 1);
            if (ky >= 0 && ky < 4 && kx >= 0 && kx < 4) v = w2[(((size_t)ci * c2 + co) * 4 + ky) * 4 + kx];
        }
        B2p[i] = v;
    }
}

// W3p[ky][kx][co][ci]
__global__ __launch_bounds__(256) void pack_back_kernel(const float *__restrict__ w3, int c2, int co_n,
                                                       float *__restrict__ W3p) {
    const int total = 16 * co_n * c2;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        const int ci = i % c2, co = (i / c2) % co_n, t = i / (c2 * co_n);
        W3p[i] = w3[(((size_t)ci * co_n + co) * 4 + (t >> 2)) * 4 + (t & 3)];
    }
}

// ---------------------------------------------------------------------------------- front
// One workgroup = one tile of TS sample slots.  Thread owns output columns n = tid + 256*j for
// all TS samples, so per-column sums need no cross-thread reduction.
template <int DMAX>
__global__ __launch_bounds__(256) void front_kernel(const float *__restrict__ z, const int32_t *__restrict__ src,
                                                   const int32_t *__restrict__ dst, const float *__restrict__ z_start,
                                                   const float *__restrict__ z_end, int64_t e_base, int64_t n_edges,
                                                   int batch, int tiles_per_group, int d, int n1,
                                                   const float *__restrict__ M01, const float *__restrict__ b01,
                                                   float *__restrict__ pre, float *__restrict__ tpre,
                                                   double *__restrict__ partial, int want_stats, int unit_d = 0) {
    // unit_d > 0 (per-latent Jacobians, run_node_jacobian): "edge" e is (latent pair e / unit_d, latent dimension e % unit_d);
    // the tangent of both sides is the unit vector of that dimension instead of the latent difference.
    // [latent dimension][sample]: two neighbouring samples of a dimension are one 8-byte broadcast read feeding one packed fma
    __shared__ __attribute__((aligned(8))) float zp[DMAX][TS];
    __shared__ __attribute__((aligned(8))) float dz[DMAX][TS];
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    __shared__ int valid_s[TS];
    __shared__ double ps[FRONT_MAX_N1][4];                  // per-column statistics before the pixel reduction
    const int tile = blockIdx.x;
    const int group = tile / tiles_per_group, tg = tile % tiles_per_group;
    const int chunk = group >> 1, side = group & 1;
    for (int i = threadIdx.x; i < TS * d; i += 256) {
        const int s = i / d, k = i % d;
        const int within = tg * TS + s;
        const int64_t e = e_base + (int64_t)chunk * batch + within;
        float a = 0.f, b = 0.f;
        const bool ok = within < batch && e < n_edges;
        if (ok) {
            if (src) {
                a = z[(int64_t)src[e] * d + k];
                b = z[(int64_t)dst[e] * d + k];
            } else {
                a = z_start[e * d + k];
                b = z_end[e * d + k];
            }
        }
        zp[k][s] = side == 0 ? a : b;
        dz[k][s] = unit_d ? ((ok && k == (int)(e % unit_d)) ? 1.f : 0.f) : b - a;
        if (k == 0) valid_s[s] = ok ? 1 : 0;
    }
    for (int i = threadIdx.x; i < TS * (DMAX - d); i += 256) {      // padded latent columns: zeros, not stale LDS
        const int s = i / (DMAX - d), k = d + i % (DMAX - d);
        zp[k][s] = 0.f;
        dz[k][s] = 0.f;
    }
    __syncthreads();
    const size_t slot0 = (size_t)tile * TS;
    for (int n = threadIdx.x; n < n1; n += 256) {
        float m[DMAX];
#pragma unroll
        for (int k = 0; k < DMAX; ++k) m[k] = k < d ? M01[(size_t)k * n1 + n] : 0.f;
        const float bias = b01[n];
        double sx = 0, sxx = 0, st = 0, sxt = 0;
        for (int s = 0; s < TS; s += 2) {                      // two samples per packed fma (v_pk_fma_f32): the same fmaf chains
            f32x2 x2 = {bias, bias}, t2 = {0.f, 0.f};
#pragma unroll
            for (int k = 0; k < DMAX; ++k) {
                const f32x2 mk = {m[k], m[k]};
                x2 = __builtin_elementwise_fma(*reinterpret_cast<const f32x2 *>(&zp[k][s]), mk, x2);
                t2 = __builtin_elementwise_fma(*reinterpret_cast<const f32x2 *>(&dz[k][s]), mk, t2);
            }
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const float x = u ? x2.y : x2.x, t = u ? t2.y : t2.x;
                pre[(slot0 + s + u) * n1 + n] = x;
                tpre[(slot0 + s + u) * n1 + n] = t;
                if (valid_s[s + u]) {
                    sx += x; sxx += (double)x * x; st += t; sxt += (double)x * t;
                }
            }
        }
        if (want_stats) { ps[n][0] = sx; ps[n][1] = sxx; ps[n][2] = st; ps[n][3] = sxt; }
    }
    if (want_stats) {                                       // the four pixels of a channel, in pixel order: [tile][c1][4]
        __syncthreads();
        const int c1 = n1 / 4;
        for (int c = threadIdx.x; c < c1; c += 256) {
            double a0 = 0, a1 = 0, a2 = 0, a3 = 0;
#pragma unroll
            for (int px = 0; px < 4; ++px) { a0 += ps[px * c1 + c][0]; a1 += ps[px * c1 + c][1]; a2 += ps[px * c1 + c][2]; a3 += ps[px * c1 + c][3]; }
            double *p = partial + ((size_t)tile * c1 + c) * 4;
            p[0] = a0; p[1] = a1; p[2] = a2; p[3] = a3;
        }
    }
}

// The same product on the float32 matrix cores for wide latents (d > 16: the VALU kernel spends 2 x d FMAs per value,
// 7.3 ms of the 50 000 x 64 configuration's step).  v_mfma_f32_32x32x2_f32 is a k-ordered fmaf chain into the accumulator,
// which starts at the bias: bit for bit the chain of front_kernel (x = bias; x = fmaf(z_k, m_k, x), k ascending).
// One wave = 32 samples x 32 columns at a time (primal and tangent accumulators), a workgroup = 4 waves over the n1 / 32
// column tiles; A (latents, differences) from LDS, B (M01) from L2; statistics from the accumulators (lane = column).
template <int DMAX>
__global__ __launch_bounds__(256) void front_mfma_kernel(const float *__restrict__ z, const int32_t *__restrict__ src,
                                                        const int32_t *__restrict__ dst, const float *__restrict__ z_start,
                                                        const float *__restrict__ z_end, int64_t e_base, int64_t n_edges,
                                                        int batch, int tiles_per_group, int d, int n1,
                                                        const float *__restrict__ M01, const float *__restrict__ b01,
                                                        float *__restrict__ pre, float *__restrict__ tpre,
                                                        double *__restrict__ partial, int want_stats) {
    __shared__ float zp[TS][DMAX + 1];
    __shared__ float dz[TS][DMAX + 1];
    __shared__ int valid_s[TS];
    __shared__ double ps[FRONT_MAX_N1][4];
    const int tile = blockIdx.x;
    const int group = tile / tiles_per_group, tg = tile % tiles_per_group;
    const int chunk = group >> 1, side = group & 1;
    for (int i = threadIdx.x; i < TS * DMAX; i += 256) {
        const int s = i / DMAX, k = i % DMAX;
        const int within = tg * TS + s;
        const int64_t e = e_base + (int64_t)chunk * batch + within;
        float a = 0.f, b = 0.f;
        const bool ok = within < batch && e < n_edges;
        if (ok && k < d) {
            if (src) {
                a = z[(int64_t)src[e] * d + k];
                b = z[(int64_t)dst[e] * d + k];
            } else {
                a = z_start[e * d + k];
                b = z_end[e * d + k];
            }
        }
        zp[s][k] = side == 0 ? a : b;
        dz[s][k] = b - a;
        if (k == 0) valid_s[s] = ok ? 1 : 0;
    }
    __syncthreads();
    const size_t slot0 = (size_t)tile * TS;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = lane & 31, h = lane >> 5;
    for (int ct = wave; ct * 32 < n1; ct += 4) {
        const int n = ct * 32 + r;                              // this lane's column
        f32x16 accp, acct;
        const float bias = b01[n];
#pragma unroll
        for (int q = 0; q < 16; ++q) { accp[q] = bias; acct[q] = 0.f; }
#pragma unroll 8
        for (int i = 0; i < DMAX / 2; ++i) {
            const int k = 2 * i + h;
            const float bm = k < d ? M01[(size_t)k * n1 + n] : 0.f;
            accp = __builtin_amdgcn_mfma_f32_32x32x2f32(zp[r][k], bm, accp, 0, 0, 0);
            acct = __builtin_amdgcn_mfma_f32_32x32x2f32(dz[r][k], bm, acct, 0, 0, 0);
        }
        double sx = 0, sxx = 0, st = 0, sxt = 0;
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const int row = (q & 3) + 8 * (q >> 2) + 4 * h;
            const float x = accp[q], t = acct[q];
            pre[(slot0 + row) * n1 + n] = x;
            tpre[(slot0 + row) * n1 + n] = t;
            if (valid_s[row]) { sx += x; sxx += (double)x * x; st += t; sxt += (double)x * t; }
        }
        if (want_stats) {
            sx += __shfl_xor(sx, 32, 64); sxx += __shfl_xor(sxx, 32, 64);
            st += __shfl_xor(st, 32, 64); sxt += __shfl_xor(sxt, 32, 64);
            if (lane < 32) { ps[n][0] = sx; ps[n][1] = sxx; ps[n][2] = st; ps[n][3] = sxt; }
        }
    }
    if (want_stats) {                                       // the four pixels of a channel, in pixel order: [tile][c1][4]
        __syncthreads();
        const int c1 = n1 / 4;
        for (int c = threadIdx.x; c < c1; c += 256) {
            double a0 = 0, a1 = 0, a2 = 0, a3 = 0;
#pragma unroll
            for (int px = 0; px < 4; ++px) { a0 += ps[px * c1 + c][0]; a1 += ps[px * c1 + c][1]; a2 += ps[px * c1 + c][2]; a3 += ps[px * c1 + c][3]; }
            double *p = partial + ((size_t)tile * c1 + c) * 4;
            p[0] = a0; p[1] = a1; p[2] = a2; p[3] = a3;
        }
    }
}

// ---------------------------------------------------------------------------------- norm constants
// mode 1 (batch statistics): reduce the per-tile partial sums of a group over tiles and pixels.
// partial: [tile][npx_stored*C][4] fp64 (npx_stored = 1 when the producer summed a tile's pixels).  consts: [group][C].
__global__ __launch_bounds__(256) void finalize_batch_kernel(const double *__restrict__ partial, int tiles_per_group,
                                                            int npx_stored, int npx, int C, int64_t e_base, int64_t n_edges,
                                                            int batch, const float *__restrict__ gamma,
                                                            const float *__restrict__ beta, float eps,
                                                            NormConst *__restrict__ consts, int n_groups,
                                                            float2 *__restrict__ stats_out) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n_groups * C; i += gridDim.x * blockDim.x) {
        const int g = i / C, c = i % C;
        const int chunk = g >> 1;
        int64_t cnt = n_edges - (e_base + (int64_t)chunk * batch);
        if (cnt > batch) cnt = batch;
        if (cnt < 0) cnt = 0;
        double sx = 0, sxx = 0, st = 0, sxt = 0;
        for (int t = 0; t < tiles_per_group; ++t)
            for (int px = 0; px < npx_stored; ++px) {      // (1 when the producing kernel already summed a tile's pixels)
                const double *p = partial + (((size_t)(g * tiles_per_group + t)) * npx_stored * C + (size_t)px * C + c) * 4;
                sx += p[0]; sxx += p[1]; st += p[2]; sxt += p[3];
            }
        const double n = (double)cnt * npx;
        NormConst k;
        if (n > 0) {
            const double mu = sx / n;
            double var = sxx / n - mu * mu;
            if (var < 0) var = 0;
            const double inv = 1.0 / sqrt(var + (double)eps);
            const double mt = st / n;
            const double mxt = (sxt - mu * st) * inv / n;      // mean(xhat * t)
            k.mu = (float)mu; k.sc = (float)(inv * gamma[c]); k.beta = beta[c];
            k.mt = (float)mt; k.c5 = (float)(inv * mxt);
            // batch mean and UNBIASED variance, what torch folds into running_mean / running_var (n = 1: no update)
            if (stats_out) stats_out[i] = make_float2((float)mu, n > 1 ? (float)(var * n / (n - 1.0)) : -1.0f);
        } else {
            k.mu = 0; k.sc = 0; k.beta = 0; k.mt = 0; k.c5 = 0;
            if (stats_out) stats_out[i] = make_float2(0.f, -1.0f);
        }
        consts[i] = k;
    }
}

// BatchNorm2d in training mode also folds every batch into its running statistics (momentum m):
//   running = (1 - m) * running + m * batch_stat, once per forward call, i.e. per (chunk, endpoint side) group IN ORDER
// (riemannian_metric.py:57-58 calls the decoder for the start side, then the end side of each chunk).  One thread per
// channel walks the groups of a pass sequentially (the rounding of every step is torch's).
__global__ __launch_bounds__(256) void running_update_kernel(const float2 *__restrict__ stats, int n_groups, int C, float m,
                                                            float *__restrict__ running_mean, float *__restrict__ running_var) {
    // One workgroup per 8 channels (one CU reading all 3.8 MB of a 60 000-latent build's statistics took 0.46 ms): all
    // threads stage blocks of 512 groups x 8 channels in LDS, then 8 threads apply the block's updates in order.
    constexpr int CH = 8, GB = 512;
    __shared__ float2 blk[GB * CH];                        // 32 KB
    const int c0 = blockIdx.x * CH;
    const int nc = C - c0 < CH ? C - c0 : CH;
    float rm = 0.f, rv = 0.f;
    if ((int)threadIdx.x < nc) { rm = running_mean[c0 + threadIdx.x]; rv = running_var[c0 + threadIdx.x]; }
    for (int g0 = 0; g0 < n_groups; g0 += GB) {
        const int ng = n_groups - g0 < GB ? n_groups - g0 : GB;
        float2 ld[GB * CH / 256];                           // all 16 loads of the thread in flight together
#pragma unroll
        for (int k = 0; k < GB * CH / 256; ++k) {
            const int i = k * 256 + threadIdx.x, g = i / CH, c = i % CH;
            ld[k] = (g < ng && c < nc) ? stats[(size_t)(g0 + g) * C + c0 + c] : make_float2(0.f, -1.f);
        }
#pragma unroll
        for (int k = 0; k < GB * CH / 256; ++k) blk[k * 256 + threadIdx.x] = ld[k];
        __syncthreads();
        if ((int)threadIdx.x < nc) {
            for (int g = 0; g < ng; g += 8) {              // 8 LDS reads in flight, then their dependent updates
                float2 st[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) st[j] = blk[(g + j) * CH + threadIdx.x];    // (rows past ng hold the skip mark)
#pragma unroll
                for (int j = 0; j < 8; ++j) {              // (selects, not branches: the chain is the critical path)
                    const bool live = st[j].y >= 0.f;      // else: empty or single-element batch (or past the end)
                    const float nm = (1.0f - m) * rm + m * st[j].x, nv = (1.0f - m) * rv + m * st[j].y;
                    rm = live ? nm : rm;
                    rv = live ? nv : rv;
                }
            }
        }
        __syncthreads();
    }
    if ((int)threadIdx.x < nc) { running_mean[c0 + threadIdx.x] = rm; running_var[c0 + threadIdx.x] = rv; }
}

// mode 0 (none) / running statistics: one row of constants shared by every group.
__global__ void finalize_fixed_kernel(int C, int norm, const float *__restrict__ gamma, const float *__restrict__ beta,
                                      const float *__restrict__ rm, const float *__restrict__ rv, float eps,
                                      NormConst *__restrict__ consts) {
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
        NormConst k;
        if (norm == 1) {
            const double inv = 1.0 / sqrt((double)rv[c] + (double)eps);
            k.mu = rm[c]; k.sc = (float)(inv * gamma[c]); k.beta = beta[c];
        } else if (norm == 2) {                               // GroupNorm: per-channel affine only, statistics per sample
            k.mu = 0.f; k.sc = gamma[c]; k.beta = beta[c];
        } else {
            k.mu = 0.f; k.sc = 1.f; k.beta = 0.f;
        }
        k.mt = 0.f; k.c5 = 0.f;
        consts[c] = k;
    }
}

__device__ __forceinline__ void norm_relu(const NormConst &k, float x, float t, float *a, float *ta) {
    const float xc = x - k.mu;
    const float y = fmaf(xc, k.sc, k.beta);
    const float tt = k.sc * (t - k.mt - xc * k.c5);
    *a = y > 0.f ? y : 0.f;
    *ta = y > 0.f ? tt : 0.f;
}

// GroupNorm (spatial_vae.py:13-16): statistics per (sample, group) over the group's channels and all pixels.
// g = {mean, 1/sqrt(var+eps), mean(t), inv * mean(xhat * t)}; gamma/beta per channel.
__device__ __forceinline__ void norm_relu_gn(const float4 &g, float gamma, float beta, float x, float t, float *a,
                                             float *ta) {
    const float xc = x - g.x;
    const float sc = g.y * gamma;
    const float y = fmaf(xc, sc, beta);
    const float tt = sc * (t - g.z - xc * g.w);
    *a = y > 0.f ? y : 0.f;
    *ta = y > 0.f ? tt : 0.f;
}

// pre / tpre: [slot][npx][C]; out: [slot][G].  fp64 accumulation, rounded once.
// `prim_row` (per-node primal): the primal row of slot s is pre[prim_row[s]] (its latent's row), the tangent row stays tpre[s].
__global__ __launch_bounds__(256) void group_stats_kernel(const float *__restrict__ pre, const float *__restrict__ tpre,
                                                         int64_t n_slots, int npx, int C, int G, float eps,
                                                         float4 *__restrict__ out, const int32_t *__restrict__ prim_row = nullptr) {
    const int cpg = C / G;
    const int64_t total = n_slots * G;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int g = (int)(i % G);
        const int64_t slot = i / G;
        const float *x = pre + (size_t)(prim_row ? prim_row[slot] : slot) * npx * C + (size_t)g * cpg;
        const float *t = tpre + (size_t)slot * npx * C + (size_t)g * cpg;
        double sx = 0.0, sxx = 0.0, st = 0.0, sxt = 0.0;
        for (int px = 0; px < npx; ++px)
            for (int k = 0; k < cpg; ++k) {
                const double xv = (double)x[(size_t)px * C + k], tv = (double)t[(size_t)px * C + k];
                sx += xv; sxx += xv * xv; st += tv; sxt += xv * tv;
            }
        const double n = (double)npx * cpg;
        const double mu = sx / n;
        double var = sxx / n - mu * mu;
        if (var < 0) var = 0;
        const double inv = 1.0 / sqrt(var + (double)eps);
        const double mt = st / n;
        const double mxt = (sxt - mu * st) * inv / n;
        out[i] = make_float4((float)mu, (float)inv, (float)mt, (float)(inv * mxt));
    }
}

// ---------------------------------------------------------------------------------- mid (MFMA)
// grid = (tiles, chunks).  LDS: A_p / A_t [TS][c1+1] f32 for the current input pixel.
// Wave w computes columns [32w, 32w+32) of the chunk for the 32 primal and the 32 tangent rows.
template <int C1>
__global__ __launch_bounds__(256) void mid_kernel(const float *__restrict__ pre1, const float *__restrict__ tpre1,
                                                 const NormConst *__restrict__ consts1, int consts_per_group,
                                                 int tiles_per_group, ChunkTable tab, int opix_per_chunk, int c2,
                                                 const float *__restrict__ B2p, const float *__restrict__ b2,
                                                 float *__restrict__ pre2, float *__restrict__ tpre2,
                                                 double *__restrict__ partial2, int want_stats,
                                                 const int32_t *__restrict__ slot_valid) {
    constexpr int LDA = C1 + 1;
    __shared__ float Ap[TS * LDA];
    __shared__ float At[TS * LDA];
    __shared__ NormConst kc[C1];
    const int tile = blockIdx.x, chunk = blockIdx.y;
    const int group = tile / tiles_per_group;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int n1 = 4 * C1, n2 = 16 * c2;
    const size_t slot0 = (size_t)tile * TS;
    for (int c = threadIdx.x; c < C1; c += 256) kc[c] = consts1[(size_t)(consts_per_group ? group : 0) * C1 + c];

    f32x16 accp, acct;
#pragma unroll
    for (int i = 0; i < 16; ++i) { accp[i] = 0.f; acct[i] = 0.f; }

    const int nblk = tab.nblk[chunk];
    for (int blk = 0; blk < nblk; ++blk) {
        const int ip = tab.ipix[chunk][blk];
        __syncthreads();                       // previous block's MFMA reads are done (and kc is visible)
        // stage A: thread -> (sample = tid/8, 16 consecutive channels)
        {
            const int s = threadIdx.x >> 3, k0 = (threadIdx.x & 7) * (C1 / 8);
            const float *xp = pre1 + (slot0 + s) * n1 + (size_t)ip * C1 + k0;
            const float *xt = tpre1 + (slot0 + s) * n1 + (size_t)ip * C1 + k0;
#pragma unroll
            for (int k = 0; k < C1 / 8; ++k) {
                float a, ta;
                norm_relu(kc[k0 + k], xp[k], xt[k], &a, &ta);
                Ap[s * LDA + k0 + k] = a;
                At[s * LDA + k0 + k] = ta;
            }
        }
        // B fragment of this block: lane holds B[k = 2*st + (lane>>5)][col = 32*wave + (lane&31)]
        float breg[C1 / 2];
        const float *bsrc = B2p + (((size_t)chunk * MAX_BLOCKS + blk) * C1 + (lane >> 5)) * NC + wave * 32 + (lane & 31);
#pragma unroll
        for (int st = 0; st < C1 / 2; ++st) breg[st] = bsrc[(size_t)st * 2 * NC];
        __syncthreads();
        const float *ap = Ap + (lane & 31) * LDA + (lane >> 5);
        const float *at = At + (lane & 31) * LDA + (lane >> 5);
#pragma unroll
        for (int st = 0; st < C1 / 2; ++st) {
            accp = __builtin_amdgcn_mfma_f32_32x32x2f32(ap[2 * st], breg[st], accp, 0, 0, 0);
            acct = __builtin_amdgcn_mfma_f32_32x32x2f32(at[2 * st], breg[st], acct, 0, 0, 0);
        }
    }

    // epilogue: C[row = (r&3) + 8*(r>>2) + 4*(lane>>5)][col = lane&31]
    const int col = wave * 32 + (lane & 31);
    const int lo = col / c2, co = col % c2;
    const bool col_ok = lo < opix_per_chunk;
    const int op = col_ok ? tab.opix[chunk][lo] : 0;
    const float bias = col_ok ? b2[co] : 0.f;
    double sx = 0, sxx = 0, st_ = 0, sxt = 0;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        const float x = accp[r] + bias, t = acct[r];
        if (col_ok) {
            pre2[(slot0 + row) * n2 + (size_t)op * c2 + co] = x;
            tpre2[(slot0 + row) * n2 + (size_t)op * c2 + co] = t;
            if (slot_valid[slot0 + row]) { sx += x; sxx += (double)x * x; st_ += t; sxt += (double)x * t; }
        }
    }
    if (want_stats) {
        sx += __shfl_xor(sx, 32, 64); sxx += __shfl_xor(sxx, 32, 64);
        st_ += __shfl_xor(st_, 32, 64); sxt += __shfl_xor(sxt, 32, 64);
        if (lane < 32 && col_ok) {
            double *p = partial2 + ((size_t)tile * n2 + (size_t)op * c2 + co) * 4;
            p[0] = sx; p[1] = sxx; p[2] = st_; p[3] = sxt;
        }
    }
}

// ---------------------------------------------------------------------------------- mid, bf16 x 3 split
// The f32 MFMA runs at 1/16 of the bf16 rate.  Every f32 operand is split exactly into three bf16 parts
// (a = a1 + a2 + a3, each part the truncated leading 8 mantissa bits of the remainder) and the product is
// formed as a1b1 + a1b2 + a2b1 + a2b2 + a1b3 + a3b1 with f32 accumulation inside v_mfma_f32_32x32x16_bf16:
// the dropped terms are below 2^-23 |a||b|, the level of one f32 rounding, for 6/16 of the matrix time.
__device__ __forceinline__ void split3(float a, unsigned short &p1, unsigned short &p2, unsigned short &p3) {
    const unsigned b1 = __float_as_uint(a) & 0xffff0000u;
    const float r1 = a - __uint_as_float(b1);                 // exact
    const unsigned b2 = __float_as_uint(r1) & 0xffff0000u;
    const float r2 = r1 - __uint_as_float(b2);                // exact
    p1 = (unsigned short)(b1 >> 16);
    p2 = (unsigned short)(b2 >> 16);
    p3 = (unsigned short)(__float_as_uint(r2) >> 16);
}

// the same split for two values, the parts packed pairwise (low half = first value): 3 byte permutes instead of shifts and ors
__device__ __forceinline__ void split3_pair(float a0, float a1, unsigned &w1, unsigned &w2, unsigned &w3) {
    const unsigned u0 = __float_as_uint(a0), u1 = __float_as_uint(a1);
    w1 = __builtin_amdgcn_perm(u1, u0, 0x07060302u);
    const float r0 = a0 - __uint_as_float(u0 & 0xffff0000u), r1 = a1 - __uint_as_float(u1 & 0xffff0000u);   // exact
    const unsigned v0 = __float_as_uint(r0), v1 = __float_as_uint(r1);
    w2 = __builtin_amdgcn_perm(v1, v0, 0x07060302u);
    const float q0 = r0 - __uint_as_float(v0 & 0xffff0000u), q1 = r1 - __uint_as_float(v1 & 0xffff0000u);   // exact
    w3 = __builtin_amdgcn_perm(__float_as_uint(q1), __float_as_uint(q0), 0x07060302u);
}

// B3[chunk][blk][ks][h][part][col][8] bf16: lane (col, h) of the 32x32x16 B operand reads 16 contiguous bytes; the three parts of
// a (block, k-step, half) lie 2 KB apart -- one address register and the immediate offsets -2048 / 0 / +2048 reach all three
__global__ __launch_bounds__(256) void pack_mid_bf16_kernel(const float *__restrict__ B2p, int c1, int n_chunks,
                                                           unsigned short *__restrict__ B3) {
    const size_t per_blk = (size_t)c1 * NC;
    const size_t total = (size_t)n_chunks * MAX_BLOCKS * per_blk;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int col = (int)(i % NC);
        const int k = (int)((i / NC) % c1);
        const size_t cb = i / per_blk;                        // chunk * MAX_BLOCKS + blk
        unsigned short p[3];
        split3(B2p[i], p[0], p[1], p[2]);
        const int ks = k >> 4, h = (k >> 3) & 1, j = k & 7;
        for (int part = 0; part < 3; ++part)
            B3[(((cb * (c1 / 16) + ks) * 2 + h) * 3 + part) * (size_t)NC * 8 + (size_t)col * 8 + j] = p[part];
    }
}

template <int C1>
__global__ __launch_bounds__(256) void mid_bf16_kernel(const float *__restrict__ pre1, const float *__restrict__ tpre1,
                                                      const NormConst *__restrict__ consts1, int consts_per_group,
                                                      int tiles_per_group, ChunkTable tab, int opix_per_chunk, int c2,
                                                      const unsigned short *__restrict__ B3, const float *__restrict__ b2,
                                                      float *__restrict__ pre2, float *__restrict__ tpre2,
                                                      double *__restrict__ partial2, int want_stats,
                                                      const int32_t *__restrict__ slot_valid) {
    constexpr int LDK = C1 + 8;                               // bf16 elements per row (+16 B pad)
    constexpr int KS = C1 / 16;                               // 16-deep MFMA steps per input pixel
    __shared__ __attribute__((aligned(16))) unsigned short A3[3][2][TS][LDK];   // [part][primal|tangent][row][k]
    __shared__ NormConst kc[C1];
    const int tile = blockIdx.x, chunk = blockIdx.y;
    const int group = tile / tiles_per_group;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int n1 = 4 * C1, n2 = 16 * c2;
    const size_t slot0 = (size_t)tile * TS;
    for (int c = threadIdx.x; c < C1; c += 256) kc[c] = consts1[(size_t)(consts_per_group ? group : 0) * C1 + c];

    f32x16 accp, acct;
#pragma unroll
    for (int i = 0; i < 16; ++i) { accp[i] = 0.f; acct[i] = 0.f; }

    const int nblk = tab.nblk[chunk];
    const int r = lane & 31, h = lane >> 5;
    for (int blk = 0; blk < nblk; ++blk) {
        const int ip = tab.ipix[chunk][blk];
        __syncthreads();
        {   // stage A: thread -> (sample = tid/8, C1/8 consecutive channels); norm1 + ReLU, then the 3-way split
            const int s = threadIdx.x >> 3, k0 = (threadIdx.x & 7) * (C1 / 8);
            const float *xp = pre1 + (slot0 + s) * n1 + (size_t)ip * C1 + k0;
            const float *xt = tpre1 + (slot0 + s) * n1 + (size_t)ip * C1 + k0;
#pragma unroll
            for (int k = 0; k < C1 / 8; ++k) {
                float a, ta;
                norm_relu(kc[k0 + k], xp[k], xt[k], &a, &ta);
                split3(a, A3[0][0][s][k0 + k], A3[1][0][s][k0 + k], A3[2][0][s][k0 + k]);
                split3(ta, A3[0][1][s][k0 + k], A3[1][1][s][k0 + k], A3[2][1][s][k0 + k]);
            }
        }
        // B fragments of this block, all three parts: lane (col r of the wave's 32, half h)
        bf16x8 b[3][KS];
        const unsigned short *bsrc = B3 + ((size_t)(chunk * MAX_BLOCKS + blk) * 3 * KS * 2) * (size_t)NC * 8;
#pragma unroll
        for (int part = 0; part < 3; ++part)
#pragma unroll
            for (int ks = 0; ks < KS; ++ks)
                b[part][ks] = *reinterpret_cast<const bf16x8 *>(
                    bsrc + (((size_t)ks * 2 + h) * 3 + part) * (size_t)NC * 8 + (size_t)(wave * 32 + r) * 8);
        __syncthreads();
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            bf16x8 ap[3], at[3];
#pragma unroll
            for (int part = 0; part < 3; ++part) {
                ap[part] = *reinterpret_cast<const bf16x8 *>(&A3[part][0][r][ks * 16 + h * 8]);
                at[part] = *reinterpret_cast<const bf16x8 *>(&A3[part][1][r][ks * 16 + h * 8]);
            }
            // smallest terms first
            accp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ap[2], b[0][ks], accp, 0, 0, 0);
            acct = __builtin_amdgcn_mfma_f32_32x32x16_bf16(at[2], b[0][ks], acct, 0, 0, 0);
            accp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ap[0], b[2][ks], accp, 0, 0, 0);
            acct = __builtin_amdgcn_mfma_f32_32x32x16_bf16(at[0], b[2][ks], acct, 0, 0, 0);
            accp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ap[1], b[1][ks], accp, 0, 0, 0);
            acct = __builtin_amdgcn_mfma_f32_32x32x16_bf16(at[1], b[1][ks], acct, 0, 0, 0);
            accp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ap[1], b[0][ks], accp, 0, 0, 0);
            acct = __builtin_amdgcn_mfma_f32_32x32x16_bf16(at[1], b[0][ks], acct, 0, 0, 0);
            accp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ap[0], b[1][ks], accp, 0, 0, 0);
            acct = __builtin_amdgcn_mfma_f32_32x32x16_bf16(at[0], b[1][ks], acct, 0, 0, 0);
            accp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ap[0], b[0][ks], accp, 0, 0, 0);
            acct = __builtin_amdgcn_mfma_f32_32x32x16_bf16(at[0], b[0][ks], acct, 0, 0, 0);
        }
    }

    // epilogue: C[row = (r&3) + 8*(r>>2) + 4*(lane>>5)][col = lane&31]  (same map as the f32 MFMA)
    const int col = wave * 32 + (lane & 31);
    const int lo = col / c2, co = col % c2;
    const bool col_ok = lo < opix_per_chunk;
    const int op = col_ok ? tab.opix[chunk][lo] : 0;
    const float bias = col_ok ? b2[co] : 0.f;
    double sx = 0, sxx = 0, st_ = 0, sxt = 0;
#pragma unroll
    for (int q = 0; q < 16; ++q) {
        const int row = (q & 3) + 8 * (q >> 2) + 4 * (lane >> 5);
        const float x = accp[q] + bias, t = acct[q];
        if (col_ok) {
            pre2[(slot0 + row) * n2 + (size_t)op * c2 + co] = x;
            tpre2[(slot0 + row) * n2 + (size_t)op * c2 + co] = t;
            if (slot_valid[slot0 + row]) { sx += x; sxx += (double)x * x; st_ += t; sxt += (double)x * t; }
        }
    }
    if (want_stats) {
        sx += __shfl_xor(sx, 32, 64); sxx += __shfl_xor(sxx, 32, 64);
        st_ += __shfl_xor(st_, 32, 64); sxt += __shfl_xor(sxt, 32, 64);
        if (lane < 32 && col_ok) {
            double *p = partial2 + ((size_t)tile * n2 + (size_t)op * c2 + co) * 4;
            p[0] = sx; p[1] = sxx; p[2] = st_; p[3] = sxt;
        }
    }
}

// Workgroup barrier that orders LDS traffic only.  __syncthreads() also waits for every outstanding global load and
// store of the wave (s_waitcnt vmcnt(0)): that drains the prefetched pre-activations of the next input pixel
// (and, in a persistent kernel, the epilogue's stores) at every phase.  Nothing here communicates through global memory inside a launch.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// ---- static geometry of ConvT2 (k4 s2 p1, 2x2 -> 4x4) in 8 chunks of two output pixels -- what make_chunks builds for
// n_chunks == 8, opix_per_chunk == 2 (the host checks its table against these before taking the mid_all path) -------------
//   chunk : output pixels      input pixels (block order)
//     0   : (0,1) (0,2)        0 1          4 : (1,1) (1,2)   0 1 2 3
//     1   : (3,1) (3,2)        2 3          5 : (2,1) (2,2)   0 1 2 3
//     2   : (1,0) (2,0)        0 2          6 : (0,0) (0,3)   0 1
//     3   : (1,3) (2,3)        1 3          7 : (3,0) (3,3)   2 3
// Wave group 0 owns chunks {0,3,4,7}, group 1 {1,2,5,6}: ten (input pixel, chunk) products each, 2 or 3 per input pixel.
struct MidGeom {
    static __host__ __device__ constexpr int chunk_of(int wg, int lc) {
        constexpr int t[2][4] = {{0, 3, 4, 7}, {1, 2, 5, 6}};
        return t[wg][lc];
    }
    static __host__ __device__ constexpr int opix(int ch, int lo) {
        constexpr int t[8][2] = {{1, 2}, {13, 14}, {4, 8}, {7, 11}, {5, 6}, {9, 10}, {0, 3}, {12, 15}};
        return t[ch][lo];
    }
    // position of input pixel ip in the chunk's block list, -1 = the chunk does not read it
    static __host__ __device__ constexpr int blk_of(int ch, int ip) {
        constexpr int t[8][4] = {{0, 1, -1, -1}, {-1, -1, 0, 1}, {0, -1, 1, -1}, {-1, 0, -1, 1},
                                 {0, 1, 2, 3},   {0, 1, 2, 3},   {0, 1, -1, -1}, {-1, -1, 0, 1}};
        return t[ch][ip];
    }
    static __host__ __device__ constexpr int n_prod(int wg, int ip) {
        int n = 0;
        for (int lc = 0; lc < 4; ++lc) n += blk_of(chunk_of(wg, lc), ip) >= 0 ? 1 : 0;
        return n;
    }
    static __host__ __device__ constexpr int prod_lc(int wg, int ip, int j) {   // j-th chunk slot of the group that reads pixel ip
        int n = 0;
        for (int lc = 0; lc < 4; ++lc)
            if (blk_of(chunk_of(wg, lc), ip) >= 0) { if (n == j) return lc; ++n; }
        return 0;
    }
    static __host__ __device__ constexpr int prod_cb(int wg, int ip, int j) {   // chunk * MAX_BLOCKS + block of that product
        const int ch = chunk_of(wg, prod_lc(wg, ip, j));
        return ch * MAX_BLOCKS + blk_of(ch, ip);
    }
    // The input pixels are processed in the order 3, 2, 1, 0: group 0 then ends a tile with a two-product pixel and group 1 starts it
    // with one -- the light pixel shares its interval with the group's epilogue in mid_pipe_kernel (19 900 instead of 23 000
    // cycles for those two intervals).  Both ConvT2 kernels use this order, so their pre-activations stay bit-identical.
    static __host__ __device__ constexpr int pix(int pos) { return 3 - pos; }
    // ring slot of the first step of the pixel at position pos: steps are numbered through the four pixels of a tile, slot = step % 3
    static __host__ __device__ constexpr int ring_off(int wg, int pos, int ks_per_pixel) {
        int n = 0;
        for (int i = 0; i < pos; ++i) n += n_prod(wg, pix(i)) * ks_per_pixel;
        return n % 3;
    }
};

// The products of ONE input pixel for one wave: steps s = (k-step ks, product j), ks outer.  The A fragments of a k-step (three
// split parts, primal and tangent: six 16-byte LDS reads) are read ONCE and feed every product of the pixel (2 or 3 chunks) --
// the chunk-outer order of rounds 1-3 read them again per chunk -- and the NEXT k-step's are read while this one multiplies
// (two register sets).  The B fragments (weights: L2 -> registers, 3 x 16 bytes per lane and step) run through a ring of three
// register sets, issued two steps (24 MFMAs, ~770 cycles) before their use; the first two steps of the NEXT pixel are issued
// before this pixel's last MFMAs so that the staging phase / barrier between pixels covers their latency.
// Accumulation order per accumulator is unchanged (pixel, k-step, the six split products): results are bit-identical.
template <int C1, bool TONLY, int WG, int POS>
__device__ __forceinline__ void mid_products(const unsigned short *__restrict__ a3, unsigned a_idx,   // &A3[0]..., element index of [buf][0][0][r][h * 8]
                                             __amdgpu_buffer_rsrc_t b_rsrc,                 // descriptor of B3 (wave-uniform)
                                             unsigned b_lane,                               // this lane's BYTE offset inside a triple
                                             f32x16 (&accp)[4], f32x16 (&acct)[4], bf16x8 (&ring)[3][3]) {
    constexpr int KS = C1 / 16, LDK = C1 + 8;
    constexpr int IP = MidGeom::pix(POS);                         // the input pixel at this position of the tile's sequence
    constexpr int NP = MidGeom::n_prod(WG, IP);
    constexpr int NSTEP = KS * NP;
    constexpr int OFF = MidGeom::ring_off(WG, POS, KS);           // ring slot of this pixel's step 0
    constexpr ptrdiff_t PART = (ptrdiff_t)NC * 8;                 // elements between the parts of a fragment triple
    constexpr size_t A_PT = (size_t)TS * LDK, A_PART = 2 * A_PT;  // A3[buf][part][primal|tangent][row][k]
    auto b_load = [&](int wg, int ip, int s, bf16x8 (&dst)[3]) {
        const int ks = s / MidGeom::n_prod(wg, ip), j = s % MidGeom::n_prod(wg, ip);
        // buffer loads: the descriptor and the block's byte offset are scalar, ONE per-lane 32-bit offset register serves every
        // load of the kernel -- no vector address arithmetic, no 64-bit address pairs
        const int soff = (int)((((ptrdiff_t)MidGeom::prod_cb(wg, ip, j) * KS + ks) * 2) * 3 * PART * 2);
        dst[0] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(b_rsrc, b_lane, soff, 0));
        dst[1] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(b_rsrc, b_lane, soff + (int)(PART * 2), 0));
        dst[2] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(b_rsrc, b_lane, soff + (int)(PART * 4), 0));
    };
    // ONE register set for the A fragments.  Within a step the six split products of an accumulator run in the order
    // a3 b1, a2 b2, a2 b1, a1 b3, a1 b2, a1 b1 (small terms first), so a3 is free after the first pair of MFMAs of a k-step's
    // LAST product, a2 after the third, a1 after the sixth: each part of the NEXT k-step is read into the registers its
    // predecessor just left, at least six MFMAs (190 cycles) before its first use.
    bf16x8 ap[3], at[3];
    // (the index passes through an empty asm: the compiler can then neither fold it with the buffer's constant into offsets beyond
    // the 16-bit immediate of ds_read, nor hoist one address register per read out of a persistent kernel's tile loop)
    asm volatile("" : "+v"(a_idx));
    const unsigned short *a_base = a3 + a_idx;
    auto a_part = [&](int ks, int part) {
        if (!TONLY) ap[part] = *reinterpret_cast<const bf16x8 *>(a_base + part * A_PART + ks * 16);
        at[part] = *reinterpret_cast<const bf16x8 *>(a_base + part * A_PART + A_PT + ks * 16);
    };
    a_part(0, 2); a_part(0, 1); a_part(0, 0);
    __builtin_amdgcn_sched_group_barrier(0x100, TONLY ? 3 : 6, 0);    // (the pipeline below starts behind these reads)
#pragma unroll
    for (int s = 0; s < NSTEP; ++s) {
        const int ks = s / NP, j = s % NP, slot = (OFF + s) % 3;
        const int lc = MidGeom::prod_lc(WG, IP, j);
        const bool reload = j == NP - 1 && ks + 1 < KS;          // last product of the k-step: refill A behind its last uses
        if (s + 2 < NSTEP) b_load(WG, IP, s + 2, ring[(OFF + s + 2) % 3]);
        else if (POS < 3) b_load(WG, MidGeom::pix(POS + 1), s + 2 - NSTEP, ring[(OFF + s + 2) % 3]);   // the next pixel's first two steps
        const bf16x8 (&b)[3] = ring[slot];
        if (!TONLY) accp[lc] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ap[2], b[0], accp[lc], 0, 0, 0);
        acct[lc] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(at[2], b[0], acct[lc], 0, 0, 0);
        if (reload) a_part(ks + 1, 2);
        if (!TONLY) accp[lc] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ap[1], b[1], accp[lc], 0, 0, 0);
        acct[lc] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(at[1], b[1], acct[lc], 0, 0, 0);
        if (!TONLY) accp[lc] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ap[1], b[0], accp[lc], 0, 0, 0);
        acct[lc] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(at[1], b[0], acct[lc], 0, 0, 0);
        if (reload) a_part(ks + 1, 1);
        if (!TONLY) accp[lc] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ap[0], b[2], accp[lc], 0, 0, 0);
        acct[lc] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(at[0], b[2], acct[lc], 0, 0, 0);
        if (!TONLY) accp[lc] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ap[0], b[1], accp[lc], 0, 0, 0);
        acct[lc] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(at[0], b[1], acct[lc], 0, 0, 0);
        if (!TONLY) accp[lc] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ap[0], b[0], accp[lc], 0, 0, 0);
        acct[lc] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(at[0], b[0], acct[lc], 0, 0, 0);
        if (reload) a_part(ks + 1, 0);
        // pin the pipeline: left alone, the scheduler sinks every load to a few MFMAs before its use (register pressure
        // heuristics) and the prefetch distance is gone.  Per step: the weight loads (for step s + 2) first, then the MFMAs with
        // the LDS reads of the next k-step behind the pairs that free their registers.
        constexpr int PAIR_M = TONLY ? 1 : 2, DS = TONLY ? 1 : 2;
        if (s + 2 < NSTEP || POS < 3) __builtin_amdgcn_sched_group_barrier(0x020, 3, 0);
        if (reload) {
            __builtin_amdgcn_sched_group_barrier(0x008, PAIR_M, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, DS, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 2 * PAIR_M, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, DS, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 3 * PAIR_M, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, DS, 0);
        } else {
            __builtin_amdgcn_sched_group_barrier(0x008, 6 * PAIR_M, 0);
        }
        // every step is a scheduling region of its own: the order ACROSS steps is program order (what the ring needs), and the
        // group solver sees 20 instructions instead of a pixel's 2 000 (compile time: minutes -> seconds)
        __builtin_amdgcn_sched_barrier(0);
    }
}

#ifdef GEO_MID_PROF
// per wave group (waves 0 and 4 report): [0] prologue + first staging, [1..4] the four pixel intervals up to their barrier,
// [5] time parked in those barriers, [6] epilogue, [7] tiles
__device__ unsigned long long g_mid_prof[2][8];
#define GEO_MP_STAMP(v) const unsigned long long v = __builtin_amdgcn_s_memtime()
#define GEO_MP_ADD(slot, val) do { if (lane == 0 && wq == 0) atomicAdd(&g_mid_prof[wg][slot], (unsigned long long)(val)); } while (0)
#else
#define GEO_MP_STAMP(v)
#define GEO_MP_ADD(slot, val)
#endif

// ---- all eight 128-column chunks of a tile in ONE workgroup (dec_channels[2] = 64) ----------------
// Staging an input pixel's A block (norm1 + ReLU + 3-way split of 32 x C1 primal and tangent values) costs
// about as much VALU time as the 96 bf16 MFMAs one chunk spends on it.  Here the block is staged once (by
// 512 threads) and used by the five chunks that need it.  Waves 0-3 own chunks {0,3,4,7}, waves 4-7 chunks
// {1,2,5,6} (10 of the 20 (pixel, chunk) products each, 2 or 3 per pixel): 4 x 2 accumulator tiles = 128
// registers per wave, so two waves share a SIMD and one's weight-fragment loads (L2) hide behind the other's
// MFMAs.
// TONLY: tangent stream only -- decoders with fixed statistics (no norm, BatchNorm in eval mode) get the primal ConvT2 output
// once per LATENT from a separate launch over the nodes (run_jvp, "per-node primal"); here the primal pre-activation only
// decides the ReLU mask of the tangent.
template <int C1, bool GN, bool TONLY = false>
__global__ __launch_bounds__(512, 2) void mid_all_kernel(const float *__restrict__ pre1, const float *__restrict__ tpre1,
                                                        const NormConst *__restrict__ consts1, int consts_per_group,
                                                        int tiles_per_group, int c2,
                                                        const unsigned short *__restrict__ B3,
                                                        const float *__restrict__ b2, float *__restrict__ pre2,
                                                        float *__restrict__ tpre2, double *__restrict__ partial2,
                                                        int want_stats, const int32_t *__restrict__ slot_valid,
                                                        const float4 *__restrict__ gs1, int64_t e_base, int64_t n_edges,
                                                        int batch) {
    // BatchNorm / no norm at 128 channels: a lane stages a channel PAIR of 4 samples (constants in registers, the packed
    // pairs leave as conflict-free 4-byte stores); GroupNorm and the narrower decoders keep (sample, C1/16 channels)
    constexpr bool PAIR = C1 == 128 && !GN;
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    constexpr int LDK = C1 + 8;
    constexpr int KS = C1 / 16;
    constexpr int NL = 4;                                      // chunks per wave group
    constexpr int CPT = C1 / 16;                               // channels staged per thread
    __shared__ __attribute__((aligned(16))) unsigned short A3[2][3][2][TS][LDK];   // double buffered over input pixels
    __shared__ NormConst kc[C1];
    GEO_MP_STAMP(mp0);
    const int tile = blockIdx.x;
    const int group = tile / tiles_per_group;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wq = wave & 3, wg = wave >> 2;                   // column quarter, chunk group
    const int n1 = 4 * C1, n2 = 16 * c2;
    const size_t slot0 = (size_t)tile * TS;
    for (int c = threadIdx.x; c < C1; c += 512) kc[c] = consts1[(size_t)(consts_per_group ? group : 0) * C1 + c];

    f32x16 accp[NL], acct[NL];
#pragma unroll
    for (int lc = 0; lc < NL; ++lc)
#pragma unroll
        for (int i = 0; i < 16; ++i) { accp[lc][i] = 0.f; acct[lc][i] = 0.f; }

    const int r = lane & 31, h = lane >> 5;
    // staging: thread -> (sample = tid/16, CPT consecutive channels).  The raw pre-activations of the NEXT input
    // pixel are fetched (HBM) while the MFMAs of the current one run; one barrier per pixel.
    const int ss = threadIdx.x >> 4, k0 = (threadIdx.x & 15) * CPT;
    float rawp[CPT], rawt[CPT];
    const int k0p = 2 * lane, s0p = 4 * wave;
    f32x2 rp[4], rt[4];
    NormConst kA = {0.f, 0.f, 0.f, 0.f, 0.f}, kB = kA;
    if (PAIR) {
        const NormConst *kp = consts1 + (size_t)(consts_per_group ? group : 0) * C1 + k0p;
        kA = kp[0];
        kB = kp[1];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            rp[i] = *reinterpret_cast<const f32x2 *>(pre1 + (slot0 + s0p + i) * n1 + (size_t)MidGeom::pix(0) * C1 + k0p);
            rt[i] = *reinterpret_cast<const f32x2 *>(tpre1 + (slot0 + s0p + i) * n1 + (size_t)MidGeom::pix(0) * C1 + k0p);
        }
    } else {
        const float *xp = pre1 + (slot0 + ss) * n1 + (size_t)MidGeom::pix(0) * C1 + k0, *xt = tpre1 + (slot0 + ss) * n1 + (size_t)MidGeom::pix(0) * C1 + k0;
#pragma unroll
        for (int k = 0; k < CPT; ++k) { rawp[k] = xp[k]; rawt[k] = xt[k]; }
    }
    const int colw = wq * 32 + (lane & 31);
    const int lo = colw / c2, co = colw % c2;
    const float bias = b2[co];
    // GroupNorm (32 groups): the thread's CPT = C1/16 channels span exactly two groups of C1/32 channels
    float4 gg0 = make_float4(0.f, 0.f, 0.f, 0.f), gg1 = gg0;
    if (GN) {
        gg0 = gs1[(slot0 + ss) * 32 + (threadIdx.x & 15) * 2];
        gg1 = gs1[(slot0 + ss) * 32 + (threadIdx.x & 15) * 2 + 1];
    }
    __syncthreads();                                           // kc visible
    // stage_px(buf, nx): normalise + ReLU + split the raw values held in registers into A3[buf], then fetch those of the pixel at
    // position nx of the sequence (nx < 4)
    auto stage_px = [&](int buf, int nx) {
        if (PAIR) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                float a0, t0, a1, t1;
                norm_relu(kA, rp[i].x, rt[i].x, &a0, &t0);
                norm_relu(kB, rp[i].y, rt[i].y, &a1, &t1);
                unsigned wa[3], wt[3];
                if (!TONLY) split3_pair(a0, a1, wa[0], wa[1], wa[2]);
                split3_pair(t0, t1, wt[0], wt[1], wt[2]);
#pragma unroll
                for (int part = 0; part < 3; ++part) {
                    if (!TONLY) *reinterpret_cast<unsigned *>(&A3[buf][part][0][s0p + i][k0p]) = wa[part];
                    *reinterpret_cast<unsigned *>(&A3[buf][part][1][s0p + i][k0p]) = wt[part];
                }
            }
            if (nx < 4) {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    rp[i] = *reinterpret_cast<const f32x2 *>(pre1 + (slot0 + s0p + i) * n1 + (size_t)MidGeom::pix(nx) * C1 + k0p);
                    rt[i] = *reinterpret_cast<const f32x2 *>(tpre1 + (slot0 + s0p + i) * n1 + (size_t)MidGeom::pix(nx) * C1 + k0p);
                }
                // (a scheduling group of their own: otherwise these eight loads fill the first load groups of mid_products'
                // pipeline and push every weight load two steps late)
                __builtin_amdgcn_sched_group_barrier(0x020, 8, 0);
            }
        } else {
            unsigned short pp[3][CPT], pt[3][CPT];
#pragma unroll
            for (int k = 0; k < CPT; ++k) {
                float a, ta;
                if (GN) norm_relu_gn(k < CPT / 2 ? gg0 : gg1, kc[k0 + k].sc, kc[k0 + k].beta, rawp[k], rawt[k], &a, &ta);
                else norm_relu(kc[k0 + k], rawp[k], rawt[k], &a, &ta);
                if (!TONLY) split3(a, pp[0][k], pp[1][k], pp[2][k]);
                split3(ta, pt[0][k], pt[1][k], pt[2][k]);
            }
#pragma unroll
            for (int part = 0; part < 3; ++part)
#pragma unroll
                for (int k = 0; k < CPT; ++k) {
                    if (!TONLY) A3[buf][part][0][ss][k0 + k] = pp[part][k];
                    A3[buf][part][1][ss][k0 + k] = pt[part][k];
                }
            if (nx < 4) {
                const float *xp = pre1 + (slot0 + ss) * n1 + (size_t)MidGeom::pix(nx) * C1 + k0;
                const float *xt = tpre1 + (slot0 + ss) * n1 + (size_t)MidGeom::pix(nx) * C1 + k0;
#pragma unroll
                for (int k = 0; k < CPT; ++k) { rawp[k] = xp[k]; rawt[k] = xt[k]; }
                __builtin_amdgcn_sched_group_barrier(0x020, 2 * ((CPT + 3) / 4), 0);
            }
        }
    };
    // The two waves of a SIMD (w and w + 4 = the two chunk groups) take the phases of an input pixel in opposite order: waves 0-3
    // multiply pixel ip and then stage pixel ip + 1 into the other buffer, waves 4-7 stage first and multiply afterwards -- one
    // partner's vector / LDS-store work runs beside the other's MFMAs instead of both reaching the matrix pipe together.
    bf16x8 ring[3][3];
    const unsigned b_lane = (unsigned)(h * 3 * NC * 8 + (wq * 32 + r) * 8) * 2u;      // bytes
    const __amdgpu_buffer_rsrc_t b_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<unsigned short *>(B3), 0, (int)((size_t)8 * MAX_BLOCKS * C1 * NC * 3 * 2), 0x00020000);
    {   // the first two steps' weight fragments of pixel 0 (both wave groups): in flight during the first staging
        const int cb0 = wg == 0 ? MidGeom::prod_cb(0, MidGeom::pix(0), 0) : MidGeom::prod_cb(1, MidGeom::pix(0), 0);
        const int cb1 = wg == 0 ? MidGeom::prod_cb(0, MidGeom::pix(0), 1) : MidGeom::prod_cb(1, MidGeom::pix(0), 1);
#pragma unroll
        for (int part = 0; part < 3; ++part) {
            ring[0][part] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(b_rsrc, b_lane, (cb0 * KS * 2 * 3 + part) * NC * 8 * 2, 0));
            ring[1][part] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(b_rsrc, b_lane, (cb1 * KS * 2 * 3 + part) * NC * 8 * 2, 0));
        }
    }
    stage_px(0, 1);
    if (wg == 1) __builtin_amdgcn_s_setprio(1);                // the later-dispatched half loses every issue arbitration otherwise
    lds_barrier();                                             // (LDS only: the next pixel's loads stay in flight)
    GEO_MP_STAMP(mp1);
    GEO_MP_ADD(0, mp1 - mp0);
    const unsigned short *a3 = &A3[0][0][0][0][0];
    const unsigned a_lane = (unsigned)(r * LDK + h * 8);
    constexpr unsigned A_BUF = 3u * 2u * TS * LDK;
    // (ring slots run through the four pixels of a tile: MidGeom::ring_off gives every pixel body its first slot)
#define GEO_MID_PIXEL(WGV, IPV)                                                                                        \
    mid_products<C1, TONLY, WGV, IPV>(a3, a_lane + ((IPV) & 1) * A_BUF, b_rsrc, b_lane, accp, acct, ring)
#ifdef GEO_MID_PROF
    unsigned long long mp_prev = mp1, mp_wait = 0;
#define GEO_MP_INTERVAL(slot, with_barrier)                                                                            \
    { const unsigned long long t_a = __builtin_amdgcn_s_memtime();                                                      \
      if (with_barrier) { lds_barrier(); }                                                                              \
      const unsigned long long t_b = __builtin_amdgcn_s_memtime();                                                      \
      GEO_MP_ADD(slot, t_a - mp_prev); mp_wait += t_b - t_a; mp_prev = t_b; }
#else
#define GEO_MP_INTERVAL(slot, with_barrier) { if (with_barrier) { lds_barrier(); } }
#endif
    if (wg == 0) {
        GEO_MID_PIXEL(0, 0); stage_px(1, 2); GEO_MP_INTERVAL(1, true)
        GEO_MID_PIXEL(0, 1); stage_px(0, 3); GEO_MP_INTERVAL(2, true)
        GEO_MID_PIXEL(0, 2); stage_px(1, 4); GEO_MP_INTERVAL(3, true)
        GEO_MID_PIXEL(0, 3); GEO_MP_INTERVAL(4, false)
    } else {
        stage_px(1, 2); GEO_MID_PIXEL(1, 0); GEO_MP_INTERVAL(1, true)
        stage_px(0, 3); GEO_MID_PIXEL(1, 1); GEO_MP_INTERVAL(2, true)
        stage_px(1, 4); GEO_MID_PIXEL(1, 2); GEO_MP_INTERVAL(3, true)
        GEO_MID_PIXEL(1, 3); GEO_MP_INTERVAL(4, false)
    }
#undef GEO_MP_INTERVAL
#undef GEO_MID_PIXEL

    // rows of the tile that hold edges of the chunk (slot_valid_kernel's rule, computed here: no loads in the epilogue)
    int64_t cnt_g = n_edges - (e_base + (int64_t)(group >> 1) * batch);
    if (cnt_g > batch) cnt_g = batch;
    const int n_valid = (int)cnt_g - (tile - group * tiles_per_group) * TS;
    // batch statistics: a lane's column is the same channel co in all four chunks of its wave, so their (and the two row
    // halves') sums are added in registers; the four waves holding channel co (column quarters wq and wq ^ 2 of both wave
    // groups) meet in LDS and are added in wave order: partial2 is [tile][c2][4], the tile's 16 pixels already summed
    __shared__ double red[8][32][4];
    double sx = 0, sxx = 0, st_ = 0, sxt = 0;
#pragma unroll
    for (int lc = 0; lc < NL; ++lc) {
        const int ch = wg == 0 ? MidGeom::chunk_of(0, lc) : MidGeom::chunk_of(1, lc);
        const int op = lo == 0 ? MidGeom::opix(ch, 0) : MidGeom::opix(ch, 1);
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const int row = (q & 3) + 8 * (q >> 2) + 4 * (lane >> 5);
            const float x = accp[lc][q] + bias, t = acct[lc][q];
            if (!TONLY) pre2[(slot0 + row) * n2 + (size_t)op * c2 + co] = x;
            tpre2[(slot0 + row) * n2 + (size_t)op * c2 + co] = t;
            if (!TONLY && row < n_valid) { sx += x; sxx += (double)x * x; st_ += t; sxt += (double)x * t; }
        }
    }
    if (want_stats) {
        sx += __shfl_xor(sx, 32, 64); sxx += __shfl_xor(sxx, 32, 64);
        st_ += __shfl_xor(st_, 32, 64); sxt += __shfl_xor(sxt, 32, 64);
        if (lane < 32) { red[wave][lane][0] = sx; red[wave][lane][1] = sxx; red[wave][lane][2] = st_; red[wave][lane][3] = sxt; }
        __syncthreads();
        if (threadIdx.x < 64) {                                // thread = channel
            const int w0 = threadIdx.x >> 5, l = threadIdx.x & 31;
            double a0 = 0, a1 = 0, a2 = 0, a3 = 0;
#pragma unroll
            for (int w = 0; w < 4; ++w) { a0 += red[w0 + 2 * w][l][0]; a1 += red[w0 + 2 * w][l][1]; a2 += red[w0 + 2 * w][l][2]; a3 += red[w0 + 2 * w][l][3]; }
            double *p = partial2 + ((size_t)tile * c2 + threadIdx.x) * 4;
            p[0] = a0; p[1] = a1; p[2] = a2; p[3] = a3;
        }
    }
#ifdef GEO_MID_PROF
    { const unsigned long long t_e = __builtin_amdgcn_s_memtime();
      GEO_MP_ADD(6, t_e - mp_prev); GEO_MP_ADD(5, mp_wait); GEO_MP_ADD(7, 1); }
#endif
}

// ---- ConvT2, persistent and skewed (dec_channels 128 -> 64, BatchNorm or no norm): the default for the shipped decoder ------
// In-kernel stamps of mid_all_kernel (one tile per workgroup, -DGEO_MID_PROF): the four pixel intervals keep the matrix pipe 85 %
// busy, but the tile's prologue (constants, first staging: 9 100 cycles) and epilogue (128 stores per wave + fp64 statistics:
// 13 800) run with NO MFMA beside them -- a quarter of the 95 600 cycles of a tile.  Here ONE workgroup per CU walks over its tiles
// and the two wave groups (waves 0-3 / 4-7 = the two chunk groups, one wave of each per SIMD) trade those phases:
//     interval 0 : group 0  P0, stage pixel 1 (all 32 samples)      group 1  EPILOGUE of the previous tile, P0
//     interval 1 : group 0  P1, stage half of pixel 2                 group 1  stage half of pixel 2, P1
//     interval 2 : group 0  P2, stage half of pixel 3                 group 1  stage half of pixel 3, P2
//     interval 3 : group 0  P3, EPILOGUE of this tile                 group 1  stage pixel 0 of the NEXT tile (all samples), P3
// (one LDS-only barrier after each interval; Pk = the products of the k-th input pixel of the sequence 3, 2, 1, 0, mid_products).  One group's stores / statistics /
// staging always run beside the other group's MFMAs.  The statistics leave per wave (partial2 [tile][4 wave slots][channel],
// finalize_batch_kernel adds them: npx_stored = 4), so no epilogue needs a workgroup barrier.  Same products in the same order as
// mid_all_kernel: pre-activations are bit-identical; the fp64 statistic sums associate differently (last-bit differences).
__global__ __launch_bounds__(512, 2) void mid_pipe_kernel(const float *__restrict__ pre1, const float *__restrict__ tpre1,
                                                         const NormConst *__restrict__ consts1, int consts_per_group,
                                                         int tiles_per_group, int n_tiles, int /*c2 == 64*/,
                                                         const unsigned short *__restrict__ B3, const float *__restrict__ b2,
                                                         float *__restrict__ pre2, float *__restrict__ tpre2,
                                                         double *__restrict__ partial2, int want_stats, int64_t e_base,
                                                         int64_t n_edges, int batch) {
    constexpr int C1 = 128, LDK = C1 + 8, KS = C1 / 16, NL = 4, c2 = 64;
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    __shared__ __attribute__((aligned(16))) unsigned short A3[2][3][2][TS][LDK];   // double buffered over input pixels
    // (readfirstlane: everything derived from the wave number is then provably wave-uniform -- scalar branches on the chunk group,
    // scalar offsets for the buffer loads / stores below instead of a 64-bit address pair per access)
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int wq = wave & 3, wg = wave >> 2;                   // column quarter, chunk group
    constexpr int n1 = 4 * C1, n2 = 16 * c2;
    const int r = lane & 31, h = lane >> 5;
    const int k0p = 2 * lane;                                  // the lane's channel pair when staging
    const int lo = wq >> 1, co = (wq & 1) * 32 + r;            // output pixel of the chunk (wave-uniform), channel
    const float bias = b2[co];
    const unsigned v_out = (unsigned)(4 * h * n2 + co) * 4u;    // the lane's byte offset inside a tile of pre2 / tpre2
    const int my_tiles = (n_tiles - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;
    if (my_tiles <= 0) return;

    f32x16 accp[NL], acct[NL];
    auto zero_acc = [&]() {
#pragma unroll
        for (int lc = 0; lc < NL; ++lc)
#pragma unroll
            for (int i = 0; i < 16; ++i) { accp[lc][i] = 0.f; acct[lc][i] = 0.f; }
    };
    zero_acc();
    // staging state of this wave: constants of the tile it stages next, and raw pre-activations in flight
    NormConst kA, kB;
    f32x2 rp[4], rt[4];
    auto load_consts = [&](int tile) {
        const NormConst *kp = consts1 + (size_t)(consts_per_group ? tile / tiles_per_group : 0) * C1 + k0p;
        kA = kp[0];
        kB = kp[1];
    };
    // raw values of NS samples (sample0 ...) of input pixel px of a tile -> rp / rt (in flight until stage())
    // (per-tile buffer descriptors: the tile's base is scalar arithmetic, the lane contributes ONE loop-invariant 32-bit offset)
    const unsigned v_pair = (unsigned)k0p * 4u;
    auto fetch = [&](int tile, int px, int sample0, auto ns_tag) {
        constexpr int NS = decltype(ns_tag)::value;
        const size_t tbase = (size_t)tile * TS * n1;
        const __amdgpu_buffer_rsrc_t rp_src = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(pre1 + tbase), 0, TS * n1 * 4, 0x00020000);
        const __amdgpu_buffer_rsrc_t rt_src = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(tpre1 + tbase), 0, TS * n1 * 4, 0x00020000);
        const int soff = (sample0 * n1 + MidGeom::pix(px) * C1) * 4;      // (px = position in the tile's pixel sequence)
#pragma unroll
        for (int i = 0; i < NS; ++i) {
            rp[i] = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(rp_src, v_pair, soff + i * n1 * 4, 0));
            rt[i] = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(rt_src, v_pair, soff + i * n1 * 4, 0));
        }
        __builtin_amdgcn_sched_group_barrier(0x020, 2 * NS, 0);  // (a group of their own: see mid_all_kernel's stage_px)
    };
    // norm1 + ReLU + 3-way split of the fetched values into A3[buf]
    auto stage = [&](int buf, int sample0, auto ns_tag) {
        constexpr int NS = decltype(ns_tag)::value;
        // one address register per call (empty asm: not folded, not hoisted out of the tile loop), everything else immediates
        unsigned w_idx = (unsigned)((buf * 3 * 2 * TS + sample0) * LDK + k0p);
        asm volatile("" : "+v"(w_idx));
        unsigned short *w = &A3[0][0][0][0][0] + w_idx;
#pragma unroll
        for (int i = 0; i < NS; ++i) {
            float a0, t0, a1, t1;
            norm_relu(kA, rp[i].x, rt[i].x, &a0, &t0);
            norm_relu(kB, rp[i].y, rt[i].y, &a1, &t1);
            unsigned wa[3], wt[3];
            split3_pair(a0, a1, wa[0], wa[1], wa[2]);
            split3_pair(t0, t1, wt[0], wt[1], wt[2]);
#pragma unroll
            for (int part = 0; part < 3; ++part) {
                *reinterpret_cast<unsigned *>(w + (part * 2 + 0) * TS * LDK + i * LDK) = wa[part];
                *reinterpret_cast<unsigned *>(w + (part * 2 + 1) * TS * LDK + i * LDK) = wt[part];
            }
        }
    };
    std::integral_constant<int, 4> HALF;
    const int s_half = 4 * wave, s_full = 8 * wq;              // first sample a wave stages in the two modes

    // weight fragments: buffer descriptor + ring (mid_products)
    bf16x8 ring[3][3];
    const unsigned b_lane = (unsigned)(h * 3 * NC * 8 + (wq * 32 + r) * 8) * 2u;      // bytes
    const __amdgpu_buffer_rsrc_t b_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<unsigned short *>(B3), 0, (int)((size_t)8 * MAX_BLOCKS * C1 * NC * 3 * 2), 0x00020000);
    auto ring_init = [&]() {                                   // the first two steps of pixel 0 (ring slots 0, 1)
        const int cb0 = wg == 0 ? MidGeom::prod_cb(0, MidGeom::pix(0), 0) : MidGeom::prod_cb(1, MidGeom::pix(0), 0);
        const int cb1 = wg == 0 ? MidGeom::prod_cb(0, MidGeom::pix(0), 1) : MidGeom::prod_cb(1, MidGeom::pix(0), 1);
#pragma unroll
        for (int part = 0; part < 3; ++part) {
            ring[0][part] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(b_rsrc, b_lane, (cb0 * KS * 2 * 3 + part) * NC * 8 * 2, 0));
            ring[1][part] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(b_rsrc, b_lane, (cb1 * KS * 2 * 3 + part) * NC * 8 * 2, 0));
        }
    };
    // stores + statistics of one tile's accumulators (this wave's four chunks), then the accumulators start over
    auto epilogue = [&](int tile) {
        const int group = tile / tiles_per_group;
        int64_t cnt_g = n_edges - (e_base + (int64_t)(group >> 1) * batch);
        if (cnt_g > batch) cnt_g = batch;
        const int n_valid = (int)cnt_g - (tile - group * tiles_per_group) * TS;
        const size_t tbase = (size_t)tile * TS * n2;
        const __amdgpu_buffer_rsrc_t p_dst = __builtin_amdgcn_make_buffer_rsrc(pre2 + tbase, 0, TS * n2 * 4, 0x00020000);
        const __amdgpu_buffer_rsrc_t t_dst = __builtin_amdgcn_make_buffer_rsrc(tpre2 + tbase, 0, TS * n2 * 4, 0x00020000);
        double sx = 0, sxx = 0, st_ = 0, sxt = 0;
#pragma unroll
        for (int lc = 0; lc < NL; ++lc) {
            const int ch = wg == 0 ? MidGeom::chunk_of(0, lc) : MidGeom::chunk_of(1, lc);
            const int op = lo == 0 ? MidGeom::opix(ch, 0) : MidGeom::opix(ch, 1);
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const int row = (q & 3) + 8 * (q >> 2) + 4 * h;
                const float x = accp[lc][q] + bias, t = acct[lc][q];
                const int soff = (((q & 3) + 8 * (q >> 2)) * n2 + op * c2) * 4;          // scalar; the lane adds v_out
                __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(x), p_dst, v_out, soff, 0);
                __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(t), t_dst, v_out, soff, 0);
                // (no branch per element: rows beyond the chunk's edges contribute zeros; all but a batch's last tile are full)
                const double xd = row < n_valid ? (double)x : 0.0, td = row < n_valid ? (double)t : 0.0;
                sx += xd; sxx = fma(xd, xd, sxx); st_ += td; sxt = fma(xd, td, sxt);
            }
        }
        if (want_stats) {
            sx += __shfl_xor(sx, 32, 64); sxx += __shfl_xor(sxx, 32, 64);
            st_ += __shfl_xor(st_, 32, 64); sxt += __shfl_xor(sxt, 32, 64);
            if (lane < 32) {                                   // wave slot = (chunk group, column half): channels (wq & 1) * 32 + lane
                double *pp = partial2 + (((size_t)tile * 4 + (wg * 2 + (wq >> 1))) * c2 + co) * 4;
                pp[0] = sx; pp[1] = sxx; pp[2] = st_; pp[3] = sxt;
            }
        }
        zero_acc();
    };

    const unsigned short *a3 = &A3[0][0][0][0][0];
    const unsigned a_lane = (unsigned)(r * LDK + h * 8);
    constexpr unsigned A_BUF = 3u * 2u * TS * LDK;
#define GEO_PIPE_PIXEL(WGV, IPV)                                                                                       \
    mid_products<C1, false, WGV, IPV>(a3, a_lane + ((IPV) & 1) * A_BUF, b_rsrc, b_lane, accp, acct, ring)

    // "all samples" by one group = two passes of four samples per wave through the same registers: the first four were fetched
    // an interval ago, the second four are fetched behind the first staging and their latency is exposed -- in a group that has
    // 10 000 cycles of slack in that interval (the other group is in its epilogue)
#define GEO_PIPE_SEP() __builtin_amdgcn_sched_barrier(0)
#define GEO_STAGE_FULL(BUF, TILE, PX)                                                                                   \
    { stage(BUF, s_full, HALF); fetch(TILE, PX, s_full + 4, HALF); GEO_PIPE_SEP(); stage(BUF, s_full + 4, HALF); }

    // ---- prologue: pixel 0 of the first tile, staged by halves; group 0 already fetches the first half of its pixel 1
    int tile = blockIdx.x;
    load_consts(tile);
    fetch(tile, 0, s_half, HALF);
    ring_init();
    stage(0, s_half, HALF);
    if (wg == 0) fetch(tile, 1, s_full, HALF);
    else fetch(tile, 2, s_half, HALF);
    // Matrix-pipe priority to the group whose vector work (staging, epilogue) FOLLOWS its products in an interval -- group 0 in all
    // four: its MFMAs go first and its tail runs beside group 1's MFMAs.  (Measured the other way round, -DGEO_MID_PROF: group 1's
    // products first, then group 0's, then group 0's staging alone: 18 400 cycles for an interval whose pipe work is 15 300.)
    if (wg == 0) __builtin_amdgcn_s_setprio(1);
    lds_barrier();
    // (one tile loop per wave group: both run the same number of barriers per tile; separate loops keep the register allocator
    // from reconciling the two groups' live ranges at every iteration)
#ifdef GEO_MID_PROF
    unsigned long long pb_prev = __builtin_amdgcn_s_memtime(), pb_wait = 0;
#define GEO_PB(slot)                                                                                                   \
    { const unsigned long long t_a = __builtin_amdgcn_s_memtime();                                                      \
      lds_barrier();                                                                                                    \
      const unsigned long long t_b = __builtin_amdgcn_s_memtime();                                                      \
      if (lane == 0 && wq == 0) { atomicAdd(&g_mid_prof[wg][1 + slot], t_a - pb_prev); atomicAdd(&g_mid_prof[wg][5], t_b - t_a);   \
                                  if (slot == 3) atomicAdd(&g_mid_prof[wg][7], 1ull); }                                 \
      pb_prev = t_b; }
#else
#define GEO_PB(slot) lds_barrier();
#endif
    if (wg == 0) {
        for (int it = 0; it < my_tiles; ++it) {
            const int next = tile + (int)gridDim.x;
            const bool has_next = it + 1 < my_tiles;
            GEO_PIPE_PIXEL(0, 0); GEO_PIPE_SEP();
            GEO_STAGE_FULL(1, tile, 1); fetch(tile, 2, s_half, HALF); GEO_PB(0);
            GEO_PIPE_PIXEL(0, 1); GEO_PIPE_SEP(); stage(0, s_half, HALF); fetch(tile, 3, s_half, HALF); GEO_PB(1);
            GEO_PIPE_PIXEL(0, 2); GEO_PIPE_SEP(); stage(1, s_half, HALF); GEO_PB(2);
            GEO_PIPE_PIXEL(0, 3); GEO_PIPE_SEP();
            epilogue(tile);
            GEO_PIPE_SEP();
            if (has_next) { load_consts(next); fetch(next, 1, s_full, HALF); ring_init(); }
            GEO_PB(3);
            tile = next;
        }
    } else {
        // (raising group 1's priority for its vector phases -- staging, epilogue -- was measured: 18.0 ms against 17.7 ms, they slow
        // group 0's product stream more than they gain)
        for (int it = 0; it < my_tiles; ++it) {
            const int next = tile + (int)gridDim.x;
            const bool has_next = it + 1 < my_tiles;
            if (it > 0) { epilogue(tile - (int)gridDim.x); GEO_PIPE_SEP(); ring_init(); }
            GEO_PIPE_PIXEL(1, 0); GEO_PB(0);
            stage(0, s_half, HALF); fetch(tile, 3, s_half, HALF); GEO_PIPE_SEP();
            GEO_PIPE_PIXEL(1, 1); GEO_PB(1);
            stage(1, s_half, HALF);
            if (has_next) { load_consts(next); fetch(next, 0, s_full, HALF); }
            GEO_PIPE_SEP();
            GEO_PIPE_PIXEL(1, 2); GEO_PB(2);
            if (has_next) { GEO_STAGE_FULL(0, next, 0); fetch(next, 2, s_half, HALF); }
            GEO_PIPE_SEP();
            GEO_PIPE_PIXEL(1, 3); GEO_PB(3);
            tile = next;
        }
        epilogue(tile - (int)gridDim.x);                         // the last tile of group 1
    }
#undef GEO_STAGE_FULL
#undef GEO_PIPE_SEP
#undef GEO_PB
#undef GEO_PIPE_PIXEL
}

__global__ __launch_bounds__(256) void slot_valid_kernel(int64_t e_base, int64_t n_edges, int batch, int tiles_per_group,
                                                        int64_t n_slots, int32_t *__restrict__ slot_valid) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_slots; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t per_group = (int64_t)tiles_per_group * TS;
        const int64_t g = i / per_group, within = i % per_group;
        const int64_t e = e_base + (g >> 1) * batch + within;
        slot_valid[i] = (within < batch && e < n_edges) ? 1 : 0;
    }
}

// per-node primal: the latent of every slot (start side: src, end side: dst; padding slots: latent 0)
__global__ __launch_bounds__(256) void slot_node_kernel(const int32_t *__restrict__ src, const int32_t *__restrict__ dst,
                                                       int64_t e_base, int64_t n_edges, int batch, int tiles_per_group,
                                                       int64_t n_slots, int32_t *__restrict__ slot_node) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_slots; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t per_group = (int64_t)tiles_per_group * TS;
        const int64_t g = i / per_group, within = i % per_group;
        const int64_t e = e_base + (g >> 1) * batch + within;
        slot_node[i] = (within < batch && e < n_edges) ? ((g & 1) ? dst[e] : src[e]) : 0;
    }
}

// ---------------------------------------------------------------------------------- back
// One workgroup = BACK_TS sample slots.  LDS (dynamic): a2 / ta2 [BACK_TS][16 px][c2] after norm2 + ReLU,
// the packed ConvT3 taps W3p [16][co][c2], and the squared tangent outputs.  A work item is
// (sample, output element, channel quarter): the 4 quarters of a dot product sit in adjacent lanes and are
// summed with two shuffles, so all 256 threads are busy even for the 16-pixel FashionMNIST head.
__global__ __launch_bounds__(256) void back_kernel(const float *__restrict__ pre2, const float *__restrict__ tpre2,
                                                  const NormConst *__restrict__ consts2, int consts_per_group,
                                                  int slots_per_group, int c2, int co_n, int s_out, int pad3,
                                                  const float *__restrict__ W3p, const float *__restrict__ b3,
                                                  float *__restrict__ norms) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int n2 = 16 * c2;
    const int p_out = co_n * s_out * s_out;
    const int ldc = c2 + 4;                             // padded pixel row: neighbouring pixels / taps start on
    const int lds_n2 = 16 * ldc;                        // different LDS banks (c2 is a multiple of the bank cycle)
    float *a2 = smem;                                   // [BACK_TS][16][ldc]
    float *ta2 = a2 + (size_t)BACK_TS * lds_n2;         // [BACK_TS][16][ldc]
    float *w3 = ta2 + (size_t)BACK_TS * lds_n2;         // [16 taps * co_n][ldc]
    float *jt2 = w3 + (size_t)16 * co_n * ldc;          // [BACK_TS][p_out]
    const size_t slot0 = (size_t)blockIdx.x * BACK_TS;
    const int group = (int)(slot0 / slots_per_group);
    const NormConst *kc = consts2 + (size_t)(consts_per_group ? group : 0) * c2;
    {
        const float4 *xp4 = reinterpret_cast<const float4 *>(pre2 + slot0 * n2);
        const float4 *xt4 = reinterpret_cast<const float4 *>(tpre2 + slot0 * n2);
        const int c4n = c2 / 4;
        for (int i = threadIdx.x; i < BACK_TS * n2 / 4; i += 256) {
            const float4 x = xp4[i], t = xt4[i];
            const int c = (i % c4n) * 4, row = i / c4n;              // row = s*16 + px
            float4 a, ta;
            norm_relu(kc[c], x.x, t.x, &a.x, &ta.x);
            norm_relu(kc[c + 1], x.y, t.y, &a.y, &ta.y);
            norm_relu(kc[c + 2], x.z, t.z, &a.z, &ta.z);
            norm_relu(kc[c + 3], x.w, t.w, &a.w, &ta.w);
            *reinterpret_cast<float4 *>(a2 + (size_t)row * ldc + c) = a;
            *reinterpret_cast<float4 *>(ta2 + (size_t)row * ldc + c) = ta;
        }
        for (int i = threadIdx.x; i < 16 * co_n * c2; i += 256) w3[(size_t)(i / c2) * ldc + (i % c2)] = W3p[i];
    }
    __syncthreads();
    const int qlen = c2 / 4;                             // c2 is a multiple of 16 (checked on the host)
    for (int item = threadIdx.x; item < BACK_TS * p_out * 4; item += 256) {
        const int q = item & 3, so = item >> 2;
        const int s = so / p_out, o = so % p_out;
        const int co = o / (s_out * s_out), oy = (o / s_out) % s_out, ox = o % s_out;
        float x = 0.f, t = 0.f;
        for (int iy = 0; iy < 4; ++iy) {
            const int ky = oy + pad3 - 2 * iy;
            if (ky < 0 || ky > 3) continue;
            for (int ix = 0; ix < 4; ++ix) {
                const int kx = ox + pad3 - 2 * ix;
                if (kx < 0 || kx > 3) continue;
                const float4 *w = reinterpret_cast<const float4 *>(w3 + ((size_t)(ky * 4 + kx) * co_n + co) * ldc + q * qlen);
                const float4 *pa = reinterpret_cast<const float4 *>(a2 + ((size_t)s * 16 + iy * 4 + ix) * ldc + q * qlen);
                const float4 *pt = reinterpret_cast<const float4 *>(ta2 + ((size_t)s * 16 + iy * 4 + ix) * ldc + q * qlen);
                for (int c4 = 0; c4 < qlen / 4; ++c4) {
                    const float4 wv = w[c4], av = pa[c4], tv = pt[c4];
                    x = fmaf(av.x, wv.x, x); x = fmaf(av.y, wv.y, x); x = fmaf(av.z, wv.z, x); x = fmaf(av.w, wv.w, x);
                    t = fmaf(tv.x, wv.x, t); t = fmaf(tv.y, wv.y, t); t = fmaf(tv.z, wv.z, t); t = fmaf(tv.w, wv.w, t);
                }
            }
        }
        x += __shfl_xor(x, 1, 64); t += __shfl_xor(t, 1, 64);
        x += __shfl_xor(x, 2, 64); t += __shfl_xor(t, 2, 64);
        if (q == 0) {
            x += b3[co];
            const float sg = 1.0f / (1.0f + expf(-x));
            const float j = t * sg * (1.0f - sg);
            jt2[so] = j * j;
        }
    }
    __syncthreads();
    // fixed-order fp64 sum per sample by one wave
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int s = wave; s < BACK_TS; s += 4) {
        double acc = 0.0;
        for (int o = lane; o < p_out; o += 64) acc += (double)jt2[s * p_out + o];
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) acc += __shfl_xor(acc, off, 64);
        if (lane == 0) norms[slot0 + s] = (float)sqrt(acc);
    }
}

// ---------------------------------------------------------------------------------- back on the matrix cores
// ConvT3 as a [32 samples x 1024] x [1024 x P] product (P = 16 or 192 outputs, padded to 32-column tiles; taps that
// fall outside the 4x4 kernel are zeros in the packed weights), same exact bf16 x 3 split as the mid kernel.
// The 1024-deep reduction is cut into 8 blocks of 128 (two input pixels); inside a block the four waves take two
// 16-deep steps each, and their partial tiles are summed through LDS before sigmoid' and the squared norm.
// W3b[kb 8][part 3][ks 8][h 2][NP][8] bf16
__global__ __launch_bounds__(256) void pack_back_bf16_kernel(const float *__restrict__ w3, int c2, int co_n, int s_out,
                                                            int pad3, int np, unsigned short *__restrict__ W3b) {
    const int K = 16 * c2, P = co_n * s_out * s_out;
    const size_t total = (size_t)K * np;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int o = (int)(i % np), k = (int)(i / np);
        float v = 0.f;
        if (o < P) {
            const int ip = k / c2, ci = k % c2, iy = ip >> 2, ix = ip & 3;
            const int co = o / (s_out * s_out), oy = (o / s_out) % s_out, ox = o % s_out;
            const int ky = oy + pad3 - 2 * iy, kx = ox + pad3 - 2 * ix;
            if (ky >= 0 && ky < 4 && kx >= 0 && kx < 4) v = w3[(((size_t)ci * co_n + co) * 4 + ky) * 4 + kx];
        }
        unsigned short q[3];
        split3(v, q[0], q[1], q[2]);
        const int kb = k >> 7, kk = k & 127, ks = kk >> 4, h = (kk >> 3) & 1, j = kk & 7;
        for (int part = 0; part < 3; ++part)
            W3b[((((size_t)kb * 3 + part) * 8 + ks) * 2 + h) * (size_t)np * 8 + (size_t)o * 8 + j] = q[part];
    }
}

// MODE 0: primal + tangent of every slot -> |J dz| per slot.
// Per-node primal (decoders with fixed statistics; run_jvp): MODE 2 runs over the LATENTS, primal only, and stores the
// sigmoid of every output (sg_node [node][NP]); MODE 1 runs over the edge slots, tangent only: the ReLU mask comes from the
// slot's node row of pre2 (gathered), sigmoid' from sg_node -- the same products in the same order as MODE 0.
template <int NT, bool GN, int MODE = 0>
__global__ __launch_bounds__(256, 2) void back_mfma_kernel(const float *__restrict__ pre2, const float *__restrict__ tpre2,
                                                       const NormConst *__restrict__ consts2, int consts_per_group,
                                                       int tiles_per_group, int co_n, int s_out,
                                                       const unsigned short *__restrict__ W3b,
                                                       const float *__restrict__ b3, float *__restrict__ norms,
                                                       const float4 *__restrict__ gs2, float *__restrict__ sg_node = nullptr,
                                                       const int32_t *__restrict__ src = nullptr,
                                                       const int32_t *__restrict__ dst = nullptr, int64_t e_base = 0,
                                                       int64_t n_edges = 0, int batch = 1, float *__restrict__ jac_node = nullptr,
                                                       int unit_d = 0) {
    constexpr int C2 = 64, KB = 128, LDK = KB + 8, NP = NT * 32;
    __shared__ __attribute__((aligned(16))) unsigned short A3[3][2][TS][LDK];     // 52 KB, reused for the reduction
    __shared__ NormConst kc[C2];
    __shared__ float4 gsl[GN ? TS : 1][32];                                       // GroupNorm: per (sample, group)
    const int tile = blockIdx.x;
    const int group = tile / tiles_per_group;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int n2 = 16 * C2, P = co_n * s_out * s_out;
    const size_t slot0 = (size_t)tile * TS;
    for (int c = threadIdx.x; c < C2; c += 256) kc[c] = consts2[(size_t)(consts_per_group ? group : 0) * C2 + c];
    if (GN)
        for (int i = threadIdx.x; i < TS * 32; i += 256) gsl[i >> 5][i & 31] = gs2[slot0 * 32 + i];

    f32x16 accp[NT], acct[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int i = 0; i < 16; ++i) { accp[nt][i] = 0.f; acct[nt][i] = 0.f; }

    const int r = lane & 31, h = lane >> 5;
    const int ss = threadIdx.x >> 3, k0 = (threadIdx.x & 7) * 16;
    // MODE 1: the latent a slot belongs to (start side: src, end side: dst; padding slots read latent 0, their result is unused)
    auto edge_of = [&](int sample) -> int64_t {                   // -1: padding slot
        const int tg = tile - group * tiles_per_group;
        const int64_t e = e_base + (int64_t)(group >> 1) * batch + (int64_t)tg * TS + sample;
        return (tg * TS + sample >= batch || e >= n_edges) ? -1 : e;
    };
    auto node_of = [&](int sample) -> size_t {
        const int64_t e = edge_of(sample);
        return e < 0 ? 0 : (size_t)((group & 1) ? dst[e] : src[e]);
    };
    const size_t prow = MODE == 1 ? node_of(ss) : slot0 + ss;      // row of the primal pre-activations this thread stages
    for (int kb = 0; kb < 8; ++kb) {
        __syncthreads();
        {
            const float *xp = pre2 + prow * n2 + (size_t)kb * KB + k0;
            const float *xt = (MODE == 2 ? pre2 : tpre2) + (slot0 + ss) * n2 + (size_t)kb * KB + k0;
#pragma unroll
            for (int k8 = 0; k8 < 16; k8 += 8) {
                u16x8 pp[3], pt[3];
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    float a, ta;
                    const int c = (k0 + k8 + k) & (C2 - 1);
                    if (GN) norm_relu_gn(gsl[GN ? ss : 0][c >> 1], kc[c].sc, kc[c].beta, xp[k8 + k], xt[k8 + k], &a, &ta);
                    else norm_relu(kc[c], xp[k8 + k], xt[k8 + k], &a, &ta);
                    unsigned short q1, q2, q3;
                    if (MODE != 1) { split3(a, q1, q2, q3); pp[0][k] = q1; pp[1][k] = q2; pp[2][k] = q3; }
                    if (MODE != 2) { split3(ta, q1, q2, q3); pt[0][k] = q1; pt[1][k] = q2; pt[2][k] = q3; }
                }
#pragma unroll
                for (int part = 0; part < 3; ++part) {
                    if (MODE != 1) *reinterpret_cast<u16x8 *>(&A3[part][0][ss][k0 + k8]) = pp[part];
                    if (MODE != 2) *reinterpret_cast<u16x8 *>(&A3[part][1][ss][k0 + k8]) = pt[part];
                }
            }
        }
        __syncthreads();
#pragma unroll
        for (int kq = 0; kq < 2; ++kq) {
            const int ks = wave * 2 + kq;                     // this wave's 16-deep steps of the block
            bf16x8 ap[3], at[3];
#pragma unroll
            for (int part = 0; part < 3; ++part) {
                if (MODE != 1) ap[part] = *reinterpret_cast<const bf16x8 *>(&A3[part][0][r][ks * 16 + h * 8]);
                if (MODE != 2) at[part] = *reinterpret_cast<const bf16x8 *>(&A3[part][1][r][ks * 16 + h * 8]);
            }
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                bf16x8 b[3];
#pragma unroll
                for (int part = 0; part < 3; ++part)
                    b[part] = *reinterpret_cast<const bf16x8 *>(
                        W3b + ((((size_t)kb * 3 + part) * 8 + ks) * 2 + h) * (size_t)NP * 8 + (size_t)(nt * 32 + r) * 8);
                if (MODE != 1) accp[nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ap[2], b[0], accp[nt], 0, 0, 0);
                if (MODE != 2) acct[nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(at[2], b[0], acct[nt], 0, 0, 0);
                if (MODE != 1) accp[nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ap[0], b[2], accp[nt], 0, 0, 0);
                if (MODE != 2) acct[nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(at[0], b[2], acct[nt], 0, 0, 0);
                if (MODE != 1) accp[nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ap[1], b[1], accp[nt], 0, 0, 0);
                if (MODE != 2) acct[nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(at[1], b[1], acct[nt], 0, 0, 0);
                if (MODE != 1) accp[nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ap[1], b[0], accp[nt], 0, 0, 0);
                if (MODE != 2) acct[nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(at[1], b[0], acct[nt], 0, 0, 0);
                if (MODE != 1) accp[nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ap[0], b[1], accp[nt], 0, 0, 0);
                if (MODE != 2) acct[nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(at[0], b[1], acct[nt], 0, 0, 0);
                if (MODE != 1) accp[nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ap[0], b[0], accp[nt], 0, 0, 0);
                if (MODE != 2) acct[nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(at[0], b[0], acct[nt], 0, 0, 0);
            }
        }
    }
    // sum the four waves' partial tiles through LDS (one 32-column tile at a time), then sigmoid' and the norm
    float *red = reinterpret_cast<float *>(&A3[0][0][0][0]);     // [wave 4][p|t 2][32 rows][33]
    const int row = threadIdx.x >> 3, c4 = (threadIdx.x & 7) * 4;
    const size_t nrow = MODE == 1 ? node_of(row) : slot0 + row;    // sg_node row: the slot's latent (1) / this latent (2)
    // unit tangents: this slot's outputs are column e % unit_d of its latent's Jacobian, kept as jac_node[latent][dimension][NP]
    float *jrow = nullptr;
    if (MODE == 1 && jac_node) {
        const int64_t e = edge_of(row);
        if (e >= 0) jrow = jac_node + (nrow * unit_d + (size_t)(e % unit_d)) * NP;
    }
    double sumsq = 0.0;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        __syncthreads();
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const int rr = (q & 3) + 8 * (q >> 2) + 4 * (lane >> 5);
            if (MODE != 1) red[((wave * 2 + 0) * 32 + rr) * 33 + (lane & 31)] = accp[nt][q];
            if (MODE != 2) red[((wave * 2 + 1) * 32 + rr) * 33 + (lane & 31)] = acct[nt][q];
        }
        __syncthreads();
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int o = nt * 32 + c4 + c;
            if (o < P) {
                float x = b3[o / (s_out * s_out)], t = 0.f;
#pragma unroll
                for (int w = 0; w < 4; ++w) {
                    if (MODE != 1) x += red[((w * 2 + 0) * 32 + row) * 33 + c4 + c];
                    if (MODE != 2) t += red[((w * 2 + 1) * 32 + row) * 33 + c4 + c];
                }
                const float sg = MODE == 1 ? sg_node[nrow * NP + o] : 1.0f / (1.0f + expf(-x));
                if (MODE == 2) sg_node[nrow * NP + o] = sg;
                const float j = t * sg * (1.0f - sg);
                if (MODE == 1 && jrow) jrow[o] = j;
                sumsq += (double)(j * j);
            }
        }
    }
    sumsq += __shfl_xor(sumsq, 1, 64);
    sumsq += __shfl_xor(sumsq, 2, 64);
    sumsq += __shfl_xor(sumsq, 4, 64);
    if (MODE != 2 && (threadIdx.x & 7) == 0) norms[slot0 + row] = (float)sqrt(sumsq);
}

__global__ __launch_bounds__(256) void combine_kernel(const float *__restrict__ norms, int64_t e_base, int64_t e_count,
                                                     int batch, int slots_per_group, float *__restrict__ len_out) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < e_count; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t chunk = i / batch, within = i % batch;
        const float a = norms[(chunk * 2 + 0) * slots_per_group + within];
        const float b = norms[(chunk * 2 + 1) * slots_per_group + within];
        len_out[e_base + i] = 0.5f * (a + b);
    }
}

// ---------------------------------------------------------------------------------- per-latent Jacobians
// Decoders with FIXED statistics and a narrow latent (d <= 16): the decoder Jacobian J(z) (p_out x d) is a function of the
// latent alone, and an edge end's pulled-back length is |J(z_node) (z_dst - z_src)|.  With more than ~d/2 neighbours per latent
// it is cheaper to push the d unit tangents of every latent through the tangent-only pass ONCE (N d slots instead of 2 E) and
// to form the edge ends from the stored columns (SURVEY 2.1 K4': "one metric tensor per latent").  The columns, not the Gram
// matrix G = J^T J: dz^T G dz squares J's condition number into the rounding error, J dz does not, and at p_out = 16 both cost
// d^2 multiply-adds per edge end.
// Pseudo-edge i = (latent pair i / d, dimension i % d): start side latent 2 * pair, end side latent 2 * pair + 1 (an odd last
// latent is its own partner: both sides then write the same column with the same bits).
__global__ __launch_bounds__(256) void pair_edges_kernel(int64_t n_nodes, int d, int64_t n_pseudo, int32_t *__restrict__ src,
                                                        int32_t *__restrict__ dst) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_pseudo; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t a = 2 * (i / d), b = a + 1 < n_nodes ? a + 1 : n_nodes - 1;
        src[i] = (int32_t)a;
        dst[i] = (int32_t)b;
    }
}

// 16 lanes per edge (lane = output element, strided), both ends: v = J(node) dz as an fmaf chain over the latent dimensions
// (ascending), |v| accumulated in fp64 like back_mfma_kernel's norm, length = the mean of the two ends (combine_kernel).
__global__ __launch_bounds__(256) void jacobian_lengths_kernel(const float *__restrict__ z, const int32_t *__restrict__ src,
                                                              const int32_t *__restrict__ dst, int64_t n_edges, int d, int p_out,
                                                              int np, const float *__restrict__ jac_node,
                                                              float *__restrict__ len_out) {
    const int64_t gid = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int l = (int)(gid & 15);
    int64_t e = gid >> 4;
    const bool live = e < n_edges;
    if (!live) e = n_edges - 1;                               // whole 16-lane teams are live or not; shuffles stay uniform
    const int64_t a = src[e], b = dst[e];
    float dz[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) dz[k] = k < d ? z[b * d + k] - z[a * d + k] : 0.f;
    float end_norm[2];
#pragma unroll
    for (int side = 0; side < 2; ++side) {
        const float *J = jac_node + (size_t)(side ? b : a) * d * np;
        double sumsq = 0.0;
        for (int o = l; o < p_out; o += 16) {
            float v = 0.f;
#pragma unroll
            for (int k = 0; k < 16; ++k)
                if (k < d) v = fmaf(dz[k], J[(size_t)k * np + o], v);
            sumsq += (double)(v * v);
        }
        sumsq += __shfl_xor(sumsq, 1, 64);
        sumsq += __shfl_xor(sumsq, 2, 64);
        sumsq += __shfl_xor(sumsq, 4, 64);
        sumsq += __shfl_xor(sumsq, 8, 64);
        end_norm[side] = (float)sqrt(sumsq);
    }
    if (live && l == 0) len_out[e] = 0.5f * (end_norm[0] + end_norm[1]);
}

// ---------------------------------------------------------------------------------- host side
bool make_shape(const geo_decoder_desc *dc, Shape *s) {
    s->d = dc->latent_dim; s->c0 = dc->c0; s->c1 = dc->c1; s->c2 = dc->c2; s->co = dc->out_channels;
    if (dc->out_size == 28) { s->s_out = 4; s->pad3 = 3; }
    else if (dc->out_size == 32) { s->s_out = 8; s->pad3 = 1; }
    else return false;
    s->p_out = s->co * s->s_out * s->s_out;
    s->n1 = 4 * s->c1; s->n2 = 16 * s->c2;
    if (s->c2 <= 0 || NC % s->c2 != 0) return false;
    if (s->c1 <= 0 || 4 * s->c1 > FRONT_MAX_N1) return false;
    s->opix_per_chunk = NC / s->c2;
    if (s->opix_per_chunk > 16) s->opix_per_chunk = 16;
    s->n_chunks = 16 / s->opix_per_chunk;
    return true;
}

// Output pixels ordered so that neighbours in the list are fed by the same input pixels.
void make_chunks(const Shape &s, ChunkTable *t) {
    static const int order[16][2] = {{0, 1}, {0, 2}, {3, 1}, {3, 2}, {1, 0}, {2, 0}, {1, 3}, {2, 3},
                                     {1, 1}, {1, 2}, {2, 1}, {2, 2}, {0, 0}, {0, 3}, {3, 0}, {3, 3}};
    for (int ch = 0; ch < 16; ++ch) { t->nblk[ch] = 0; for (int i = 0; i < 16; ++i) t->opix[ch][i] = 0; }
    for (int ch = 0; ch < s.n_chunks; ++ch) {
        bool used[4] = {false, false, false, false};
        for (int lo = 0; lo < s.opix_per_chunk; ++lo) {
            const int oy = order[ch * s.opix_per_chunk + lo][0], ox = order[ch * s.opix_per_chunk + lo][1];
            t->opix[ch][lo] = oy * 4 + ox;
            for (int iy = 0; iy < 2; ++iy)
                for (int ix = 0; ix < 2; ++ix) {
                    const int ky = oy + 1 - 2 * iy, kx = ox + 1 - 2 * ix;
                    if (ky >= 0 && ky < 4 && kx >= 0 && kx < 4) used[iy * 2 + ix] = true;
                }
        }
        for (int ip = 0; ip < 4; ++ip)
            if (used[ip]) t->ipix[ch][t->nblk[ch]++] = ip;
    }
}

struct Plan {
    Shape sh;
    int batch, tiles_per_group, slots_per_group;
    int64_t chunks_per_pass;
    size_t bytes;
};

constexpr int64_t MAX_SLOTS_PER_PASS = 1 << 20;      // bounds the activation workspace (~12.5 GB; the C2 graph takes two passes)

// bytes of the per-node primal buffers (pre2 of every latent + the sigmoid of every output), 0 when the decoder / sizes do not
// take that path
size_t node_bytes(const Shape &s, int64_t n_nodes) {
    if (n_nodes <= 0) return 0;
    const size_t node_slots = ((size_t)n_nodes + TS - 1) / TS * TS;
    const size_t np = (size_t)((s.p_out + 31) / 32) * 32;
    return geo::align_up(node_slots * s.n2 * 4) + geo::align_up(node_slots * np * 4);
}

bool make_plan(const geo_decoder_desc *dc, int64_t n_edges, int batch, Plan *p) {
    if (!make_shape(dc, &p->sh)) return false;
    p->batch = batch;
    p->tiles_per_group = (batch + TS - 1) / TS;
    p->slots_per_group = p->tiles_per_group * TS;
    const int64_t chunks = (n_edges + batch - 1) / batch;
    int64_t cpp = MAX_SLOTS_PER_PASS / (2 * (int64_t)p->slots_per_group);
    if (cpp < 1) cpp = 1;
    if (cpp > chunks) cpp = chunks > 0 ? chunks : 1;
    p->chunks_per_pass = cpp;
    const Shape &s = p->sh;
    const size_t slots = (size_t)cpp * 2 * p->slots_per_group, tiles = slots / TS, groups = (size_t)cpp * 2;
    size_t b = 0;
    b += geo::align_up((size_t)s.d * s.n1 * 4) + geo::align_up((size_t)s.n1 * 4);          // M01, b01
    b += geo::align_up((size_t)s.n_chunks * MAX_BLOCKS * s.c1 * NC * 4);                   // B2p
    b += geo::align_up((size_t)s.n_chunks * MAX_BLOCKS * s.c1 * NC * 2 * 3);               // B3 (bf16 x 3)
    b += geo::align_up((size_t)16 * s.co * s.c2 * 4);                                       // W3p
    b += geo::align_up((size_t)16 * s.c2 * 192 * 3 * 2);                                    // W3b (bf16 x 3)
    b += 2 * geo::align_up(slots * s.n1 * 4) + 2 * geo::align_up(slots * s.n2 * 4);         // pre1,tpre1,pre2,tpre2
    b += geo::align_up(tiles * s.n1 * 4 * 8) + geo::align_up(tiles * s.n2 * 4 * 8);         // partial sums
    b += geo::align_up((groups + 1) * s.c1 * sizeof(NormConst)) + geo::align_up((groups + 1) * s.c2 * sizeof(NormConst));
    b += geo::align_up((groups + 1) * (size_t)(s.c1 > s.c2 ? s.c1 : s.c2) * sizeof(float2));      // batch statistics
    b += geo::align_up(slots * 4) * 2;                                                      // norms, slot_valid
    if (dc->norm == 2) b += 2 * geo::align_up(slots * 32 * sizeof(float4));                 // GroupNorm statistics
    p->bytes = b + 4096;
    return true;
}

int run_jvp(const geo_decoder_desc *dc, const float *z, int64_t n_nodes, const int32_t *src, const int32_t *dst,
            const float *z_start, const float *z_end, int64_t n_edges, int32_t batch, float *len_out, void *ws,
            size_t ws_bytes, hipStream_t stream, float *jac_node = nullptr) {
    // jac_node != nullptr (run_node_jacobian): the edges are pseudo-edges (latent pair, latent dimension), their tangents unit
    // vectors; the outputs go to jac_node[latent][dimension][np] and no lengths are formed
    GEO_REQUIRE(dc && (len_out || jac_node) && ws, "geo_decoder_jvp: null pointer");
    GEO_REQUIRE(batch > 0, "geo_decoder_jvp: batch_size must be positive");
    if (n_edges == 0) return GEO_OK;
    Plan pl;
    GEO_REQUIRE(make_plan(dc, n_edges, batch, &pl),
                "geo_decoder_jvp: unsupported decoder (out_size %d, c2=%d must divide %d)", dc->out_size, dc->c2, NC);
    const Shape &s = pl.sh;
    GEO_REQUIRE(s.d >= 1 && s.d <= 64, "geo_decoder_jvp: latent_dim %d not in [1,64]", s.d);
    GEO_REQUIRE(s.c1 == 128 || s.c1 == 64 || s.c1 == 32, "geo_decoder_jvp: dec_channels[1]=%d not in {32,64,128}", s.c1);
    GEO_REQUIRE(dc->norm >= 0 && dc->norm <= 2, "geo_decoder_jvp: unknown norm code %d", dc->norm);
    GEO_REQUIRE(s.c2 % 16 == 0, "geo_decoder_jvp: dec_channels[2]=%d must be a multiple of 16", s.c2);
    const size_t back_lds = ((size_t)2 * BACK_TS * 16 * (s.c2 + 4) + (size_t)16 * s.co * (s.c2 + 4) +
                             (size_t)BACK_TS * s.p_out) * 4;
    GEO_REQUIRE(back_lds <= 160 * 1024, "geo_decoder_jvp: decoder too wide for the back kernel (%zu B LDS)", back_lds);
    if (ws_bytes < pl.bytes) {
        geo::set_error("geo_decoder_jvp: workspace %zu < %zu", ws_bytes, pl.bytes);
        return GEO_E_WORKSPACE;
    }
    const size_t slots = (size_t)pl.chunks_per_pass * 2 * pl.slots_per_group, tiles = slots / TS;
    const size_t groups = (size_t)pl.chunks_per_pass * 2;
    geo::Arena ar(ws, ws_bytes);
    float *M01 = ar.take<float>((size_t)s.d * s.n1);
    float *b01 = ar.take<float>((size_t)s.n1);
    float *B2p = ar.take<float>((size_t)s.n_chunks * MAX_BLOCKS * s.c1 * NC);
    unsigned short *B3 = ar.take<unsigned short>((size_t)s.n_chunks * MAX_BLOCKS * s.c1 * NC * 3);
    float *W3p = ar.take<float>((size_t)16 * s.co * s.c2);
    unsigned short *W3b = ar.take<unsigned short>((size_t)16 * s.c2 * 192 * 3);
    float *pre1 = ar.take<float>(slots * s.n1), *tpre1 = ar.take<float>(slots * s.n1);
    float *pre2 = ar.take<float>(slots * s.n2), *tpre2 = ar.take<float>(slots * s.n2);
    double *part1 = ar.take<double>(tiles * s.n1 * 4), *part2 = ar.take<double>(tiles * s.n2 * 4);
    NormConst *k1 = ar.take<NormConst>((groups + 1) * s.c1), *k2 = ar.take<NormConst>((groups + 1) * s.c2);
    float *norms = ar.take<float>(slots);
    float2 *stats = ar.take<float2>((groups + 1) * (size_t)(s.c1 > s.c2 ? s.c1 : s.c2));
    int32_t *slot_valid = ar.take<int32_t>(slots);
    float4 *gs1 = nullptr, *gs2 = nullptr;
    if (dc->norm == 2) {
        gs1 = ar.take<float4>(slots * 32);
        gs2 = ar.take<float4>(slots * 32);
        GEO_REQUIRE(gs2 != nullptr, "geo_decoder_jvp: workspace carve failed");
    }
    GEO_REQUIRE(slot_valid != nullptr, "geo_decoder_jvp: workspace carve failed");
    // per-node primal (see below): the buffers exist when the caller sized the workspace with geo_jvp_edges_workspace_bytes
    const size_t node_slots = n_nodes > 0 ? ((size_t)n_nodes + TS - 1) / TS * TS : 0;
    const size_t np_pad = (size_t)((s.p_out + 31) / 32) * 32;
    float *pre2_node = nullptr, *sg_node = nullptr;
    if (node_slots && node_slots <= slots && ws_bytes >= pl.bytes + node_bytes(s, n_nodes)) {
        pre2_node = ar.take<float>(node_slots * s.n2);
        sg_node = ar.take<float>(node_slots * np_pad);
        if (!sg_node) pre2_node = nullptr;
    }

    ChunkTable tab;
    make_chunks(s, &tab);
    if (s.n_chunks == 8 && s.opix_per_chunk == 2) {          // mid_all_kernel's compiled-in geometry must be this table
        for (int ch = 0; ch < 8; ++ch) {
            int nb = 0;
            for (int ip = 0; ip < 4; ++ip) {
                const int blk = MidGeom::blk_of(ch, ip);
                if (blk >= 0) { GEO_REQUIRE(tab.ipix[ch][blk] == ip, "geo_decoder_jvp: chunk table / MidGeom mismatch (chunk %d)", ch); ++nb; }
            }
            GEO_REQUIRE(nb == tab.nblk[ch] && tab.opix[ch][0] == MidGeom::opix(ch, 0) && tab.opix[ch][1] == MidGeom::opix(ch, 1),
                        "geo_decoder_jvp: chunk table / MidGeom mismatch (chunk %d)", ch);
        }
    }
    compose_front_kernel<<<geo::grid_for((int64_t)(s.d + 1) * s.n1, 256), 256, 0, stream>>>(
        dc->w_in, dc->b_in, dc->w1, dc->b1, s.d, s.c0, s.c1, M01, b01);
    GEO_LAUNCH_CHECK();
    pack_mid_kernel<<<geo::grid_for((int64_t)s.n_chunks * MAX_BLOCKS * s.c1 * NC, 256), 256, 0, stream>>>(
        dc->w2, s.c1, s.c2, tab, s.n_chunks, s.opix_per_chunk, B2p);
    GEO_LAUNCH_CHECK();
    pack_back_kernel<<<geo::grid_for(16 * s.co * s.c2, 256), 256, 0, stream>>>(dc->w3, s.c2, s.co, W3p);
    GEO_LAUNCH_CHECK();
    // ConvT2 on the matrix cores: bf16 x 3 split by default, exact-f32 MFMA with GEO_JVP_MID=f32
    const int mid_opt = geo::options().jvp_mid;
    const bool mid_split = mid_opt != 1 && s.c1 % 16 == 0;
    const int back_nt = (s.p_out + 31) / 32;
    const bool back_mfma = mid_split && s.c2 == 64 && (back_nt == 1 || back_nt == 6) && !geo::options().jvp_back_valu;
    if (back_mfma) {
        pack_back_bf16_kernel<<<geo::grid_for((int64_t)16 * s.c2 * back_nt * 32, 256), 256, 0, stream>>>(
            dc->w3, s.c2, s.co, s.s_out, s.pad3, back_nt * 32, W3b);
        GEO_LAUNCH_CHECK();
    }
    if (mid_split) {
        pack_mid_bf16_kernel<<<geo::grid_for((int64_t)s.n_chunks * MAX_BLOCKS * s.c1 * NC, 256), 256, 0, stream>>>(
            B2p, s.c1, s.n_chunks, B3);
        GEO_LAUNCH_CHECK();
    }
    if (dc->norm == 2) {
        // the GroupNorm path exists for the 32-group layouts of the matrix-core kernels (reference default decoder)
        GEO_REQUIRE(dc->groups1 == 32 && dc->groups2 == 32 && s.c2 == 64 && back_mfma && mid_split && s.n_chunks == 8 &&
                        s.opix_per_chunk == 2 && mid_opt != 2,
                    "geo_decoder_jvp: GroupNorm needs 32 groups per layer and dec_channels[2] == 64 (got %d/%d groups, c2=%d)",
                    dc->groups1, dc->groups2, s.c2);
    }
    const bool batch_stats = dc->norm == 1 && dc->bn_train;
    // train-mode BatchNorm with tracked statistics: the batches are folded into running_mean / running_var in call order
    const bool track = batch_stats && dc->update_running && dc->rm1 && dc->rv1 && dc->rm2 && dc->rv2;
    if (!batch_stats) {
        finalize_fixed_kernel<<<1, 256, 0, stream>>>(s.c1, dc->norm, dc->g1, dc->be1, dc->rm1, dc->rv1, dc->eps, k1);
        GEO_LAUNCH_CHECK();
        finalize_fixed_kernel<<<1, 256, 0, stream>>>(s.c2, dc->norm, dc->g2, dc->be2, dc->rm2, dc->rv2, dc->eps, k2);
        GEO_LAUNCH_CHECK();
    }
    // Per-node primal (round-2 review: "pre-activations depend on the node only"): with FIXED statistics (no norm layer,
    // BatchNorm in eval mode) the whole primal pass is a function of the latent, not of the edge -- 60 000 rows instead of
    // 1.89 M edge ends.  One launch sequence over the latents keeps pre2 and the output sigmoids per node; the edge slots then
    // carry the tangent alone (ConvT2 and ConvT3 at half the matrix work, no primal stores), taking the ReLU masks and
    // sigmoid' from their node's rows.  Same products in the same order: lengths are bit-identical to the per-slot path
    // (`jvp_per_node = 0`).  GroupNorm's statistics are per sample, hence per latent for the primal; the tangent's own group
    // statistics stay per slot (group_stats_kernel with the slot -> latent map).  Train-mode BatchNorm (batch statistics change
    // with the chunk) keeps the per-slot path.
    const bool mid_all_path = mid_split && s.n_chunks == 8 && s.opix_per_chunk == 2 && s.c1 >= 32 && mid_opt != 2;
    const bool per_node = !batch_stats && src && dst && pre2_node && mid_all_path && back_mfma &&
                          geo::options().jvp_per_node != 0;
    GEO_REQUIRE(!jac_node || (per_node && s.d <= 16), "geo_decoder_jvp: per-latent Jacobians need the per-node primal path");
    const int unit_d = jac_node ? s.d : 0;
    if (per_node) {
        const int64_t nt_node = (int64_t)(node_slots / TS);
        const int big_batch = (int)node_slots;                  // one group: every latent on the "start" side
#define GEO_FRONT_N(DM)                                                                                            \
    front_kernel<DM><<<(unsigned)nt_node, 256, 0, stream>>>(nullptr, nullptr, nullptr, z, z, 0, n_nodes, big_batch,   \
                                                            (int)nt_node, s.d, s.n1, M01, b01, pre1, tpre1, part1, 0)
#define GEO_FRONT_MFMA_N(DM)                                                                                       \
    front_mfma_kernel<DM><<<(unsigned)nt_node, 256, 0, stream>>>(nullptr, nullptr, nullptr, z, z, 0, n_nodes,         \
                                                                 big_batch, (int)nt_node, s.d, s.n1, M01, b01, pre1,  \
                                                                 tpre1, part1, 0)
        const bool front_mfma_n = s.d > 16 && s.n1 % 32 == 0 && geo::options().jvp_front_valu == 0;
        if (s.d <= 16) GEO_FRONT_N(16);
        else if (s.d <= 32) { if (front_mfma_n) GEO_FRONT_MFMA_N(32); else GEO_FRONT_N(32); }
        else { if (front_mfma_n) GEO_FRONT_MFMA_N(64); else GEO_FRONT_N(64); }
#undef GEO_FRONT_MFMA_N
#undef GEO_FRONT_N
        GEO_LAUNCH_CHECK();
        if (gs1) {
            group_stats_kernel<<<geo::grid_for((int64_t)node_slots * 32, 256), 256, 0, stream>>>(pre1, tpre1, (int64_t)node_slots, 4,
                                                                                                 s.c1, 32, dc->eps, gs1);
            GEO_LAUNCH_CHECK();
        }
#define GEO_MIDA_N(C1V, GNV)                                                                                       \
    mid_all_kernel<C1V, GNV><<<(unsigned)nt_node, 512, 0, stream>>>(pre1, tpre1, k1, 0, (int)nt_node, s.c2, B3,         \
                                                                    dc->b2, pre2_node, tpre2, part2, 0, slot_valid,   \
                                                                    gs1, 0, n_nodes, big_batch)
        if (gs1) { if (s.c1 == 128) GEO_MIDA_N(128, true); else if (s.c1 == 64) GEO_MIDA_N(64, true); else GEO_MIDA_N(32, true); }
        else if (s.c1 == 128) GEO_MIDA_N(128, false);
        else if (s.c1 == 64) GEO_MIDA_N(64, false);
        else GEO_MIDA_N(32, false);
#undef GEO_MIDA_N
        GEO_LAUNCH_CHECK();
        if (gs2) {
            group_stats_kernel<<<geo::grid_for((int64_t)node_slots * 32, 256), 256, 0, stream>>>(pre2_node, tpre2, (int64_t)node_slots,
                                                                                                 16, s.c2, 32, dc->eps, gs2);
            GEO_LAUNCH_CHECK();
        }
#define GEO_BACK_N(NTV, GNV)                                                                                       \
    back_mfma_kernel<NTV, GNV, 2><<<(unsigned)nt_node, 256, 0, stream>>>(pre2_node, tpre2, k2, 0, (int)nt_node, s.co,   \
                                                                         s.s_out, W3b, dc->b3, norms, gs2, sg_node)
        if (back_nt == 1) { if (gs2) GEO_BACK_N(1, true); else GEO_BACK_N(1, false); }
        else { if (gs2) GEO_BACK_N(6, true); else GEO_BACK_N(6, false); }
#undef GEO_BACK_N
        GEO_LAUNCH_CHECK();
    }
    const int64_t total_chunks = (n_edges + batch - 1) / batch;
    for (int64_t c0 = 0; c0 < total_chunks; c0 += pl.chunks_per_pass) {
        int64_t nch = total_chunks - c0;
        if (nch > pl.chunks_per_pass) nch = pl.chunks_per_pass;
        const int64_t e_base = c0 * batch;
        int64_t e_count = n_edges - e_base;
        if (e_count > nch * batch) e_count = nch * batch;
        const int64_t p_groups = nch * 2, p_tiles = p_groups * pl.tiles_per_group, p_slots = p_tiles * TS;
        slot_valid_kernel<<<geo::grid_for(p_slots, 256), 256, 0, stream>>>(e_base, n_edges, batch, pl.tiles_per_group,
                                                                          p_slots, slot_valid);
        GEO_LAUNCH_CHECK();
#define GEO_FRONT(DM)                                                                                              \
    front_kernel<DM><<<(unsigned)p_tiles, 256, 0, stream>>>(z, src, dst, z_start, z_end, e_base, n_edges, batch,    \
                                                            pl.tiles_per_group, s.d, s.n1, M01, b01, pre1, tpre1,   \
                                                            part1, batch_stats ? 1 : 0, unit_d)
#define GEO_FRONT_MFMA(DM)                                                                                         \
    front_mfma_kernel<DM><<<(unsigned)p_tiles, 256, 0, stream>>>(z, src, dst, z_start, z_end, e_base, n_edges, batch, \
                                                                 pl.tiles_per_group, s.d, s.n1, M01, b01, pre1,       \
                                                                 tpre1, part1, batch_stats ? 1 : 0)
        const bool front_mfma = s.d > 16 && s.n1 % 32 == 0 && geo::options().jvp_front_valu == 0;
        if (s.d <= 16) { if (geo::options().jvp_front_valu == 2 && s.n1 % 32 == 0 && !unit_d) GEO_FRONT_MFMA(16); else GEO_FRONT(16); }
        else if (s.d <= 32) { if (front_mfma) GEO_FRONT_MFMA(32); else GEO_FRONT(32); }
        else { if (front_mfma) GEO_FRONT_MFMA(64); else GEO_FRONT(64); }
#undef GEO_FRONT_MFMA
#undef GEO_FRONT
        GEO_LAUNCH_CHECK();
        if (batch_stats) {
            finalize_batch_kernel<<<geo::grid_for(p_groups * s.c1, 256), 256, 0, stream>>>(
                part1, pl.tiles_per_group, 1, 4, s.c1, e_base, n_edges, batch, dc->g1, dc->be1, dc->eps, k1, (int)p_groups,
                track ? stats : nullptr);
            if (track)
                running_update_kernel<<<(unsigned)((s.c1 + 7) / 8), 256, 0, stream>>>(stats, (int)p_groups, s.c1, dc->momentum,
                                                                          const_cast<float *>(dc->rm1), const_cast<float *>(dc->rv1));
            GEO_LAUNCH_CHECK();
        }
        if (gs1) {
            group_stats_kernel<<<geo::grid_for(p_slots * 32, 256), 256, 0, stream>>>(pre1, tpre1, p_slots, 4, s.c1, 32,
                                                                                    dc->eps, gs1);
            GEO_LAUNCH_CHECK();
        }
        const dim3 mgrid((unsigned)p_tiles, (unsigned)s.n_chunks);
#define GEO_MID(C1V)                                                                                               \
    mid_kernel<C1V><<<mgrid, 256, 0, stream>>>(pre1, tpre1, k1, batch_stats ? 1 : 0, pl.tiles_per_group, tab,       \
                                               s.opix_per_chunk, s.c2, B2p, dc->b2, pre2, tpre2, part2,             \
                                               batch_stats ? 1 : 0, slot_valid)
#define GEO_MID3(C1V)                                                                                              \
    mid_bf16_kernel<C1V><<<mgrid, 256, 0, stream>>>(pre1, tpre1, k1, batch_stats ? 1 : 0, pl.tiles_per_group, tab,  \
                                                    s.opix_per_chunk, s.c2, B3, dc->b2, pre2, tpre2, part2,        \
                                                    batch_stats ? 1 : 0, slot_valid)
        const bool mid_all = mid_split && s.n_chunks == 8 && s.opix_per_chunk == 2 && s.c1 >= 32 &&
                             mid_opt != 2;
        // the shipped widths with BatchNorm / no norm, primal + tangent: the persistent skewed kernel (jvp_mid = 3: mid_all_kernel)
        const bool mid_pipe = mid_all && !per_node && !gs1 && s.c1 == 128 && s.c2 == 64 && mid_opt != 3;
        if (mid_all && per_node) {
#define GEO_MIDA_T(C1V, GNV)                                                                                       \
    mid_all_kernel<C1V, GNV, true><<<(unsigned)p_tiles, 512, 0, stream>>>(pre1, tpre1, k1, 0, pl.tiles_per_group,       \
                                                                          s.c2, B3, dc->b2, pre2, tpre2, part2, 0,     \
                                                                          slot_valid, gs1, e_base, n_edges, batch)
            if (gs1) { if (s.c1 == 128) GEO_MIDA_T(128, true); else if (s.c1 == 64) GEO_MIDA_T(64, true); else GEO_MIDA_T(32, true); }
            else if (s.c1 == 128) GEO_MIDA_T(128, false);
            else if (s.c1 == 64) GEO_MIDA_T(64, false);
            else GEO_MIDA_T(32, false);
#undef GEO_MIDA_T
        } else if (mid_pipe) {
            static int n_cu = 0;
            if (n_cu == 0) {
                int devid = 0;
                hipDeviceProp_t prop;
                GEO_HIP_CHECK(hipGetDevice(&devid));
                GEO_HIP_CHECK(hipGetDeviceProperties(&prop, devid));
                n_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
            }
            const int per_cu = geo::options().jvp_pipe_grid > 0 ? geo::options().jvp_pipe_grid : 1;
            const unsigned pgrid = (unsigned)std::min<int64_t>(p_tiles, (int64_t)n_cu * per_cu);   // persistent workgroups, one resident per CU
            mid_pipe_kernel<<<pgrid, 512, 0, stream>>>(pre1, tpre1, k1, batch_stats ? 1 : 0, pl.tiles_per_group, (int)p_tiles,
                                                       s.c2, B3, dc->b2, pre2, tpre2, part2, batch_stats ? 1 : 0, e_base,
                                                       n_edges, batch);
        } else if (mid_all) {
#define GEO_MIDA(C1V, GNV)                                                                                         \
    mid_all_kernel<C1V, GNV><<<(unsigned)p_tiles, 512, 0, stream>>>(pre1, tpre1, k1, batch_stats ? 1 : 0,           \
                                                                    pl.tiles_per_group, s.c2, B3, dc->b2,           \
                                                                    pre2, tpre2, part2, batch_stats ? 1 : 0,        \
                                                                    slot_valid, gs1, e_base, n_edges, batch)
            if (gs1) {
                if (s.c1 == 128) GEO_MIDA(128, true);
                else if (s.c1 == 64) GEO_MIDA(64, true);
                else GEO_MIDA(32, true);
            } else if (s.c1 == 128) GEO_MIDA(128, false);
            else if (s.c1 == 64) GEO_MIDA(64, false);
            else GEO_MIDA(32, false);
#undef GEO_MIDA
        } else if (mid_split) {
            if (s.c1 == 128) GEO_MID3(128);
            else if (s.c1 == 64) GEO_MID3(64);
            else GEO_MID3(32);
        } else {
            if (s.c1 == 128) GEO_MID(128);
            else if (s.c1 == 64) GEO_MID(64);
            else GEO_MID(32);
        }
#undef GEO_MID3
#undef GEO_MID
        GEO_LAUNCH_CHECK();
        if (batch_stats) {
            finalize_batch_kernel<<<geo::grid_for(p_groups * s.c2, 256), 256, 0, stream>>>(
                part2, pl.tiles_per_group, mid_pipe ? 4 : (mid_all ? 1 : 16), 16, s.c2, e_base, n_edges, batch, dc->g2, dc->be2, dc->eps, k2, (int)p_groups,
                track ? stats : nullptr);
            if (track)
                running_update_kernel<<<(unsigned)((s.c2 + 7) / 8), 256, 0, stream>>>(stats, (int)p_groups, s.c2, dc->momentum,
                                                                          const_cast<float *>(dc->rm2), const_cast<float *>(dc->rv2));
            GEO_LAUNCH_CHECK();
        }
        if (gs2 && per_node) {                                  // (slot_valid is not read on this path: it carries the slot -> latent map)
            slot_node_kernel<<<geo::grid_for(p_slots, 256), 256, 0, stream>>>(src, dst, e_base, n_edges, batch, pl.tiles_per_group,
                                                                             p_slots, slot_valid);
            group_stats_kernel<<<geo::grid_for(p_slots * 32, 256), 256, 0, stream>>>(pre2_node, tpre2, p_slots, 16, s.c2, 32,
                                                                                    dc->eps, gs2, slot_valid);
            GEO_LAUNCH_CHECK();
        } else if (gs2) {
            group_stats_kernel<<<geo::grid_for(p_slots * 32, 256), 256, 0, stream>>>(pre2, tpre2, p_slots, 16, s.c2, 32,
                                                                                    dc->eps, gs2);
            GEO_LAUNCH_CHECK();
        }
#define GEO_BACK(NTV, GNV)                                                                                         \
    back_mfma_kernel<NTV, GNV><<<(unsigned)p_tiles, 256, 0, stream>>>(pre2, tpre2, k2, batch_stats ? 1 : 0,         \
                                                                      pl.tiles_per_group, s.co, s.s_out, W3b,       \
                                                                      dc->b3, norms, gs2)
#define GEO_BACK_T(NTV, GNV)                                                                                       \
    back_mfma_kernel<NTV, GNV, 1><<<(unsigned)p_tiles, 256, 0, stream>>>(pre2_node, tpre2, k2, 0, pl.tiles_per_group, s.co, \
                                                                         s.s_out, W3b, dc->b3, norms, gs2, sg_node, src,   \
                                                                         dst, e_base, n_edges, batch, jac_node, unit_d)
        if (per_node && back_nt == 1) { if (gs2) GEO_BACK_T(1, true); else GEO_BACK_T(1, false); }
        else if (per_node) { if (gs2) GEO_BACK_T(6, true); else GEO_BACK_T(6, false); }
#undef GEO_BACK_T
        else if (back_mfma && back_nt == 1) { if (gs2) GEO_BACK(1, true); else GEO_BACK(1, false); }
        else if (back_mfma) { if (gs2) GEO_BACK(6, true); else GEO_BACK(6, false); }
#undef GEO_BACK
        else
            back_kernel<<<(unsigned)(p_slots / BACK_TS), 256, back_lds, stream>>>(
                pre2, tpre2, k2, batch_stats ? 1 : 0, pl.slots_per_group, s.c2, s.co, s.s_out, s.pad3, W3p, dc->b3, norms);
        GEO_LAUNCH_CHECK();
        if (len_out) {
            combine_kernel<<<geo::grid_for(e_count, 256), 256, 0, stream>>>(norms, e_base, e_count, batch,
                                                                            pl.slots_per_group, len_out);
            GEO_LAUNCH_CHECK();
        }
    }
    return GEO_OK;
}

// Per-latent Jacobians (kernels above): which calls take that route, its workspace, and the run.
struct JacobianPlan {
    int64_t n_pseudo;
    size_t np, extra_bytes, bytes;                            // extra: pseudo-edge lists + the columns; bytes: extra + inner run
};

bool node_jacobian_plan(const geo_decoder_desc *dc, int64_t n_nodes, int64_t n_edges, int batch, JacobianPlan *jp) {
    const geo::Options &o = geo::options();
    Shape s;
    if (!dc || n_nodes <= 0 || n_edges <= 0 || !make_shape(dc, &s)) return false;
    if (o.jvp_node_jacobian == 0 || o.jvp_per_node == 0 || o.jvp_back_valu || o.jvp_mid == 1 || o.jvp_mid == 2) return false;
    if (dc->norm == 1 && dc->bn_train) return false;          // batch statistics: the Jacobian depends on the edge's batch
    if (s.d > 16 || s.c1 % 16 != 0 || s.c1 < 32 || s.c2 != 64 || s.n_chunks != 8 || s.opix_per_chunk != 2) return false;
    const int back_nt = (s.p_out + 31) / 32;
    if (back_nt != 1 && back_nt != 6) return false;
    // worth it from ~1.5 x fewer slots (the columns pass stores and the edge pass re-reads p_out x d floats per latent)
    if (o.jvp_node_jacobian == 1 && 4 * n_edges < 3 * n_nodes * s.d) return false;
    jp->n_pseudo = (n_nodes + 1) / 2 * s.d;
    if (jp->n_pseudo >= (int64_t)1 << 31) return false;
    jp->np = (size_t)back_nt * 32;
    Plan pl;
    if (!make_plan(dc, jp->n_pseudo, batch, &pl)) return false;
    const size_t node_slots = ((size_t)n_nodes + TS - 1) / TS * TS;
    if (node_slots > (size_t)pl.chunks_per_pass * 2 * pl.slots_per_group) return false;     // per-node primal would not fit a pass
    jp->extra_bytes = 2 * geo::align_up((size_t)jp->n_pseudo * 4) + geo::align_up((size_t)n_nodes * s.d * jp->np * 4);
    jp->bytes = jp->extra_bytes + pl.bytes + node_bytes(s, n_nodes);
    return true;
}

int run_node_jacobian(const geo_decoder_desc *dc, const JacobianPlan &jp, const float *z, int64_t n_nodes, const int32_t *src,
                      const int32_t *dst, int64_t n_edges, int32_t batch, float *len_out, void *ws, size_t ws_bytes,
                      hipStream_t stream) {
    Shape s;
    make_shape(dc, &s);
    geo::Arena ar(ws, ws_bytes);
    int32_t *psrc = ar.take<int32_t>((size_t)jp.n_pseudo), *pdst = ar.take<int32_t>((size_t)jp.n_pseudo);
    float *jac = ar.take<float>((size_t)n_nodes * s.d * jp.np);
    GEO_REQUIRE(jac != nullptr, "geo_decoder_jvp_edges: workspace carve failed");
    pair_edges_kernel<<<geo::grid_for(jp.n_pseudo, 256), 256, 0, stream>>>(n_nodes, s.d, jp.n_pseudo, psrc, pdst);
    GEO_LAUNCH_CHECK();
    const int rc = run_jvp(dc, z, n_nodes, psrc, pdst, nullptr, nullptr, jp.n_pseudo, batch, nullptr,
                           static_cast<char *>(ws) + ar.off, ws_bytes - ar.off, stream, jac);
    if (rc != GEO_OK) return rc;
    const int64_t teams = (n_edges * 16 + 255) / 256;
    GEO_REQUIRE(teams < ((int64_t)1 << 31), "geo_decoder_jvp_edges: too many edges for one launch (%lld)", (long long)n_edges);
    jacobian_lengths_kernel<<<(unsigned)teams, 256, 0, stream>>>(z, src, dst, n_edges, s.d, s.p_out, (int)jp.np, jac, len_out);
    GEO_LAUNCH_CHECK();
    return GEO_OK;
}

}  // namespace

#ifdef GEO_MID_PROF
extern "C" int geo_debug_mid_prof(unsigned long long *out, int reset) {
    if (reset) { unsigned long long z[16] = {0}; return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_mid_prof), z, sizeof(z)); }
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_mid_prof), sizeof(unsigned long long) * 16);
}
#endif

extern "C" size_t geo_jvp_workspace_bytes(const geo_decoder_desc *dec, int64_t n_edges, int32_t batch_size) {
    Plan pl;
    if (!dec || batch_size <= 0 || n_edges < 0 || !make_plan(dec, n_edges, batch_size, &pl)) return 0;
    return pl.bytes;
}

extern "C" size_t geo_jvp_edges_workspace_bytes(const geo_decoder_desc *dec, int64_t n_nodes, int64_t n_edges,
                                                int32_t batch_size) {
    Plan pl;
    if (!dec || batch_size <= 0 || n_edges < 0 || n_nodes < 0 || !make_plan(dec, n_edges, batch_size, &pl)) return 0;
    const size_t per_edge_end = pl.bytes + node_bytes(pl.sh, n_nodes);
    JacobianPlan jp;
    if (node_jacobian_plan(dec, n_nodes, n_edges, batch_size, &jp) && jp.bytes > per_edge_end) return jp.bytes;
    return per_edge_end;
}

extern "C" int geo_decoder_jvp_edges(const geo_decoder_desc *dec, const float *z, int64_t n_nodes, const int32_t *src,
                                     const int32_t *dst, int64_t n_edges, int32_t batch_size, float *len_out, void *ws,
                                     size_t ws_bytes, void *stream) {
    GEO_REQUIRE(n_edges == 0 || (z && src && dst && n_nodes > 0), "geo_decoder_jvp_edges: null pointer");
    JacobianPlan jp;
    if (batch_size > 0 && len_out && ws && node_jacobian_plan(dec, n_nodes, n_edges, batch_size, &jp) && ws_bytes >= jp.bytes)
        return run_node_jacobian(dec, jp, z, n_nodes, src, dst, n_edges, batch_size, len_out, ws, ws_bytes,
                                 static_cast<hipStream_t>(stream));
    return run_jvp(dec, z, n_nodes, src, dst, nullptr, nullptr, n_edges, batch_size, len_out, ws, ws_bytes,
                   static_cast<hipStream_t>(stream));
}

extern "C" int geo_decoder_jvp_pairs(const geo_decoder_desc *dec, const float *z_start, const float *z_end,
                                     int64_t n_edges, int32_t batch_size, float *len_out, void *ws, size_t ws_bytes,
                                     void *stream) {
    GEO_REQUIRE(n_edges == 0 || (z_start && z_end), "geo_decoder_jvp_pairs: null pointer");
    return run_jvp(dec, nullptr, 0, nullptr, nullptr, z_start, z_end, n_edges, batch_size, len_out, ws, ws_bytes,
                   static_cast<hipStream_t>(stream));
}

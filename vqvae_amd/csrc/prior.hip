// prior.hip -- hand-written pieces of the code prior's training step (gfx950).
//
// The reference trains a 4-layer, 256-wide Transformer over 15-token code sequences with stock PyTorch
// (src/models/transformer.py:98-133 attention, src/scripts/train_transformer.py:39-44,60-66 AdamW step).  At that size a
// step is ~300 kernel launches over tensors of a few hundred kilobytes: the step is bound by launches and by round trips
// of tiny intermediates, not by arithmetic.  Two kernels take the worst of it:
//
//   prior_attention_{fwd,bwd}: causal multi-head attention for sequences of at most 16 tokens, one wave per (sample, head):
//       q, k, v of the head live in LDS, scores / softmax / dropout / context (and their whole backward) never leave the CU.
//       Replaces bmm + scale + masked_fill + softmax + dropout + bmm (6 launches, 5 intermediates; 11 more in backward).
//   prior_adamw: the decoupled-weight-decay Adam step over the model's ONE flat parameter arena: one launch, one pass
//       over 4 arrays (torch's foreach AdamW: ~10 launches over ~40 tensors each).  lr and the step count are read from
//       device memory, so the launch is identical every step (HIP-graph friendly).
//
// float32 throughout, fmaf accumulation; results agree with torch's to rounding (tests compare loss curves at 2e-4).
#include "geo_common.h"

#include <cmath>

namespace {

constexpr int PT_MAX = 16;            // tokens per sequence (max_seq_len of the reference configs: 4x4 grid)

// qkv: [B][T][3][H][HD] (the c_attn projection's output, viewed), out: [B][T][H*HD].
// probs: [B][H][T][T] softmax rows (before dropout; zeros above the diagonal) kept for the backward pass.
// keep: [B][H][T][T] bytes (1 = kept) or null; kept probabilities are scaled by keep_scale = 1 / (1 - p).
template <int HD>
__global__ __launch_bounds__(256) void prior_attention_fwd_kernel(const float *__restrict__ qkv, const uint8_t *__restrict__ keep,
                                                                 float keep_scale, int B, int T, int H, float scale,
                                                                 float *__restrict__ out, float *__restrict__ probs) {
    __shared__ float sq[4][PT_MAX][HD + 1], sk[4][PT_MAX][HD + 1], sv[4][PT_MAX][HD + 1];
    __shared__ float sp[4][PT_MAX][PT_MAX + 1];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int bh = blockIdx.x * 4 + wave;
    if (bh >= B * H) return;                                   // (whole wave; no block barrier below)
    const int b = bh / H, h = bh % H;
    const int C = H * HD;
    float(*q)[HD + 1] = sq[wave], (*k)[HD + 1] = sk[wave], (*v)[HD + 1] = sv[wave];
    float(*p)[PT_MAX + 1] = sp[wave];
    for (int i = lane; i < T * HD; i += 64) {
        const int t = i / HD, l = i % HD;
        const float *row = qkv + ((size_t)(b * T + t) * 3) * C + h * HD + l;
        q[t][l] = row[0];
        k[t][l] = row[C];
        v[t][l] = row[2 * C];
    }
    __builtin_amdgcn_wave_barrier();
    // scores of the causal pairs (i, j <= i): a lane per pair
    for (int pair = lane; pair < T * T; pair += 64) {
        const int i = pair / T, j = pair % T;
        float s = 0.f;
        if (j <= i) {
#pragma unroll 8
            for (int l = 0; l < HD; ++l) s = fmaf(q[i][l], k[j][l], s);
            s *= scale;
        }
        p[i][j] = s;
    }
    __builtin_amdgcn_wave_barrier();
    if (lane < T) {                                            // a lane per row: softmax over j <= i
        const int i = lane;
        float m = p[i][0];
        for (int j = 1; j <= i; ++j) m = fmaxf(m, p[i][j]);
        float sum = 0.f;
        for (int j = 0; j <= i; ++j) { const float e = expf(p[i][j] - m); p[i][j] = e; sum += e; }
        const float inv = 1.0f / sum;
        float *pr = probs + ((size_t)bh * T + i) * T;
        const uint8_t *kp = keep ? keep + ((size_t)bh * T + i) * T : nullptr;
        for (int j = 0; j < T; ++j) {
            const float pj = j <= i ? p[i][j] * inv : 0.f;
            pr[j] = pj;
            p[i][j] = kp ? (kp[j] ? pj * keep_scale : 0.f) : pj;         // what multiplies v
        }
    }
    __builtin_amdgcn_wave_barrier();
    for (int l = lane; l < HD; l += 64)                        // context: a lane per head column
        for (int i = 0; i < T; ++i) {
            float acc = 0.f;
            for (int j = 0; j <= i; ++j) acc = fmaf(p[i][j], v[j][l], acc);
            out[(size_t)(b * T + i) * C + h * HD + l] = acc;
        }
}

// dqkv: [B][T][3][H][HD].  dS = P * (dP - rowsum(dP * P)) * scale with dP = (dO v^T) * keep * keep_scale.
template <int HD, int WPB>
__global__ __launch_bounds__(64 * WPB) void prior_attention_bwd_kernel(const float *__restrict__ qkv, const float *__restrict__ probs,
                                                                 const uint8_t *__restrict__ keep, float keep_scale,
                                                                 const float *__restrict__ dout, int B, int T, int H, float scale,
                                                                 float *__restrict__ dqkv) {
    __shared__ float sq[WPB][PT_MAX][HD + 1], sk[WPB][PT_MAX][HD + 1], sv[WPB][PT_MAX][HD + 1], sdo[WPB][PT_MAX][HD + 1];
    __shared__ float sp[WPB][PT_MAX][PT_MAX + 1], sds[WPB][PT_MAX][PT_MAX + 1];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int bh = blockIdx.x * WPB + wave;
    if (bh >= B * H) return;
    const int b = bh / H, h = bh % H;
    const int C = H * HD;
    float(*q)[HD + 1] = sq[wave], (*k)[HD + 1] = sk[wave], (*v)[HD + 1] = sv[wave], (*go)[HD + 1] = sdo[wave];
    float(*pd)[PT_MAX + 1] = sp[wave], (*ds)[PT_MAX + 1] = sds[wave];      // pd: dropped probabilities P'
    for (int i = lane; i < T * HD; i += 64) {
        const int t = i / HD, l = i % HD;
        const float *row = qkv + ((size_t)(b * T + t) * 3) * C + h * HD + l;
        q[t][l] = row[0];
        k[t][l] = row[C];
        v[t][l] = row[2 * C];
        go[t][l] = dout[(size_t)(b * T + t) * C + h * HD + l];
    }
    __builtin_amdgcn_wave_barrier();
    // P' = P * keep * keep_scale (what multiplied v in the forward pass; needed for dV): a lane per causal pair
    for (int pair = lane; pair < T * T; pair += 64) {
        const int i = pair / T, j = pair % T;
        float pp = 0.f;
        if (j <= i) {
            const float kf = keep ? (keep[((size_t)bh * T + i) * T + j] ? keep_scale : 0.f) : 1.0f;
            pp = probs[((size_t)bh * T + i) * T + j] * kf;
        }
        pd[i][j] = pp;
    }
    // dS: a lane per row.  dP[i][j] = <dO[i], v[j]> * keep * keep_scale;  dS = P * (dP - sum_k dP[i][k] P[i][k]) * scale
    if (lane < T) {
        const int i = lane;
        float gi[PT_MAX], pi[PT_MAX];                          // (fully unrolled: both stay in registers)
        float rowsum = 0.f;
#pragma unroll
        for (int j = 0; j < PT_MAX; ++j) {
            gi[j] = 0.f;
            pi[j] = 0.f;
            if (j < T && j <= i) {
                pi[j] = probs[((size_t)bh * T + i) * T + j];
                const float kf = keep ? (keep[((size_t)bh * T + i) * T + j] ? keep_scale : 0.f) : 1.0f;
                float dp = 0.f;
                for (int l = 0; l < HD; ++l) dp = fmaf(go[i][l], v[j][l], dp);
                gi[j] = dp * kf;
                rowsum = fmaf(gi[j], pi[j], rowsum);
            }
        }
#pragma unroll
        for (int j = 0; j < PT_MAX; ++j)
            if (j < T) ds[i][j] = j <= i ? pi[j] * (gi[j] - rowsum) * scale : 0.f;       // dS, zeros above the diagonal
    }
    __builtin_amdgcn_wave_barrier();
    for (int l = lane; l < HD; l += 64) {
        for (int t = 0; t < T; ++t) {
            float dq = 0.f, dk = 0.f, dv = 0.f;
            for (int j = 0; j <= t; ++j) dq = fmaf(ds[t][j], k[j][l], dq);            // dQ[t] = sum_j dS[t][j] k[j]
            for (int i = t; i < T; ++i) {
                dk = fmaf(ds[i][t], q[i][l], dk);                                       // dK[t] = sum_i dS[i][t] q[i]
                dv = fmaf(pd[i][t], go[i][l], dv);                                      // dV[t] = sum_i P'[i][t] dO[i]
            }
            float *row = dqkv + ((size_t)(b * T + t) * 3) * C + h * HD + l;
            row[0] = dq;
            row[C] = dk;
            row[2 * C] = dv;
        }
    }
}

// torch.optim.AdamW's step on one flat buffer (decoupled weight decay, bias-corrected moments); lr and step on the device.
__global__ __launch_bounds__(256) void prior_adamw_kernel(float *__restrict__ p, const float *__restrict__ g, float *__restrict__ m,
                                                         float *__restrict__ v, int64_t n, const float *__restrict__ lr_dev,
                                                         const int64_t *__restrict__ step_dev, float beta1, float beta2, float eps,
                                                         float weight_decay) {
    const double lr = (double)lr_dev[0];
    const double step = (double)step_dev[0];
    const double bc1 = 1.0 - pow((double)beta1, step), bc2 = 1.0 - pow((double)beta2, step);
    const float decay = (float)(1.0 - lr * (double)weight_decay);
    const float step_size = (float)(lr / bc1), bc2_sqrt = (float)sqrt(bc2);
    const float w1 = (float)(1.0 - (double)beta1), w2 = (float)(1.0 - (double)beta2);
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const float gi = g[i];
        const float pi = p[i] * decay;
        const float mi = m[i] + w1 * (gi - m[i]);               // lerp(m, g, 1 - beta1)
        const float vi = fmaf(w2 * gi, gi, v[i] * beta2);       // v * beta2 + (1 - beta2) g g
        m[i] = mi;
        v[i] = vi;
        const float denom = sqrtf(vi) / bc2_sqrt + eps;
        p[i] = pi - step_size * (mi / denom);
    }
}

}  // namespace

extern "C" int geo_prior_attention_fwd(const float *qkv, const uint8_t *keep, float keep_scale, int32_t B, int32_t T, int32_t H,
                                       int32_t head_dim, float *out, float *probs, void *stream_) {
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    GEO_REQUIRE(qkv && out && probs, "geo_prior_attention_fwd: null pointer");
    GEO_REQUIRE(B > 0 && H > 0 && T >= 1 && T <= PT_MAX, "geo_prior_attention_fwd: B=%d H=%d T=%d (T <= %d)", B, H, T, PT_MAX);
    const float scale = 1.0f / sqrtf((float)head_dim);
    const unsigned grid = (unsigned)((B * H + 3) / 4);
    switch (head_dim) {
        case 16: prior_attention_fwd_kernel<16><<<grid, 256, 0, stream>>>(qkv, keep, keep_scale, B, T, H, scale, out, probs); break;
        case 32: prior_attention_fwd_kernel<32><<<grid, 256, 0, stream>>>(qkv, keep, keep_scale, B, T, H, scale, out, probs); break;
        case 64: prior_attention_fwd_kernel<64><<<grid, 256, 0, stream>>>(qkv, keep, keep_scale, B, T, H, scale, out, probs); break;
        default: geo::set_error("geo_prior_attention_fwd: head_dim %d not in {16, 32, 64}", head_dim); return GEO_E_ARG;
    }
    GEO_LAUNCH_CHECK();
    return GEO_OK;
}

extern "C" int geo_prior_attention_bwd(const float *qkv, const float *probs, const uint8_t *keep, float keep_scale, const float *dout,
                                       int32_t B, int32_t T, int32_t H, int32_t head_dim, float *dqkv, void *stream_) {
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    GEO_REQUIRE(qkv && probs && dout && dqkv, "geo_prior_attention_bwd: null pointer");
    GEO_REQUIRE(B > 0 && H > 0 && T >= 1 && T <= PT_MAX, "geo_prior_attention_bwd: B=%d H=%d T=%d (T <= %d)", B, H, T, PT_MAX);
    const float scale = 1.0f / sqrtf((float)head_dim);
    const unsigned grid4 = (unsigned)((B * H + 3) / 4), grid2 = (unsigned)((B * H + 1) / 2);
    switch (head_dim) {
        case 16: prior_attention_bwd_kernel<16, 4><<<grid4, 256, 0, stream>>>(qkv, probs, keep, keep_scale, dout, B, T, H, scale, dqkv); break;
        case 32: prior_attention_bwd_kernel<32, 4><<<grid4, 256, 0, stream>>>(qkv, probs, keep, keep_scale, dout, B, T, H, scale, dqkv); break;
        case 64: prior_attention_bwd_kernel<64, 2><<<grid2, 128, 0, stream>>>(qkv, probs, keep, keep_scale, dout, B, T, H, scale, dqkv); break;   // (LDS: 4 x 16 x 65 floats per wave)
        default: geo::set_error("geo_prior_attention_bwd: head_dim %d not in {16, 32, 64}", head_dim); return GEO_E_ARG;
    }
    GEO_LAUNCH_CHECK();
    return GEO_OK;
}

extern "C" int geo_prior_adamw(float *param, const float *grad, float *exp_avg, float *exp_avg_sq, int64_t n, const float *lr_dev,
                               const int64_t *step_dev, float beta1, float beta2, float eps, float weight_decay, void *stream_) {
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    GEO_REQUIRE(param && grad && exp_avg && exp_avg_sq && lr_dev && step_dev && n > 0, "geo_prior_adamw: bad arguments");
    prior_adamw_kernel<<<geo::grid_for(n, 256, 2048), 256, 0, stream>>>(param, grad, exp_avg, exp_avg_sq, n, lr_dev, step_dev, beta1,
                                                                      beta2, eps, weight_decay);
    GEO_LAUNCH_CHECK();
    return GEO_OK;
}

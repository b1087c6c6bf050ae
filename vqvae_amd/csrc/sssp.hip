// sssp.hip -- fp64 label-correcting shortest paths on a pull CSR (gfx950).
//
// Replaces scipy.sparse.csgraph.dijkstra as used by the reference
// (src/geo/geo_shortest_paths.py:36-49; src/geo/kmeans_optimized.py:43,97,125).
//
// Why label-correcting reproduces Dijkstra bit for bit: every value ever stored in dist[] is the
// left-to-right fp64 sum of the weights along some real path from the source (one rounding per
// hop), and fl(a + w) is monotone in a.  The iteration d[v] <- min(d[v], min_u fl(d[u] + w(u,v)))
// therefore decreases towards, and stops exactly at, the same fixed point Dijkstra settles on.
// Updates are made IN PLACE (chaotic relaxation): a racing reader sees either the old or the new
// 8-byte value, both valid path sums.  Convergence is declared only by a sweep (one kernel
// launch) in which no store happened; inside such a launch every read returns the launch-start
// state (kernel boundaries write back L2 and invalidate L1), so the state is a true fixed point.
//
// Layouts
//   multi-source: dist[batch][node][64] fp64 -- the 64 sources of a batch are the fast axis, so
//     one wave relaxes one node for 64 sources with 512-byte coalesced row reads while the CSR
//     row (column ids, weights) is wave-uniform and comes through the scalar cache;
//   single-source: dist[node] fp64, 16 lanes share one node's adjacency row.
#include "geo_common.h"
#include "sssp_device.h"

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <vector>

namespace {

constexpr int WAVES_PER_BLOCK = 4;
constexpr int SWEEP_GROUP = 4;    // sweeps enqueued between host convergence checks

__device__ __forceinline__ double inf64() { return __longlong_as_double(0x7ff0000000000000LL); }

// lane J of every 16-lane row, broadcast to the whole row (DPP row_newbcast)
template <int J>
__device__ __forceinline__ int row_bcast(int x) {
    return __builtin_amdgcn_update_dpp(0, x, 0x150 + J, 0xf, 0xf, false);
}

// One step of the 16-source relaxation: edge J of the 16 edges held one per lane in (idx, wbits).
// Unconditional (lanes past the row end carry the node itself with weight +inf), so the 16 gathers of a
// chunk are independent loads in flight together; 32-bit element offsets keep the address math to one op.
template <int J>
__device__ __forceinline__ void relax_edge16(const double *__restrict__ D, unsigned s, int idx, int wbits, double &best) {
    const unsigned u = (unsigned)row_bcast<J>(idx);
    const float w = __int_as_float(row_bcast<J>(wbits));
    // 32-bit BYTE offset from the (wave-uniform) batch base: scalar base + one VGPR offset, no 64-bit VALU math
    const unsigned off = (u << 7) | (s << 3);
    const double du = *reinterpret_cast<const double *>(reinterpret_cast<const char *>(D) + off);
    best = fmin(best, du + (double)w);
}

// ------------------------------------------------------------------------------------ multi-source
// dist[batch][node][sb]: `sb` sources per batch are the fast axis (sb in {64, 16}).
__global__ __launch_bounds__(256) void init_multi_kernel(double *__restrict__ dist, const int32_t *__restrict__ src,
                                                        int32_t n, int32_t nb, int32_t sb) {
    const int64_t total = (int64_t)nb * n * sb;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int s = (int)(i % sb);
        const int64_t r = i / sb;
        const int32_t v = (int32_t)(r % n);
        const int32_t b = (int32_t)(r / n);
        dist[i] = (src[b * sb + s] == v) ? 0.0 : inf64();
    }
}

// One sweep over every (batch, node).  flags: ring of 3 slots x nb ints; this sweep reads slot
// `prev`, sets slot `cur` where something changed and clears slot `next`.
//
// A wave relaxes 64/SBT nodes for the SBT sources of one batch (lane = node slot * SBT + source): each
// neighbour row is one SBT*8-byte contiguous read.  Block order is batch-group major: blocks
// [g*8*P, (g+1)*8*P) cover batches 8g .. 8g+7 with batch = 8g + (bid % 8) (fewer than 8 batches: bid % nb).  Blocks are dealt round-robin
// over the 8 XCDs and dispatched in order, so at any moment one XCD's L2 works on ONE batch whose
// n*SBT*8 bytes of distances it can hold (4 MiB L2): every row is re-read ~deg times per sweep from L2
// instead of the Infinity Cache (measured 17-19 TB/s vs 9 TB/s of gathered bytes).  Placement is a
// speed matter only; results do not depend on it.
template <int SBT, bool WEIGHTED>
__global__ __launch_bounds__(256) void sweep_multi_kernel(const int32_t *__restrict__ indptr,
                                                         const int32_t *__restrict__ indices,
                                                         const float *__restrict__ weights, int32_t n, int32_t nb,
                                                         int32_t blocks_per_batch, int32_t gs, double *dist,
                                                         int32_t *flags, int prev, int cur, int next, int first, int cs) {
    constexpr int NPW = 64 / SBT;            // nodes per wave
    constexpr int NPB = NPW * WAVES_PER_BLOCK;
    const int bid = blockIdx.x;
    if (bid == 0)
        for (int i = threadIdx.x; i < nb; i += blockDim.x) flags[next * cs + i] = 0;
    const int group = bid / (gs * blocks_per_batch);          // gs = min(nb, 8) batches share a block group
    const int b = group * gs + (bid % gs);
    const int xb = (bid % (gs * blocks_per_batch)) / gs;
    if (b >= nb) return;
    if (!first && flags[prev * cs + b] == 0) return;   // this batch already reached its fixed point

    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int slot = lane / SBT, s = lane % SBT;
    double *D = dist + (size_t)b * n * SBT;
    bool any = false;
    for (int32_t v0 = xb * NPB; v0 < n; v0 += blocks_per_batch * NPB) {
        int32_t v = v0 + wave * NPW + slot;
        if (SBT == 64) v = __builtin_amdgcn_readfirstlane(v);   // whole wave on one node: CSR row via scalar loads
        if (v >= n) continue;
        const int32_t e0 = indptr[v], e1 = indptr[v + 1];
        const double curv = D[(size_t)v * SBT + s];
        double best = curv;
        if (SBT == 16) {
            // the 16 lanes of a node slot fetch 16 consecutive CSR entries with ONE coalesced load each for
            // columns and weights, then share them lane by lane through DPP: one gather per edge remains
            for (int32_t e = e0; e < e1; e += 16) {
                const int32_t cnt = e1 - e;                        // uniform inside the slot
                const int idx = (s < cnt) ? indices[e + s] : v;    // padding: the node itself ...
                const int wb = (s < cnt) ? (WEIGHTED ? __float_as_int(weights[e + s]) : 0x3f800000)
                                         : 0x7f800000;             // ... at distance +inf (no effect)
                const unsigned us = (unsigned)s;
                relax_edge16<0>(D, us, idx, wb, best); relax_edge16<1>(D, us, idx, wb, best);
                relax_edge16<2>(D, us, idx, wb, best); relax_edge16<3>(D, us, idx, wb, best);
                relax_edge16<4>(D, us, idx, wb, best); relax_edge16<5>(D, us, idx, wb, best);
                relax_edge16<6>(D, us, idx, wb, best); relax_edge16<7>(D, us, idx, wb, best);
                relax_edge16<8>(D, us, idx, wb, best); relax_edge16<9>(D, us, idx, wb, best);
                relax_edge16<10>(D, us, idx, wb, best); relax_edge16<11>(D, us, idx, wb, best);
                relax_edge16<12>(D, us, idx, wb, best); relax_edge16<13>(D, us, idx, wb, best);
                relax_edge16<14>(D, us, idx, wb, best); relax_edge16<15>(D, us, idx, wb, best);
            }
        } else {
            // 16 neighbour rows in flight per wave: the sweep is bound by gather latency x occupancy, not by
            // bytes, so memory-level parallelism per wave is what moves it
            int32_t e = e0;
            for (; e + 16 <= e1; e += 16) {
                double dd[16];
#pragma unroll
                for (int j = 0; j < 16; ++j) dd[j] = D[(size_t)indices[e + j] * SBT + s];
#pragma unroll
                for (int j = 0; j < 16; ++j) best = fmin(best, dd[j] + (WEIGHTED ? (double)weights[e + j] : 1.0));
            }
            for (; e + 4 <= e1; e += 4) {
                const int32_t u0 = indices[e], u1 = indices[e + 1], u2 = indices[e + 2], u3 = indices[e + 3];
                const double d0 = D[(size_t)u0 * SBT + s], d1 = D[(size_t)u1 * SBT + s];
                const double d2 = D[(size_t)u2 * SBT + s], d3 = D[(size_t)u3 * SBT + s];
                const double w0 = WEIGHTED ? (double)weights[e] : 1.0, w1 = WEIGHTED ? (double)weights[e + 1] : 1.0;
                const double w2 = WEIGHTED ? (double)weights[e + 2] : 1.0, w3 = WEIGHTED ? (double)weights[e + 3] : 1.0;
                best = fmin(best, fmin(fmin(d0 + w0, d1 + w1), fmin(d2 + w2, d3 + w3)));
            }
            for (; e < e1; ++e) {
                const double w = WEIGHTED ? (double)weights[e] : 1.0;
                best = fmin(best, D[(size_t)indices[e] * SBT + s] + w);
            }
        }
        if (best < curv) {
            D[(size_t)v * SBT + s] = best;
            any = true;
        }
    }
    if (__any(any) && lane == 0) flags[cur * cs + b] = 1;
}

// ---- 16-source batches, work item = one 16-edge chunk of one node's row -------------------------
// With 16 sources per batch a batch's distances are n*128 bytes: at N = 60 000 that is 7.7 MB, which one
// XCD's L2 serves at ~17 TB/s of gathered bytes, against ~9 TB/s from the Infinity Cache for 64-source
// batches (30.7 MB).  Rows are cut into 16-edge chunks so that the four 16-lane slots of a wave always have
// the same trip count (node degrees range from k to several hundred); a chunk's minimum is folded into
// the node's distance with a 64-bit atomicMin on the bit pattern (non-negative doubles order like
// unsigned integers), which keeps the in-place, only-decreasing update rule of the plain sweep.
__global__ __launch_bounds__(256) void chunk_count_kernel(const int32_t *__restrict__ indptr, int32_t n,
                                                         int32_t *__restrict__ cnt) {
    for (int32_t v = blockIdx.x * blockDim.x + threadIdx.x; v < n; v += gridDim.x * blockDim.x)
        cnt[v] = (indptr[v + 1] - indptr[v] + 15) >> 4;
}

__global__ __launch_bounds__(256) void chunk_fill_kernel(const int32_t *__restrict__ indptr, int32_t n,
                                                        const int32_t *__restrict__ chunk_off,
                                                        int32_t *__restrict__ chunk_node,
                                                        int32_t *__restrict__ chunk_start) {
    for (int32_t v = blockIdx.x * blockDim.x + threadIdx.x; v < n; v += gridDim.x * blockDim.x) {
        const int32_t c0 = chunk_off[v], c1 = chunk_off[v + 1];
        for (int32_t c = c0; c < c1; ++c) {
            chunk_node[c] = v;
            chunk_start[c] = indptr[v] + ((c - c0) << 4);
        }
    }
}

// first sweep's need-map: the neighbours of every source (one byte per (batch, node))
__global__ __launch_bounds__(256) void source_need_kernel(const int32_t *__restrict__ src, int32_t n_sources, int32_t n,
                                                         int32_t sb, int32_t words, const int32_t *__restrict__ indptr,
                                                         const int32_t *__restrict__ indices, uint32_t *__restrict__ bits) {
    uint8_t *map = reinterpret_cast<uint8_t *>(bits);
    const int sub = threadIdx.x & 15;
    for (int32_t i = (blockIdx.x * blockDim.x + threadIdx.x) >> 4; i < n_sources; i += (gridDim.x * blockDim.x) >> 4) {
        const int32_t v = src[i];
        if (v < 0 || v >= n) continue;
        uint8_t *m = map + (size_t)(i / sb) * words * 4;
        for (int32_t e = indptr[v] + sub; e < indptr[v + 1]; e += 16) m[indices[e]] = 1;
    }
}

template <bool WEIGHTED>
__global__ __launch_bounds__(256) void sweep_chunk16_kernel(const int32_t *__restrict__ indptr,
                                                           const int32_t *__restrict__ indices,
                                                           const float *__restrict__ weights, int32_t n, int32_t nb,
                                                           const int32_t *__restrict__ chunk_node,
                                                           const int32_t *__restrict__ chunk_start, int32_t n_chunks,
                                                           int32_t blocks_per_batch, int32_t gs, double *dist,
                                                           int32_t *flags, int32_t *counts,
                                                           const uint32_t *__restrict__ bits_prev,
                                                           uint32_t *__restrict__ bits_cur, int32_t words, int prev,
                                                           int cur, int next, int first, int act_mode,
                                                           int sparse_div, int map_div, int c_pprev, int c_prev,
                                                           int c_cur, int c_clear, int cs) {
    const int bid = blockIdx.x;
    if (bid == 0)
        for (int i = threadIdx.x; i < nb; i += blockDim.x) flags[next * cs + i] = 0;
    const int group = bid / (gs * blocks_per_batch);
    const int b = group * gs + (bid % gs);
    const int xb = (bid % (gs * blocks_per_batch)) / gs;
    if (b >= nb) return;
    if (!first && flags[prev * cs + b] == 0) return;                     // fixed point reached for this batch
    // visiting only flagged rows pays when few rows moved (first sweeps, last sweeps): `counts` holds a
    // 1-in-16 block sample of the number of improved (node, source) pairs of the previous sweep
    // counts[.] >= 0: sampled count, map written;  < 0: -(sampled count), map NOT written by that sweep
    const int32_t cp = first ? 0 : counts[c_prev * cs + b];
    const int32_t cpp = first ? 0 : counts[c_pprev * cs + b];
    const int32_t est_prev = (cp >= 0 ? cp : -cp) * 16;
    // expected improvements of THIS sweep: last sweep's count times its growth over the sweep before (a front
    // that is still spreading multiplies by up to the mean degree per sweep; a converging solve shrinks)
    const int32_t est_pprev = (cpp >= 0 ? cpp : -cpp) * 16;
    float ratio = est_pprev > 0 ? (float)est_prev / (float)est_pprev : 32.0f;
    ratio = ratio < 1.0f ? 1.0f : (ratio > 64.0f ? 64.0f : ratio);
    const float expect = (float)est_prev * ratio;
    const bool sparse_sweep = act_mode && (first || (cp >= 0 && expect * (float)sparse_div < (float)n * 16.0f));
    // flagging costs a scattered byte store per edge of every improved row: keep it only while few rows move
    const bool write_map = sparse_sweep || (act_mode && expect * (float)map_div < (float)n * 16.0f);
    if (bid == 0)
        for (int i = threadIdx.x; i < nb; i += blockDim.x) counts[c_clear * cs + i] = 0;

    const int lane = threadIdx.x & 63;
    const int slot_in_block = threadIdx.x >> 4;              // 16 slots of 16 lanes per block
    const unsigned s = lane & 15;
    double *D = dist + (size_t)b * n * 16;
    unsigned long long *Dbits = reinterpret_cast<unsigned long long *>(D);
    // need-map: one byte per (batch, node), "a neighbour's distance was lowered (for any of this batch's 16
    // sources) during the previous sweep".  A row that improves sets the byte of all its neighbours with plain
    // idempotent stores (no atomics); every change is thus followed by an evaluation of its dependents in the
    // next sweep, so the fixed point is the same, and sweeps in which few rows move (the first ones, the last
    // ones, and every sweep of a long-geodesic graph) only touch the flagged rows.  `bits_prev` was filled by
    // the previous sweep (by source_need_kernel before the first).
    const uint8_t *bp = reinterpret_cast<const uint8_t *>(bits_prev) + (size_t)b * words * 4;
    uint8_t *bc = reinterpret_cast<uint8_t *>(bits_cur) + (size_t)b * words * 4;
    const int slot_in_wave = lane >> 4;
    int32_t n_better = 0;
#define GEO_RELAX16_ALL()                                                                                   \
    relax_edge16<0>(D, s, idx, wb, best); relax_edge16<1>(D, s, idx, wb, best);                             \
    relax_edge16<2>(D, s, idx, wb, best); relax_edge16<3>(D, s, idx, wb, best);                             \
    relax_edge16<4>(D, s, idx, wb, best); relax_edge16<5>(D, s, idx, wb, best);                             \
    relax_edge16<6>(D, s, idx, wb, best); relax_edge16<7>(D, s, idx, wb, best);                             \
    relax_edge16<8>(D, s, idx, wb, best); relax_edge16<9>(D, s, idx, wb, best);                             \
    relax_edge16<10>(D, s, idx, wb, best); relax_edge16<11>(D, s, idx, wb, best);                           \
    relax_edge16<12>(D, s, idx, wb, best); relax_edge16<13>(D, s, idx, wb, best);                           \
    relax_edge16<14>(D, s, idx, wb, best); relax_edge16<15>(D, s, idx, wb, best)
    if (sparse_sweep) {
        // few rows moved last sweep: a 16-lane slot takes a whole row, and only flagged rows are evaluated
        for (int32_t v = xb * 16 + slot_in_block; v < n; v += blocks_per_batch * 16) {
            if (bp[v] == 0) continue;                                            // slot-uniform
            const int32_t e0 = indptr[v], e1 = indptr[v + 1];
            const double curv = D[(unsigned)v * 16u + s];
            double best = curv;
            for (int32_t e = e0; e < e1; e += 16) {
                const int32_t cnt = e1 - e;
                const int idx = ((int)s < cnt) ? indices[e + s] : v;
                const int wb = ((int)s < cnt) ? (WEIGHTED ? __float_as_int(weights[e + s]) : 0x3f800000) : 0x7f800000;
                GEO_RELAX16_ALL();
            }
            const bool better = best < curv;
            if (better) {
                atomicMin(&Dbits[(unsigned)v * 16u + s], (unsigned long long)__double_as_longlong(best));
                ++n_better;
            }
            if ((__ballot(better) >> (slot_in_wave * 16)) & 0xffffull)           // the row moved: flag its neighbours
                for (int32_t e = e0 + (int)s; e < e1; e += 16) bc[indices[e]] = 1;
        }
    } else {
        // dense sweep: straight-line body, no wave-level votes between a chunk's loads and the next chunk's
        for (int32_t c = xb * 16 + slot_in_block; c < n_chunks; c += blocks_per_batch * 16) {
            const int32_t v = chunk_node[c], e = chunk_start[c];
            const int32_t cnt = indptr[v + 1] - e;           // >= 1; more than 16 means further chunks follow
            const int idx = ((int)s < cnt) ? indices[e + s] : v;
            const int wb = ((int)s < cnt) ? (WEIGHTED ? __float_as_int(weights[e + s]) : 0x3f800000) : 0x7f800000;
            const double curv = D[(unsigned)v * 16u + s];
            double best = curv;
            GEO_RELAX16_ALL();
            const bool better = best < curv;
            if (better) {
                atomicMin(&Dbits[(unsigned)v * 16u + s], (unsigned long long)__double_as_longlong(best));
                ++n_better;
            }
            if (write_map && ((__ballot(better) >> (slot_in_wave * 16)) & 0xffffull))   // flag ALL neighbours of v
                for (int32_t e2 = indptr[v] + (int)s; e2 < indptr[v + 1]; e2 += 16) bc[indices[e2]] = 1;
        }
    }
#undef GEO_RELAX16_ALL
    // n_better counts improved (node, source) pairs of this thread; /16 ~ improved slots
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) n_better += __shfl_xor(n_better, off, 64);
    if (lane == 0 && n_better > 0) {
        flags[cur * cs + b] = 1;                                          // exact: something changed
        if ((xb & 15) == 0) atomicAdd(&counts[c_cur * cs + b], write_map ? n_better : -n_better);   // sampled: how much
    }
}

// ---- exact 32-bit fixed point: 32 sources per 128-byte row ----------------------------------------------------
// scipy accumulates path sums in fp64 (geo_shortest_paths.py:44-50).  Every float32 weight is an integer multiple of
// 2^(e_min - 23), e_min the exponent of the smallest positive weight; while a path sum stays below 2^53 of those units
// every fp64 addition along it is EXACT, so the fp64 distances are exactly the integer sums of the weights in units.
// When the weights span few binades (24 + e_max - e_min <= 28 bits) and no distance reaches 2^32 units, the whole
// solve runs on uint32 -- identical results, half the bytes: 32 sources share one 128-byte row, so a relaxation
// gathers one cache line per (edge, 32 sources) instead of one per (edge, 16 sources).  Sums saturate at U_OVF; if one
// survives to the fixed point (or the weights do not qualify) the fp64 kernels below solve the call instead.
constexpr uint32_t U_INF = 0xffffffffu, U_OVF = 0xfffffffeu;

__global__ __launch_bounds__(256) void weight_range_kernel(const float *__restrict__ w, int64_t nnz, uint32_t *__restrict__ out) {
    uint32_t lo = 0xffffffffu, hi = 0u, bad = 0u;              // bit patterns of non-negative floats order like the floats
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nnz; i += (int64_t)gridDim.x * blockDim.x) {
        const float x = w[i];
        const uint32_t b = __float_as_uint(x);
        if (!(x >= 0.0f) || b >= 0x7f800000u) bad = 1u;
        else if (b != 0u) { lo = b < lo ? b : lo; hi = b > hi ? b : hi; }
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        const uint32_t l2 = __shfl_xor(lo, off, 64), h2 = __shfl_xor(hi, off, 64);
        lo = l2 < lo ? l2 : lo; hi = h2 > hi ? h2 : hi; bad |= __shfl_xor(bad, off, 64);
    }
    // one set of atomics per BLOCK (4 096 waves hitting three addresses serialised at the L2: 98 us for 7.6 MB)
    __shared__ uint32_t s_lo[4], s_hi[4], s_bad[4];
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { s_lo[wave] = lo; s_hi[wave] = hi; s_bad[wave] = bad; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < 4; ++w) { lo = s_lo[w] < lo ? s_lo[w] : lo; hi = s_hi[w] > hi ? s_hi[w] : hi; bad |= s_bad[w]; }
        atomicMin(&out[0], lo);
        atomicMax(&out[1], hi);
        if (bad) atomicOr(&out[2], 1u);
    }
}

__global__ __launch_bounds__(256) void weight_units_kernel(const float *__restrict__ w, int64_t nnz, int shift,
                                                          uint32_t *__restrict__ wu) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nnz; i += (int64_t)gridDim.x * blockDim.x)
        wu[i] = w ? (uint32_t)ldexp((double)w[i], shift) : 1u;     // exact: an integer below 2^29 by the host's check
}

__global__ __launch_bounds__(256) void init_multi32_kernel(uint32_t *__restrict__ dist, const int32_t *__restrict__ src,
                                                          int32_t n, int32_t nb) {
    const int64_t total = (int64_t)nb * n * 32;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int s = (int)(i & 31);
        const int64_t r = i >> 5;
        dist[i] = (src[(r / n) * 32 + s] == (int32_t)(r % n)) ? 0u : U_INF;
    }
}

template <int J>
__device__ __forceinline__ void relax_edge32u(const uint32_t *__restrict__ D, unsigned sidx, int idx, int wu, uint32_t &best) {
    const unsigned u = (unsigned)row_bcast<J>(idx);
    const uint32_t w = (uint32_t)row_bcast<J>(wu);
    const uint32_t du = *reinterpret_cast<const uint32_t *>(reinterpret_cast<const char *>(D) + ((u << 7) | (sidx << 2)));
    const uint32_t sum = du + w;
    const uint32_t cand = du >= U_OVF ? du : ((sum < du || sum >= U_OVF) ? U_OVF : sum);    // saturating, inf stays inf
    best = cand < best ? cand : best;
}

// rows sorted by decreasing number of 16-entry chunks: counting sort in ONE workgroup (histogram and cursors in LDS; the
// lanes of a wave that fall into the same bin -- most rows have 2 or 3 chunks -- share one LDS atomic)
__global__ __launch_bounds__(1024) void row_order_kernel(const int32_t *__restrict__ chunk_cnt, int32_t n,
                                                        int32_t *__restrict__ row_order) {
    __shared__ int32_t hist[256];
    const int lane = threadIdx.x & 63;
    const unsigned long long lane_lt = (1ull << lane) - 1ull;
    if (threadIdx.x < 256) hist[threadIdx.x] = 0;
    __syncthreads();
    for (int pass = 0; pass < 2; ++pass) {
        for (int32_t v0 = 0; v0 < n; v0 += 1024) {
            const int32_t v = v0 + threadIdx.x;
            const bool live = v < n;
            int bin = -1;
            if (live) { const int32_t c = chunk_cnt[v]; bin = 255 - (c < 255 ? c : 255); }      // bin 0 = longest rows
            unsigned long long todo = __ballot(live);
            while (todo) {                                                   // one LDS atomic per distinct bin of the wave
                const int leader = __ffsll((long long)todo) - 1;
                const int b = __builtin_amdgcn_readlane(bin, leader);
                const unsigned long long same = __ballot(live && bin == b);
                int base = 0;
                if (lane == leader) base = atomicAdd(&hist[b], __popcll(same));
                base = __builtin_amdgcn_readlane(base, leader);
                if (pass == 1 && live && bin == b) row_order[base + __popcll(same & lane_lt)] = v;
                todo &= ~same;
            }
        }
        __syncthreads();
        if (pass == 0) {
            if (threadIdx.x == 0) {                                           // counts -> exclusive offsets (the pass-1 cursors)
                int32_t run = 0;
                for (int b = 0; b < 256; ++b) { const int32_t c = hist[b]; hist[b] = run; run += c; }
            }
            __syncthreads();
        }
    }
}

// Same roles as in sweep_chunk16_kernel (dense body, flagged-row body, need-map, sampled improvement counts); a slot is
// 32 lanes = 32 sources and takes whole rows, its two 16-lane DPP rows each hold a copy of the current 16 edges.
__global__ __launch_bounds__(256) void sweep_chunk32u_kernel(const int32_t *__restrict__ indptr,
                                                            const int32_t *__restrict__ indices,
                                                            const uint32_t *__restrict__ wunits, int32_t n, int32_t nb,
                                                            const int32_t *__restrict__ row_order,
                                                            int32_t blocks_per_batch, int32_t gs, uint32_t *dist,
                                                            int32_t *flags, int32_t *counts,
                                                            const uint32_t *__restrict__ bits_prev,
                                                            uint32_t *__restrict__ bits_cur, int32_t words, int prev,
                                                            int cur, int next, int first, int act_mode,
                                                            int sparse_div, int map_div, int c_pprev, int c_prev,
                                                            int c_cur, int c_clear, int cs) {
    const int bid = blockIdx.x;
    if (bid == 0)
        for (int i = threadIdx.x; i < nb; i += blockDim.x) flags[next * cs + i] = 0;
    const int group = bid / (gs * blocks_per_batch);
    const int b = group * gs + (bid % gs);
    const int xb = (bid % (gs * blocks_per_batch)) / gs;
    if (b >= nb) return;
    if (!first && flags[prev * cs + b] == 0) return;                     // fixed point reached for this batch
    const int32_t cp = first ? 0 : counts[c_prev * cs + b];
    const int32_t cpp = first ? 0 : counts[c_pprev * cs + b];
    const int32_t est_prev = (cp >= 0 ? cp : -cp) * 16;
    const int32_t est_pprev = (cpp >= 0 ? cpp : -cpp) * 16;
    float ratio = est_pprev > 0 ? (float)est_prev / (float)est_pprev : 32.0f;
    ratio = ratio < 1.0f ? 1.0f : (ratio > 64.0f ? 64.0f : ratio);
    const float expect = (float)est_prev * ratio;
    const bool sparse_sweep = act_mode && (first || (cp >= 0 && expect * (float)sparse_div < (float)n * 32.0f));
    const bool write_map = sparse_sweep || (act_mode && expect * (float)map_div < (float)n * 32.0f);
    if (bid == 0)
        for (int i = threadIdx.x; i < nb; i += blockDim.x) counts[c_clear * cs + i] = 0;

    const int lane = threadIdx.x & 63;
    const int slot_in_block = threadIdx.x >> 5;              // 8 slots of 32 lanes per block
    const int s16 = lane & 15;                               // position in the DPP row = edge of the chunk this lane loads
    const unsigned sidx = lane & 31;                         // source of this lane
    const int slot_in_wave = lane >> 5;
    uint32_t *D = dist + (size_t)b * n * 32;
    const uint8_t *bp = reinterpret_cast<const uint8_t *>(bits_prev) + (size_t)b * words * 4;
    uint8_t *bc = reinterpret_cast<uint8_t *>(bits_cur) + (size_t)b * words * 4;
    int32_t n_better = 0;
#define GEO_RELAX32_ALL()                                                                                       \
    relax_edge32u<0>(D, sidx, idx, wb, best); relax_edge32u<1>(D, sidx, idx, wb, best);                         \
    relax_edge32u<2>(D, sidx, idx, wb, best); relax_edge32u<3>(D, sidx, idx, wb, best);                         \
    relax_edge32u<4>(D, sidx, idx, wb, best); relax_edge32u<5>(D, sidx, idx, wb, best);                         \
    relax_edge32u<6>(D, sidx, idx, wb, best); relax_edge32u<7>(D, sidx, idx, wb, best);                         \
    relax_edge32u<8>(D, sidx, idx, wb, best); relax_edge32u<9>(D, sidx, idx, wb, best);                         \
    relax_edge32u<10>(D, sidx, idx, wb, best); relax_edge32u<11>(D, sidx, idx, wb, best);                       \
    relax_edge32u<12>(D, sidx, idx, wb, best); relax_edge32u<13>(D, sidx, idx, wb, best);                       \
    relax_edge32u<14>(D, sidx, idx, wb, best); relax_edge32u<15>(D, sidx, idx, wb, best)
    if (sparse_sweep) {
        for (int32_t v = xb * 8 + slot_in_block; v < n; v += blocks_per_batch * 8) {
            if (bp[v] == 0) continue;                                            // slot-uniform
            const int32_t e0 = indptr[v], e1 = indptr[v + 1];
            const uint32_t curv = D[(unsigned)v * 32u + sidx];
            uint32_t best = curv;
            for (int32_t e = e0; e < e1; e += 16) {
                const int32_t cnt = e1 - e;
                const int idx = (s16 < cnt) ? indices[e + s16] : v;              // padding: the node itself, weight 0
                const int wb = (s16 < cnt) ? (int)wunits[e + s16] : 0;
                GEO_RELAX32_ALL();
            }
            const bool better = best < curv;
            if (better) {
                D[(unsigned)v * 32u + sidx] = best;                              // the only writer of this (row, source)
                ++n_better;
            }
            if ((__ballot(better) >> (slot_in_wave * 32)) & 0xffffffffull)       // the row moved: flag its neighbours
                for (int32_t e = e0 + (int)sidx; e < e1; e += 32) bc[indices[e]] = 1;
        }
    } else {
        // dense sweep: a slot takes a whole row, rows in order of decreasing length (row_order: the two slots of a wave
        // and the waves of a block run rows of about the same length).  One writer per (row, source) and sweep, so the
        // improved distances are PLAIN stores: no atomics, no re-read of the row's own distance per chunk.
        for (int32_t r = xb * 8 + slot_in_block; r < n; r += blocks_per_batch * 8) {
            const int32_t v = row_order[r];
            const int32_t e0 = indptr[v], e1 = indptr[v + 1];
            const uint32_t curv = D[(unsigned)v * 32u + sidx];
            uint32_t best = curv;
            for (int32_t e = e0; e < e1; e += 16) {
                const int32_t cnt = e1 - e;
                const int idx = (s16 < cnt) ? indices[e + s16] : v;
                const int wb = (s16 < cnt) ? (int)wunits[e + s16] : 0;
                GEO_RELAX32_ALL();
            }
            const bool better = best < curv;
            if (better) {
                D[(unsigned)v * 32u + sidx] = best;
                ++n_better;
            }
            if (write_map && ((__ballot(better) >> (slot_in_wave * 32)) & 0xffffffffull))   // flag ALL neighbours of v
                for (int32_t e2 = e0 + (int)sidx; e2 < e1; e2 += 32) bc[indices[e2]] = 1;
        }
    }
#undef GEO_RELAX32_ALL
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) n_better += __shfl_xor(n_better, off, 64);
    if (lane == 0 && n_better > 0) {
        flags[cur * cs + b] = 1;                                          // exact: something changed
        if ((xb & 15) == 0) atomicAdd(&counts[c_cur * cs + b], write_map ? n_better : -n_better);   // sampled: how much
    }
}

// ---- near-far push solve: long geodesics -------------------------------------------------------------------------
// Label-correcting PULL sweeps re-read a whole row whenever any of its ~25 neighbours moved for any source of the batch;
// on a graph with ~150-hop geodesics and edge weights spread over a factor 16 (pull-back lengths on a manifold-like
// latent cloud) that is ~27 row evaluations per row and ~7 improvements per (node, source) -- measured, DESIGN.md.
// This solve is the GPU form of delta-stepping (north_star names it; scipy's Dijkstra settles each node once,
// geo_shortest_paths.py:36-49): work is driven from the rows that CHANGED (push), and a changed row relaxes its edges
// only once its distance is below the batch's threshold theta (near); rows beyond wait in a far pile until the near
// work has drained and theta advances by delta.  Measured in simulation on the swiss-roll bench graph: 4 pushed rows
// per row instead of 27 evaluated rows, 1.9 improvements per pair instead of 6.9.
//
// Exactness: every stored value is the fp64 left-to-right sum along a real path (cand = D[u] + w, one rounding per hop),
// stores only lower a value (64-bit atomicMin on the bit pattern: non-negative doubles order like unsigned integers),
// and every lowered row is pushed again before the solve ends (it sits in the near list or the far pile until then).
// The state at termination -- no near work, no far pile, in any batch -- is therefore the same unique fixed point
// min_u fl(D[u] + w) that Dijkstra computes; the order of pushes and the choice of delta only change the work.
//
// State per batch b (16 sources, dist[b][node][16] as in the chunked solve), all in the caller's workspace:
//   near[parity][b][.]  rows to push this sweep / next sweep, near_cnt ring of 3 (consume, build, clear)
//   far[ring of 3][b][.] the far pile; a release sweep (near empty) compacts it into the next ring slot
//   near_bits[parity][b] one bit per row: already queued for the sweep of that parity; far_bits[b]: sits in the pile.
//     Bitmaps, not one word per row: 7.5 KB per batch stay in the L2, where a returning atomic costs a few hundred ns -- the
//     same dedupe through one int per (batch, row) was a random access into 7.7 MB per enqueue and half of the solve's time.
//   theta[parity][b], far_slot[parity][b]: written by the batch's first block for the next sweep.
// Every block of a batch derives the same decision (push / release / idle) from the counts the previous launch left.
struct PushState {
    int32_t *near_lists, *near_cnt, *far_lists, *far_cnt, *far_slot, *active;
    uint32_t *near_bits, *far_bits;                          // [2][nb][words], [nb][words]
    int32_t words;
    double *theta;
    const int32_t *chunk_off, *chunk_node, *chunk_start;     // 16-entry chunks of the CSR rows (the near list holds chunk ids)
    int32_t cs;          // row stride of the per-batch rings (multiple of 32 ints: a 128-byte line of their own)
    int32_t cap;         // entries per near list (= chunks of the graph)
};

__device__ __forceinline__ double slot_min16(double x) {
#pragma unroll
    for (int off = 8; off >= 1; off >>= 1) x = fmin(x, __shfl_xor(x, off, 16));
    return x;
}

// Atomics of the push solve (relaxed, device scope; on gfx950 an atomic RMW is the same instruction at workgroup and agent
// scope, so pinning a batch to one XCD buys L2 locality for its rows, stamps and counters, not a cheaper instruction).
template <typename T>
__device__ __forceinline__ T push_atomic_min(T *p, T v) { return __hip_atomic_fetch_min(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ int32_t push_atomic_exch(int32_t *p, int32_t v) { return __hip_atomic_exchange(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ int32_t push_atomic_add(int32_t *p, int32_t v) { return __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// exclusive prefix sum over the 16 lanes of a slot (DPP-free: 4 shuffle steps), and the slot's total
__device__ __forceinline__ int slot_scan16(int x, int lane16, int *total) {
    int incl = x;
#pragma unroll
    for (int off = 1; off < 16; off <<= 1) {
        const int y = __shfl_up(incl, off, 16);
        if (lane16 >= off) incl += y;
    }
    *total = __shfl(incl, 15, 16);
    return incl - x;
}

// One sweep of ONE batch by the blocks assigned to it (xb = this block's position among them, blocks_per_batch of them).
// Work item of a push sweep = one 16-entry CHUNK of a changed row (hub rows of several hundred entries spread over slots
// instead of serialising one), named by its chunk id in the near list.  Returns false once the batch has nothing left.
// SLOTS = 16-lane slots per block (blockDim.x / 16).  Every block of the batch derives the same decision (push / release /
// idle) from the counts the previous sweep left.
template <bool WEIGHTED, int SLOTS>
__device__ __forceinline__ bool push_batch_sweep(const int32_t *__restrict__ indptr, const int32_t *__restrict__ indices,
                                                 const float *__restrict__ weights, int32_t n, int32_t nb, int32_t b, int32_t xb,
                                                 int32_t blocks_per_batch, double *dist, const PushState &st, double delta,
                                                 int32_t sweep) {
    const int par = sweep & 1, c_in = sweep % 3, c_out = (sweep + 1) % 3, c_clr = (sweep + 2) % 3;
    const int32_t nn = st.near_cnt[c_in * st.cs + b];
    const int32_t f_now = st.far_slot[par * st.cs + b];
    const int32_t nf = st.far_cnt[f_now * st.cs + b];
    const double th = st.theta[par * st.cs + b];
    const bool release = nn == 0 && nf > 0;
    const int32_t f_out = release ? (f_now + 1) % 3 : f_now;
    const double th_next = release ? th + delta : th;
    if (xb == 0 && threadIdx.x == 0) {                      // this batch's state for the next sweep
        st.theta[(par ^ 1) * st.cs + b] = th_next;
        st.far_slot[(par ^ 1) * st.cs + b] = f_out;
        st.near_cnt[c_clr * st.cs + b] = 0;
        if (release) st.far_cnt[((f_now + 2) % 3) * st.cs + b] = 0;
    }
    if (nn == 0 && nf == 0) return false;                    // this batch has reached its fixed point

    const int lane = threadIdx.x & 63;
    const unsigned s = lane & 15;
    const int sh = (lane >> 4) * 16;
    const int slot_in_block = threadIdx.x >> 4;
    const int32_t stride = blocks_per_batch * SLOTS;
    double *D = dist + (size_t)b * n * 16;
    const int32_t *near_in = st.near_lists + ((size_t)par * nb + b) * st.cap;
    int32_t *near_out = st.near_lists + ((size_t)(par ^ 1) * nb + b) * st.cap;
    int32_t *far_now = st.far_lists + ((size_t)f_now * nb + b) * n;
    uint32_t *qbits = st.near_bits + ((size_t)(par ^ 1) * nb + b) * st.words;      // dedupe of the list being built
    uint32_t *fbits = st.far_bits + (size_t)b * st.words;
    {   // the other parity's bitmap served the list this sweep consumes: clear it for the sweep after next
        uint32_t *old = st.near_bits + ((size_t)par * nb + b) * st.words;
        for (int32_t i = xb * SLOTS * 16 + threadIdx.x; i < st.words; i += blocks_per_batch * SLOTS * 16) old[i] = 0u;
    }
    int32_t *ncount = &st.near_cnt[c_out * st.cs + b];

    // List appends are reserved per BLOCK and round: every slot posts how many near-list entries (chunks) and far-pile
    // entries (rows) it has, one thread adds the block's totals to the batch's counters, the slots write at their
    // offsets.  (One returning atomic per slot on a counter shared by the ~1000 slots of a batch serialises at the L2:
    // measured ~40 us per round, whatever the amount of work.)
    __shared__ int32_t sh_cnt[2][2][SLOTS];                    // [round parity][near, far][slot]
    __shared__ int32_t sh_base[2][2];
    const int32_t n_items = release ? (nf + 15) / 16 : nn;     // work items: 16 pile entries / one chunk per slot
    const int32_t rounds = (n_items + stride - 1) / stride;    // block-uniform
    int32_t *far_out = st.far_lists + ((size_t)f_out * nb + b) * n;       // release: the next ring slot; push: the current one
    int32_t *fcount = &st.far_cnt[f_out * st.cs + b];
    for (int32_t r = 0; r < rounds; ++r) {
        const int32_t item = r * stride + xb * SLOTS + slot_in_block;
        const bool active = item < n_items;
        int32_t v = -1;                                        // the row this lane speaks for
        bool to_near = false, to_far = false;
        if (active && release) {
            // theta advanced: rows of the pile that are now near move to the near list, the others to the next ring slot.
            // Lane j of a slot takes entry 16 * item + j of the pile; a row's 16 distances are read by the whole slot.
            const int32_t i0 = item * 16;
            const int32_t mine = i0 + (int32_t)s < nf ? far_now[i0 + s] : -1;
            double key_mine = inf64();
            for (int j = 0; j < 16; ++j) {
                const int32_t vj = __shfl(mine, j, 16);
                if (vj < 0) break;                                                  // slot-uniform
                const double key = slot_min16(D[(unsigned)vj * 16u + s]);
                if ((int)s == j) key_mine = key;
            }
            v = mine;
            if (mine >= 0) {
                if (key_mine < th_next) {
                    const uint32_t bit = 1u << (mine & 31);
                    __hip_atomic_fetch_and(&fbits[mine >> 5], ~bit, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    to_near = (__hip_atomic_fetch_or(&qbits[mine >> 5], bit, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & bit) == 0u;
                } else {
                    to_far = true;                                                  // stays flagged
                }
            }
        } else if (active) {
            // push: the slot takes one chunk (16 entries of a changed row u) and relaxes it for the batch's 16 sources
            const int32_t c = near_in[item];
            const int32_t u = st.chunk_node[c], e = st.chunk_start[c];
            const int32_t cnt = indptr[u + 1] - e;
            const double du = D[(unsigned)u * 16u + s];
            const int idx = ((int)s < cnt) ? indices[e + s] : u;         // padding: the row itself at +inf (no effect)
            const int wb = ((int)s < cnt) ? (WEIGHTED ? __float_as_int(weights[e + s]) : 0x3f800000) : 0x7f800000;
            const double mu = slot_min16(du);                           // smallest distance of this row over the sources
            double dv[16];
            unsigned long long old[16];
            unsigned off[16];
            float ww[16];
#define GEO_PUSH_LOAD(J)                                                                                    \
    off[J] = ((unsigned)row_bcast<J>(idx) << 7) | (s << 3); ww[J] = __int_as_float(row_bcast<J>(wb));       \
    dv[J] = *reinterpret_cast<const double *>(reinterpret_cast<const char *>(D) + off[J]);
            GEO_PUSH_LOAD(0) GEO_PUSH_LOAD(1) GEO_PUSH_LOAD(2) GEO_PUSH_LOAD(3) GEO_PUSH_LOAD(4) GEO_PUSH_LOAD(5)
            GEO_PUSH_LOAD(6) GEO_PUSH_LOAD(7) GEO_PUSH_LOAD(8) GEO_PUSH_LOAD(9) GEO_PUSH_LOAD(10) GEO_PUSH_LOAD(11)
            GEO_PUSH_LOAD(12) GEO_PUSH_LOAD(13) GEO_PUSH_LOAD(14) GEO_PUSH_LOAD(15)
#undef GEO_PUSH_LOAD
            // all minima of the chunk are issued before any result is looked at (16 atomics in flight, not 16 round trips)
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const double cand = du + (double)ww[j];
                old[j] = 0ull;
                if (cand < dv[j])
                    old[j] = push_atomic_min(reinterpret_cast<unsigned long long *>(reinterpret_cast<char *>(D) + off[j]),
                                             (unsigned long long)__double_as_longlong(cand));
            }
            unsigned movedmask = 0u;                                     // bit j: some source of the slot lowered entry j's far end
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const double cand = du + (double)ww[j];
                const bool better = old[j] > (unsigned long long)__double_as_longlong(cand);
                if ((__ballot(better) >> sh) & 0xffffull) movedmask |= 1u << j;
            }
            if ((movedmask >> s) & 1u) {                                 // lane s speaks for entry s of the chunk
                v = idx;
                const uint32_t bit = 1u << (idx & 31);
                uint32_t *word = (mu + (double)__int_as_float(wb) < th ? qbits : fbits) + (idx >> 5);
                const bool first = (__hip_atomic_fetch_or(word, bit, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & bit) == 0u;
                if (mu + (double)__int_as_float(wb) < th) to_near = first;
                else to_far = first;
            }
        }
        // ---- reserve list space for the block, then write
        int cc = 0, c0 = 0;
        if (to_near) { c0 = st.chunk_off[v]; cc = st.chunk_off[v + 1] - c0; }
        int near_total;
        const int near_before = slot_scan16(cc, (int)s, &near_total);
        const unsigned fm = (unsigned)((__ballot(to_far) >> sh) & 0xffffull);
        const int pr = r & 1;
        if (s == 0) { sh_cnt[pr][0][slot_in_block] = near_total; sh_cnt[pr][1][slot_in_block] = __popc(fm); }
        __syncthreads();
        if (threadIdx.x < 2) {
            int tot = 0;
#pragma unroll
            for (int k = 0; k < SLOTS; ++k) tot += sh_cnt[pr][threadIdx.x][k];
            sh_base[pr][threadIdx.x] = tot ? push_atomic_add(threadIdx.x == 0 ? ncount : fcount, tot) : 0;
        }
        __syncthreads();
        int nbase = sh_base[pr][0], fbase = sh_base[pr][1];
        for (int k = 0; k < slot_in_block; ++k) { nbase += sh_cnt[pr][0][k]; fbase += sh_cnt[pr][1][k]; }
        for (int t = 0; t < cc; ++t) near_out[nbase + near_before + t] = c0 + t;
        if (to_far) far_out[fbase + __popc(fm & ((1u << s) - 1u))] = v;
    }
    return true;
}

// ---- one launch per sweep.  Block -> batch: blocks are dealt round-robin over the 8 XCDs, so  batch = xcd + 8 * (..)
// keeps ALL blocks of a batch on one XCD (its rows, stamps and counters stay in that XCD's L2 during the launch).
template <bool WEIGHTED>
__global__ __launch_bounds__(256) void push_sweep_kernel(const int32_t *__restrict__ indptr, const int32_t *__restrict__ indices,
                                                        const float *__restrict__ weights, int32_t n, int32_t nb,
                                                        int32_t blocks_per_batch, double *dist, PushState st, double delta,
                                                        int32_t sweep) {
    const int groups8 = (nb + 7) >> 3;
    const int q = blockIdx.x >> 3;
    const int b = (blockIdx.x & 7) + 8 * (q % groups8), xb = q / groups8;
    if (blockIdx.x == 0 && threadIdx.x == 0) st.active[(sweep + 1) % 3] = 0;
    if (b >= nb) return;
    const bool busy = push_batch_sweep<WEIGHTED, 16>(indptr, indices, weights, n, nb, b, xb, blocks_per_batch, dist, st, delta, sweep);
    if (busy && xb == 0 && threadIdx.x == 0) st.active[sweep % 3] = 1;
}

__global__ __launch_bounds__(256) void push_init_kernel(const int32_t *__restrict__ src_pad, int32_t n, int32_t nb, PushState st,
                                                       double delta) {
    // one block per batch: near list = the chunks of the batch's distinct sources, everything else empty
    const int b = blockIdx.x;
    if (threadIdx.x < 3) {
        st.near_cnt[threadIdx.x * st.cs + b] = 0;
        st.far_cnt[threadIdx.x * st.cs + b] = 0;
    }
    if (threadIdx.x < 2) {
        st.theta[threadIdx.x * st.cs + b] = delta;
        st.far_slot[threadIdx.x * st.cs + b] = 0;
    }
    if (b == 0 && threadIdx.x < 16) st.active[threadIdx.x] = 0;
    __syncthreads();
    if (threadIdx.x < 16) {
        const int32_t v = src_pad[b * 16 + threadIdx.x];
        // (dedupe of repeated sources through parity-0 bits; sweep 0 clears them again)
        if (v >= 0 && v < n && (atomicOr(&st.near_bits[(size_t)b * st.words + (v >> 5)], 1u << (v & 31)) & (1u << (v & 31))) == 0u) {
            const int32_t c0 = st.chunk_off[v], cc = st.chunk_off[v + 1] - c0;
            const int32_t base = atomicAdd(&st.near_cnt[b], cc);                   // parity 0, ring slot 0 = sweep 0
            for (int32_t t = 0; t < cc; ++t) st.near_lists[(size_t)b * st.cap + base + t] = c0 + t;
        }
    }
}

__global__ __launch_bounds__(256) void weight_sum_kernel(const float *__restrict__ w, int64_t nnz, double *__restrict__ out) {
    double acc = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nnz; i += (int64_t)gridDim.x * blockDim.x) acc += (double)w[i];
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) acc += __shfl_xor(acc, off, 64);
    if ((threadIdx.x & 63) == 0) atomicAdd(out, acc);
}

size_t push_bytes(int32_t n, int64_t nnz, int32_t nb) {
    const size_t cs = ((size_t)nb + 31) / 32 * 32, per = (size_t)nb * n;
    const size_t cap = (size_t)n + (size_t)(nnz / 16) + 16;                     // chunks of the graph (upper bound)
    return geo::align_up(2 * (size_t)nb * cap * 4) + geo::align_up(3 * per * 4) + 2 * geo::align_up(per * 4) +
           geo::align_up(3 * cs * 4) * 2 + geo::align_up(2 * cs * 4) + geo::align_up(2 * cs * 8) + 512 + 256 + 4096;
}

__device__ __forceinline__ double units_to_f64(uint32_t u, double unit) { return u == U_INF ? inf64() : (double)u * unit; }

// any distance that saturated?  (checked once, at the fixed point)
__global__ __launch_bounds__(256) void overflow_scan_kernel(const uint32_t *__restrict__ dist, int64_t total, int32_t *flag) {
    bool any = false;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x)
        any |= dist[i] == U_OVF;
    if (__any(any) && (threadIdx.x & 63) == 0) *flag = 1;
}

__global__ __launch_bounds__(256) void transpose_out32_kernel(const uint32_t *__restrict__ in, float *__restrict__ out,
                                                             int32_t n, int32_t n_sources, double unit,
                                                             const int32_t *__restrict__ row_of) {
    __shared__ float tile[64][33];
    const int b = blockIdx.y;
    const int32_t v0 = blockIdx.x * 64;
    for (int i = threadIdx.x; i < 64 * 32; i += 256) {
        const int vi = i >> 5, si = i & 31;
        if (v0 + vi < n) tile[vi][si] = (float)units_to_f64(in[((size_t)b * n + v0 + vi) * 32 + si], unit);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 64 * 32; i += 256) {
        const int si = i >> 6, vi = i & 63;
        const int32_t slot = b * 32 + si;
        if (slot < n_sources && v0 + vi < n) out[(size_t)row_of[slot] * n + v0 + vi] = tile[vi][si];
    }
}

__global__ __launch_bounds__(256) void colmin32_kernel(const uint32_t *__restrict__ dist, int32_t n, int32_t n_sources,
                                                      double unit, const int32_t *__restrict__ row_of,
                                                      float *__restrict__ dmin, int32_t *__restrict__ argmin) {
    const int lane = threadIdx.x & 63;
    const int32_t v = blockIdx.x * WAVES_PER_BLOCK + (threadIdx.x >> 6);
    if (v >= n) return;
    float best = __int_as_float(0x7f800000);
    int32_t barg = 0x7fffffff;
    for (int32_t r0 = 0; r0 < n_sources; r0 += 64) {
        const int32_t row = r0 + lane;
        float val = __int_as_float(0x7f800000);
        if (row < n_sources) val = (float)units_to_f64(dist[((size_t)(row >> 5) * n + v) * 32 + (row & 31)], unit);
        int32_t idx = row < n_sources ? row_of[row] : 0x7fffffff;     // ties: lowest row of the CALLER's order
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) {
            const float ov = __shfl_xor(val, off, 64);
            const int32_t oi = __shfl_xor(idx, off, 64);
            if (ov < val || (ov == val && oi < idx)) { val = ov; idx = oi; }
        }
        if (val < best || (val == best && idx < barg)) { best = val; barg = idx; }
    }
    if (lane == 0) {
        if (dmin) dmin[v] = best;
        if (argmin) argmin[v] = barg;
    }
}

// predecessors on the integer distances: same tie rule as pred_multi_kernel
__global__ __launch_bounds__(256) void pred_multi32_kernel(const int32_t *__restrict__ indptr,
                                                          const int32_t *__restrict__ indices,
                                                          const uint32_t *__restrict__ wunits, int32_t n, int32_t nb,
                                                          const uint32_t *__restrict__ dist,
                                                          const int32_t *__restrict__ src, int32_t *__restrict__ pred) {
    const int64_t total = (int64_t)nb * n * 32;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int s = (int)(i & 31);
        const int64_t r = i >> 5;
        const int32_t v = (int32_t)(r % n);
        const int32_t b = (int32_t)(r / n);
        const uint32_t *D = dist + (size_t)b * n * 32;
        const uint32_t dv = dist[i];
        int32_t p = -9999;
        uint32_t dp = U_INF;
        if (dv < U_OVF && v != src[b * 32 + s]) {
            for (int32_t e = indptr[v]; e < indptr[v + 1]; ++e) {
                const int32_t u = indices[e];
                const uint32_t du = D[(size_t)u * 32 + s];
                if (du < U_OVF && du + wunits[e] == dv && (du < dp || (du == dp && u < p))) { dp = du; p = u; }
            }
        }
        pred[i] = p;
    }
}

// dist[b][v][s] -> out[row_of[b*sb+s]][v], 64-node x sb-source tiles through LDS.
template <typename TIn, typename TOut>
__global__ __launch_bounds__(256) void transpose_out_kernel(const TIn *__restrict__ in, TOut *__restrict__ out,
                                                           int32_t n, int32_t n_sources, int32_t sb,
                                                           const int32_t *__restrict__ row_of) {
    __shared__ TOut tile[64][65];
    const int b = blockIdx.y;
    const int32_t v0 = blockIdx.x * 64;
    for (int i = threadIdx.x; i < 64 * sb; i += 256) {
        const int vi = i / sb, si = i % sb;
        if (v0 + vi < n) tile[vi][si] = (TOut)in[((size_t)b * n + v0 + vi) * sb + si];
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 64 * sb; i += 256) {
        const int si = i / 64, vi = i % 64;
        const int32_t slot = b * sb + si;                   // row_of: the caller's row of this (batch, slot)
        if (slot < n_sources && v0 + vi < n) out[(size_t)row_of[slot] * n + v0 + vi] = tile[vi][si];
    }
}

// Column minimum of the f32 matrix and the first row attaining it (D.argmin(axis=0)).
__global__ __launch_bounds__(256) void colmin_kernel(const double *__restrict__ dist, int32_t n, int32_t n_sources,
                                                    int32_t sb, const int32_t *__restrict__ row_of,
                                                    float *__restrict__ dmin, int32_t *__restrict__ argmin) {
    const int lane = threadIdx.x & 63;
    const int32_t v = blockIdx.x * WAVES_PER_BLOCK + (threadIdx.x >> 6);
    if (v >= n) return;
    float best = __int_as_float(0x7f800000);
    int32_t barg = 0x7fffffff;
    for (int32_t r0 = 0; r0 < n_sources; r0 += 64) {           // 64 consecutive source slots per step
        const int32_t row = r0 + lane;
        float val = __int_as_float(0x7f800000);
        if (row < n_sources) val = (float)dist[((size_t)(row / sb) * n + v) * sb + (row % sb)];
        int32_t idx = row < n_sources ? row_of[row] : 0x7fffffff;     // ties: lowest row of the CALLER's order
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) {
            const float ov = __shfl_xor(val, off, 64);
            const int32_t oi = __shfl_xor(idx, off, 64);
            if (ov < val || (ov == val && oi < idx)) { val = ov; idx = oi; }
        }
        if (val < best || (val == best && idx < barg)) { best = val; barg = idx; }
    }
    if (lane == 0) {
        if (dmin) dmin[v] = best;
        if (argmin) argmin[v] = barg;
    }
}

// Predecessor of every (source, node): the in-neighbour u with fl(d[u] + w) == d[v], smallest d[u]
// first, then smallest index.  Written as pred[b][v][sb] (i32) for the transposing store.
template <bool WEIGHTED>
__global__ __launch_bounds__(256) void pred_multi_kernel(const int32_t *__restrict__ indptr,
                                                        const int32_t *__restrict__ indices,
                                                        const float *__restrict__ weights, int32_t n, int32_t nb,
                                                        int32_t sb, const double *__restrict__ dist,
                                                        const int32_t *__restrict__ src, int32_t *__restrict__ pred) {
    const int64_t total = (int64_t)nb * n * sb;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int s = (int)(i % sb);
        const int64_t r = i / sb;
        const int32_t v = (int32_t)(r % n);
        const int32_t b = (int32_t)(r / n);
        const double *D = dist + (size_t)b * n * sb;
        const double dv = dist[i];
        int32_t p = -9999;
        double dp = inf64();
        if (dv < inf64() && v != src[b * sb + s]) {
            for (int32_t e = indptr[v]; e < indptr[v + 1]; ++e) {
                const int32_t u = indices[e];
                const double du = D[(size_t)u * sb + s];
                const double w = WEIGHTED ? (double)weights[e] : 1.0;
                if (du + w == dv && (du < dp || (du == dp && u < p))) { dp = du; p = u; }
            }
        }
        pred[i] = p;
    }
}

// ------------------------------------------------------------------------------------ single source
__global__ __launch_bounds__(256) void init_single_kernel(double *__restrict__ d, int32_t n, int32_t source,
                                                         int32_t *flags) {
    for (int32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x)
        d[i] = (i == source) ? 0.0 : inf64();
    if (blockIdx.x == 0 && threadIdx.x < 3) flags[threadIdx.x] = 0;
}

template <bool WEIGHTED>
__global__ __launch_bounds__(256) void sweep_single_kernel(const int32_t *__restrict__ indptr,
                                                          const int32_t *__restrict__ indices,
                                                          const float *__restrict__ weights, int32_t n, double *d,
                                                          int32_t *flags, int prev, int cur, int next, int first) {
    if (blockIdx.x == 0 && threadIdx.x == 0) flags[next] = 0;
    if (!first && flags[prev] == 0) return;
    if (geo::sweep_single_body<WEIGHTED>(indptr, indices, weights, n, d)) flags[cur] = 1;
}

__global__ __launch_bounds__(256) void finish_single_kernel(const double *__restrict__ d, int32_t n,
                                                           float *__restrict__ d_out, float *__restrict__ dmin,
                                                           int32_t *__restrict__ argmin, int32_t center_pos) {
    for (int32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const float x = (float)d[i];
        if (d_out) d_out[i] = x;
        if (dmin && x < dmin[i]) {
            dmin[i] = x;
            if (argmin) argmin[i] = center_pos;
        }
    }
}

// device time of the relaxation sweeps of the last geo_sssp_multi call (HIP events on its stream)
// (per host thread: two threads may each drive a build on their own stream and workspace)
thread_local double g_last_sweep_ms = 0.0;
thread_local int32_t g_last_sweep_launches = 0;
thread_local int32_t g_last_layout = 0;             // sources per batch of the last geo_sssp_multi call, +1000 when chunked
thread_local hipEvent_t g_ev0 = nullptr, g_ev1 = nullptr;

// fp64 single-source solve into d[n] (flags: 4 ints); `group` sweeps between convergence checks
int solve_single(const int32_t *indptr, const int32_t *indices, const float *weights, int32_t n, int32_t source,
                 double *d, int32_t *flags, int group, hipStream_t stream, int32_t *sweeps_out) {
    init_single_kernel<<<geo::grid_for(n, 256, 2048), 256, 0, stream>>>(d, n, source, flags);
    GEO_LAUNCH_CHECK();
    const int grid = geo::grid_for(n, 16, 2048);      // 16 nodes per 256-thread block
    int32_t sweeps = 0, hflag = 1;
    const int64_t limit = (int64_t)n + 2;
    while (hflag) {
        int last_cur = 0;
        for (int g = 0; g < group; ++g, ++sweeps) {
            const int cur = sweeps % 3, prev = (sweeps + 2) % 3, next = (sweeps + 1) % 3;
            if (weights)
                sweep_single_kernel<true><<<grid, 256, 0, stream>>>(indptr, indices, weights, n, d, flags, prev, cur,
                                                                    next, sweeps == 0);
            else
                sweep_single_kernel<false><<<grid, 256, 0, stream>>>(indptr, indices, weights, n, d, flags, prev, cur,
                                                                     next, sweeps == 0);
            GEO_LAUNCH_CHECK();
            last_cur = cur;
        }
        GEO_HIP_CHECK(hipMemcpyAsync(&hflag, flags + last_cur, sizeof(int32_t), hipMemcpyDeviceToHost, stream));
        GEO_HIP_CHECK(hipStreamSynchronize(stream));
        if (hflag && sweeps > limit) {
            geo::set_error("single-source solve: no fixed point after %d sweeps", sweeps);
            return GEO_E_NOCONV;
        }
    }
    if (sweeps_out) *sweeps_out = sweeps;
    return GEO_OK;
}

__global__ void gather_i32_kernel(const int32_t *__restrict__ a, const int32_t *__restrict__ at, int32_t m, int32_t *__restrict__ out) {
    for (int32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < m; i += gridDim.x * blockDim.x) out[i] = a[at[i]];
}

__global__ void gather_f32_kernel(const double *__restrict__ d, const int32_t *__restrict__ at, int32_t m,
                                  float *__restrict__ out) {
    for (int32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < m; i += gridDim.x * blockDim.x) out[i] = (float)d[at[i]];
}

// largest finite entry of a non-negative fp64 vector (bit patterns of non-negative doubles order like the doubles)
__global__ __launch_bounds__(256) void max_finite_kernel(const double *__restrict__ d, int32_t n, unsigned long long *__restrict__ out) {
    unsigned long long m = 0ull;
    for (int32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const unsigned long long b = (unsigned long long)__double_as_longlong(d[i]);
        if (b < 0x7ff0000000000000ull && b > m) m = b;
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        const unsigned long long o = __shfl_xor(m, off, 64);
        m = o > m ? o : m;
    }
    if ((threadIdx.x & 63) == 0 && m) atomicMax(out, m);
}

// ---- ordering the sources by the cells of the graph they sit in (cheap replacement of two landmark solves) --------------
// Multi-source BFS by hop count: label[v] = position of the source whose front reaches v first (ties: the first neighbour in
// CSR order that was reached one level earlier); one launch per level, a level only reads level - 1.
__global__ __launch_bounds__(256) void cell_seed_kernel(const int32_t *__restrict__ src, int32_t n_sources, int32_t n,
                                                       int32_t *__restrict__ label, int32_t *__restrict__ level_of,
                                                       int32_t *__restrict__ cell_of_source) {
    for (int32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n_sources; i += gridDim.x * blockDim.x) {
        const int32_t v = src[i];
        if (v < 0 || v >= n) continue;
        atomicMin(reinterpret_cast<unsigned *>(&label[v]), (unsigned)i);      // repeated sources: the first position owns the cell
        level_of[v] = 0;
    }
}
__global__ __launch_bounds__(256) void cell_grow_kernel(const int32_t *__restrict__ indptr, const int32_t *__restrict__ indices, int32_t n,
                                                       int32_t *label, int32_t *level_of, int32_t level, int32_t *__restrict__ changed) {
    bool any = false;
    for (int32_t v = blockIdx.x * blockDim.x + threadIdx.x; v < n; v += gridDim.x * blockDim.x) {
        if (level_of[v] >= 0) continue;
        for (int32_t e = indptr[v]; e < indptr[v + 1]; ++e) {
            const int32_t u = indices[e];
            if (level_of[u] == level - 1) {
                label[v] = label[u];
                level_of[v] = level;
                any = true;
                break;
            }
        }
    }
    if (__any(any) && (threadIdx.x & 63) == 0) *changed = 1;
}
__global__ __launch_bounds__(256) void cell_adjacency_kernel(const int32_t *__restrict__ indptr, const int32_t *__restrict__ indices, int32_t n,
                                                            const int32_t *__restrict__ label, int32_t words, uint32_t *__restrict__ adj) {
    for (int32_t v = blockIdx.x * blockDim.x + threadIdx.x; v < n; v += gridDim.x * blockDim.x) {
        const int32_t lv = label[v];
        if (lv < 0) continue;
        for (int32_t e = indptr[v]; e < indptr[v + 1]; ++e) {
            const int32_t lu = label[indices[e]];
            if (lu >= 0 && lu != lv) atomicOr(&adj[(size_t)lv * words + (lu >> 5)], 1u << (lu & 31));
        }
    }
}
constexpr int32_t CELL_ORDER_MAX_SOURCES = 16384;       // quotient graph as a bitmap: 32 MB at this size

struct MultiWs {
    double *dist;
    int32_t *pred;
    int32_t *flags;
    int32_t *src_pad;
};

// sources per batch.  64 (one 512-byte row per node and wave, CSR row through the scalar cache) is the
// fastest layout measured on MI355X at every graph size: a 16-source batch would fit one XCD's L2 at
// N=60 000, but its per-lane addressing and four CSR rows per wave cost more than the L2 hits return
// (7-10 TB/s of gathered bytes against 9-19 TB/s).  16 is kept for calls with few sources, where it
// avoids relaxing padded lanes.
// The 16- and 32-source row kernels form gather addresses as 32-bit BYTE offsets from a batch's base (node << 7 | ...):
// a batch of n nodes spans n * 128 bytes, so these layouts exist only below 2^25 nodes.
constexpr int32_t ROW128_MAX_NODES = 1 << 25;

int choose_sb(int32_t n, int32_t n_sources, int forced_here) {
    const int forced = forced_here ? forced_here : geo::options().sssp_sb;   // (option: experiment switch)
    if (forced == 16 || forced == 64) return forced;
    if (n_sources <= 16) return 16;
    if (n >= ROW128_MAX_NODES) return 64;                                    // 64-bit indexing only (sweep_multi_kernel)
    if (n_sources >= 32 && (size_t)n * 512 > ((size_t)6 << 20)) {
        // 16-source batches pay off while one batch (n * 128 bytes) stays within reach of an XCD's 4 MiB L2 ...
        if ((size_t)n * 128 <= ((size_t)12 << 20)) return 16;
        // ... and beyond that as the carrier of the 32-bit fixed-point solve (half the gathered bytes per source:
        // 145 ms against 414 ms for 1024 sources at N = 1 M); weights that do not qualify come back for 64
        if (geo::options().sssp_u32 != 0) return 16;
    }
    return 64;
}

size_t chunk_bytes(int32_t n, int64_t nnz, int32_t nb) {
    const size_t max_chunks = (size_t)n + (size_t)(nnz / 16) + 16;
    const size_t words = ((size_t)n + 3) / 4;
    return 2 * geo::align_up(((size_t)n + 1) * 4) + 2 * geo::align_up(max_chunks * 4) +
           geo::align_up(geo::scan_tmp_bytes((int64_t)n + 1)) + geo::align_up(3 * (size_t)nb * words * 4) +
           geo::align_up(4 * (((size_t)nb + 31) / 32 * 32) * 4) + geo::align_up((size_t)n * 8) + geo::align_up(2 * (size_t)nb * 16 * 4) +
           geo::align_up((size_t)(nnz > 0 ? nnz : 1) * 4) + 256 + 256 + geo::align_up((size_t)n * 4) + 1024 + push_bytes(n, nnz, nb) +
           geo::align_up((size_t)std::min(nb * 16, CELL_ORDER_MAX_SOURCES) * ((std::min(nb * 16, CELL_ORDER_MAX_SOURCES) + 31) / 32) * 4);
}

// slots of a solve: whole batches of `sb` sources, and (for the 32-source fixed-point layout of the same buffers) of 32
size_t padded_slots(int32_t nb, int32_t sb) { return ((size_t)nb * sb + 31) / 32 * 32; }

size_t multi_bytes(int32_t n, int32_t nb, int32_t sb, bool with_pred) {
    const size_t slots = padded_slots(nb, sb);
    size_t b = geo::align_up(slots * n * sizeof(double));
    if (with_pred) b += geo::align_up(slots * n * sizeof(int32_t));
    b += geo::align_up(3 * (((size_t)nb + 31) / 32 * 32) * sizeof(int32_t)) + 2 * geo::align_up(slots * sizeof(int32_t));
    return b;
}

}  // namespace

extern "C" size_t geo_sssp_workspace_bytes(int32_t n, int64_t nnz, int32_t n_sources) {
    if (n < 0 || n_sources < 0 || nnz < 0) return 0;
    const int32_t nn = n > 0 ? n : 1, ss = n_sources > 0 ? n_sources : 1;
    size_t multi = 0;
    for (int sb : {16, 64}) {
        const size_t m = multi_bytes(nn, (ss + sb - 1) / sb, sb, true);
        multi = m > multi ? m : multi;
    }
    multi += chunk_bytes(nn, nnz, (ss + 15) / 16);
    size_t single = geo::align_up((size_t)nn * sizeof(double)) + 256;
    return (multi > single ? multi : single) + 1024;
}

static int sssp_multi_impl(const int32_t *indptr, const int32_t *indices, const float *weights, int32_t n,
                           int64_t nnz, const int32_t *sources, int32_t n_sources, float *D_out, int32_t *P_out,
                           float *dmin_out, int32_t *argmin_out, void *ws, size_t ws_bytes,
                           int32_t *sweeps_out, void *stream_, int force_sb) {
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    GEO_REQUIRE(n > 0 && n_sources > 0, "geo_sssp_multi: n=%d n_sources=%d must be positive", n, n_sources);
    GEO_REQUIRE(indptr && indices && sources && ws, "geo_sssp_multi: null pointer");
    const int sb = choose_sb(n, n_sources, force_sb);
    const int32_t nb = (n_sources + sb - 1) / sb;
    // 16-edge chunk work items (see sweep_chunk16_kernel); their 32-bit row offsets need n * 128 bytes < 2^32
    const bool chunked = sb == 16 && n_sources > 16 && n < ROW128_MAX_NODES;
    GEO_REQUIRE(nnz >= 0, "geo_sssp_multi: nnz must be given");
    if (ws_bytes < multi_bytes(n, nb, sb, P_out != nullptr) + (chunked ? chunk_bytes(n, nnz, nb) : 0)) {
        geo::set_error("geo_sssp_multi: workspace %zu < %zu", ws_bytes,
                       multi_bytes(n, nb, sb, P_out != nullptr) + (chunked ? chunk_bytes(n, nnz, nb) : 0));
        return GEO_E_WORKSPACE;
    }
    geo::Arena ar(ws, ws_bytes);
    MultiWs w;
    const size_t slots = padded_slots(nb, sb);
    w.dist = ar.take<double>(slots * n);
    w.pred = P_out ? ar.take<int32_t>(slots * n) : nullptr;
    // rows of the flag / count rings start on their own 128-byte lines: every block reads the previous sweep's row while
    // the current sweep's row takes stores and atomics (a shared line made alternate sweeps 3x slower at nb = 16)
    const int cs = ((nb + 31) / 32) * 32;
    w.flags = ar.take<int32_t>(3 * (size_t)cs);
    w.src_pad = ar.take<int32_t>(slots);
    int32_t *row_of = ar.take<int32_t>(slots);                    // (batch, slot) -> row of the caller's `sources`

    int32_t *chunk_node = nullptr, *chunk_start = nullptr;
    uint32_t *bits = nullptr;
    int32_t *counts = nullptr;
    double *lm_d = nullptr;
    float *lm_key = nullptr;
    int32_t *lm_flags = nullptr;
    uint32_t *wunits = nullptr, *wrange = nullptr;           // 32-bit fixed-point weights / their range
    int32_t *chunk_cnt = nullptr, *row_order = nullptr;
    PushState push{};
    uint32_t *cell_adj = nullptr;
    double *wsum = nullptr;
    const geo::Options &opt = geo::options();
    const int act_mode = opt.sssp_act;
    // sparse body while fewer than n/sparse_div rows moved in the previous sweep; map kept below n/map_div
    const int sparse_div = opt.sssp_sparse_div > 0 ? opt.sssp_sparse_div : 8;
    const int map_div = opt.sssp_map_div > 0 ? opt.sssp_map_div : 2;
    const int group_mode = opt.sssp_group;                   // 0 never, 1 auto, 2 always
    const int32_t words = (n + 3) / 4;                       // need-map: one byte per node, in 4-byte words
    int64_t n_chunks = 0;
    if (chunked) {
        bits = ar.take<uint32_t>(3 * (size_t)nb * words);
        counts = ar.take<int32_t>(4 * (size_t)cs);
        int32_t *ccnt = ar.take<int32_t>((size_t)n + 1), *coff = ar.take<int32_t>((size_t)n + 1);
        chunk_cnt = ccnt;
        row_order = ar.take<int32_t>((size_t)n);
        (void)ar.take<int32_t>(256);                          // (layout kept in step with geo_sssp_workspace_bytes)
        const size_t max_chunks = (size_t)n + (size_t)(nnz / 16) + 16;
        chunk_node = ar.take<int32_t>(max_chunks);
        chunk_start = ar.take<int32_t>(max_chunks);
        const size_t sbytes = geo::scan_tmp_bytes((int64_t)n + 1);
        void *stmp = ar.take<char>(sbytes);
        lm_d = ar.take<double>((size_t)n);
        lm_key = ar.take<float>(2 * (size_t)nb * sb);
        lm_flags = ar.take<int32_t>(16);                    // [0..3] solve flags, [8..9] the landmark eccentricity (u64)
        wunits = ar.take<uint32_t>((size_t)(nnz > 0 ? nnz : 1));
        wrange = ar.take<uint32_t>(4);
        const size_t per = (size_t)nb * n;
        push.cs = cs;
        push.cap = (int32_t)max_chunks;
        push.chunk_off = coff;
        push.chunk_node = chunk_node;
        push.chunk_start = chunk_start;
        push.near_lists = ar.take<int32_t>(2 * (size_t)nb * max_chunks);
        push.far_lists = ar.take<int32_t>(3 * per);
        push.words = (n + 31) / 32;
        push.near_bits = ar.take<uint32_t>(2 * (size_t)nb * push.words);
        push.far_bits = ar.take<uint32_t>((size_t)nb * push.words);
        push.near_cnt = ar.take<int32_t>(3 * (size_t)cs);
        push.far_cnt = ar.take<int32_t>(3 * (size_t)cs);
        push.far_slot = ar.take<int32_t>(2 * (size_t)cs);
        push.theta = ar.take<double>(2 * (size_t)cs);
        push.active = ar.take<int32_t>(64);
        (void)ar.take<int32_t>(512);                          // (layout kept in step with geo_sssp_workspace_bytes)
        {
            const size_t S = (size_t)std::min(nb * 16, CELL_ORDER_MAX_SOURCES);
            cell_adj = ar.take<uint32_t>(S * ((S + 31) / 32));
        }
        wsum = ar.take<double>(8);
        GEO_REQUIRE(bits && counts && stmp && lm_d && lm_key && lm_flags && wunits && wrange && push.near_lists &&
                        push.far_lists && push.near_bits && push.far_bits && push.near_cnt && push.far_cnt && push.far_slot &&
                        push.theta && push.active && cell_adj && wsum,
                    "geo_sssp_multi: workspace carve failed");
        chunk_count_kernel<<<geo::grid_for(n, 256, 2048), 256, 0, stream>>>(indptr, n, ccnt);
        GEO_LAUNCH_CHECK();
        int rc = geo::exclusive_scan_i32(ccnt, coff, n, stmp, sbytes, &n_chunks, stream);
        if (rc) return rc;
        GEO_REQUIRE((size_t)n_chunks <= max_chunks, "geo_sssp_multi: nnz=%lld does not match the graph", (long long)nnz);
        chunk_fill_kernel<<<geo::grid_for(n, 256, 2048), 256, 0, stream>>>(indptr, n, coff, chunk_node, chunk_start);
        GEO_LAUNCH_CHECK();
    }
    // blocks per batch: enough to cover the work items, capped so that one sweep stays <= ~64k blocks
    const int nodes_per_block = (64 / sb) * WAVES_PER_BLOCK;
    const int gs = nb < 8 ? nb : 8;
    const int groups = (nb + gs - 1) / gs;
    const int cap = 65536 / (gs * groups);
    std::vector<int32_t> hflags(nb), hcounts(2 * (size_t)nb), order(n_sources), hsrc, host_sources(n_sources);
    for (int32_t i = 0; i < n_sources; ++i) order[i] = i;
    GEO_HIP_CHECK(hipMemcpyAsync(host_sources.data(), sources, (size_t)n_sources * 4, hipMemcpyDeviceToHost, stream));
    GEO_HIP_CHECK(hipStreamSynchronize(stream));
    for (int32_t i = 0; i < n_sources; ++i)
        GEO_REQUIRE(host_sources[i] >= 0 && host_sources[i] < n, "geo_sssp_multi: sources[%d] = %d is outside 0..%d", i,
                    host_sources[i], n - 1);
    if (!g_ev0) {
        GEO_HIP_CHECK(hipEventCreate(&g_ev0));
        GEO_HIP_CHECK(hipEventCreate(&g_ev1));
    }
    // ---- order the sources so that a batch holds neighbouring ones (see below: "Sources that lie close together ...") ----
    bool grouped = false;
    double landmark_ecc = 0.0;
    // (a) by the cells of the graph (`sssp_order=1`, experiment): multi-source BFS by hops gives every node the source it is
    //     nearest to; two sources are adjacent when an edge joins their cells; sources are taken in BFS order over that K-node
    //     quotient graph (host).  ~20 launches over the graph and one K x K bitmap: 0.3 ms at 60 000 nodes / 512 sources,
    //     against 5.8 ms for the two exact single-source solves of (b) on the swiss bench graph (288 sweeps each) -- but the
    //     batches it forms are worse: the push solve then takes 36.9 ms instead of 20.5 ms (measured), so (b) is the default.
    auto order_by_cells = [&]() -> int {
        int32_t *label = reinterpret_cast<int32_t *>(lm_d), *level_of = label + n;            // lm_d: n doubles
        const int32_t wordsK = (n_sources + 31) / 32;
        GEO_HIP_CHECK(hipMemsetAsync(label, 0xff, 2 * (size_t)n * sizeof(int32_t), stream));    // -1 everywhere
        GEO_HIP_CHECK(hipMemsetAsync(cell_adj, 0, (size_t)n_sources * wordsK * 4, stream));
        cell_seed_kernel<<<geo::grid_for(n_sources, 256, 64), 256, 0, stream>>>(sources, n_sources, n, label, level_of, nullptr);
        GEO_LAUNCH_CHECK();
        int32_t level = 1, hchanged = 1;
        while (hchanged && level < n + 2) {
            GEO_HIP_CHECK(hipMemsetAsync(lm_flags, 0, sizeof(int32_t), stream));
            for (int g = 0; g < 8; ++g, ++level) {
                if (g == 7) GEO_HIP_CHECK(hipMemsetAsync(lm_flags, 0, sizeof(int32_t), stream));   // the group's last level decides
                cell_grow_kernel<<<geo::grid_for(n, 256, 2048), 256, 0, stream>>>(indptr, indices, n, label, level_of, level, lm_flags);
            }
            GEO_LAUNCH_CHECK();
            GEO_HIP_CHECK(hipMemcpyAsync(&hchanged, lm_flags, sizeof(int32_t), hipMemcpyDeviceToHost, stream));
            GEO_HIP_CHECK(hipStreamSynchronize(stream));
        }
        cell_adjacency_kernel<<<geo::grid_for(n, 256, 2048), 256, 0, stream>>>(indptr, indices, n, label, wordsK, cell_adj);
        GEO_LAUNCH_CHECK();
        std::vector<uint32_t> adj((size_t)n_sources * wordsK);
        std::vector<int32_t> cell(n_sources);
        gather_i32_kernel<<<geo::grid_for(n_sources, 256, 64), 256, 0, stream>>>(label, sources, n_sources, reinterpret_cast<int32_t *>(lm_key));
        GEO_HIP_CHECK(hipMemcpyAsync(adj.data(), cell_adj, adj.size() * 4, hipMemcpyDeviceToHost, stream));
        GEO_HIP_CHECK(hipMemcpyAsync(cell.data(), lm_key, (size_t)n_sources * 4, hipMemcpyDeviceToHost, stream));
        GEO_HIP_CHECK(hipStreamSynchronize(stream));
        // BFS over the quotient graph from the cell of sources[0]; unreached cells (other components) start new searches
        std::vector<int32_t> rank(n_sources, -1), queue;
        queue.reserve(n_sources);
        int32_t next_rank = 0;
        auto search = [&](int32_t start) {
            size_t head = queue.size();
            queue.push_back(start);
            rank[start] = next_rank++;
            while (head < queue.size()) {
                const int32_t c = queue[head++];
                const uint32_t *row = adj.data() + (size_t)c * wordsK;
                for (int32_t wi = 0; wi < wordsK; ++wi) {
                    uint32_t bits = row[wi];
                    while (bits) {
                        const int32_t d = wi * 32 + __builtin_ctz(bits);
                        bits &= bits - 1;
                        if (d < n_sources && rank[d] < 0) { rank[d] = next_rank++; queue.push_back(d); }
                    }
                }
            }
        };
        search(cell[0] >= 0 ? cell[0] : 0);
        for (int32_t i = 0; i < n_sources; ++i) {
            const int32_t c = cell[i] >= 0 ? cell[i] : i;
            if (rank[c] < 0) search(c);
        }
        std::vector<uint64_t> key(n_sources);
        for (int32_t i = 0; i < n_sources; ++i) key[i] = ((uint64_t)(uint32_t)rank[cell[i] >= 0 ? cell[i] : i] << 32) | (uint32_t)i;
        std::sort(key.begin(), key.end());
        for (int32_t i = 0; i < n_sources; ++i) order[i] = (int32_t)(key[i] & 0xffffffffu);
        grouped = true;
        return GEO_OK;
    };
    // (c) along two landmark HOP counts (`sssp_order=2`): the Morton order of (b) needs two smooth coordinates on the graph,
    //     not exact distances; breadth-first levels from sources[0] and from the source farthest from it (in hops) cost one
    //     cheap launch per level instead of one full relaxation sweep per hop of an exact solve.
    auto order_by_hops = [&]() -> int {
        int32_t *label = reinterpret_cast<int32_t *>(lm_d), *level_of = label + n;
        std::vector<int32_t> ka(n_sources), kb(n_sources);
        auto bfs = [&](int32_t seed_pos, std::vector<int32_t> &out) -> int {
            GEO_HIP_CHECK(hipMemsetAsync(label, 0xff, 2 * (size_t)n * sizeof(int32_t), stream));
            cell_seed_kernel<<<1, 64, 0, stream>>>(sources + seed_pos, 1, n, label, level_of, nullptr);
            GEO_LAUNCH_CHECK();
            int32_t level = 1, hchanged = 1;
            while (hchanged && level < n + 2) {
                for (int g = 0; g < 16; ++g, ++level) {
                    if (g == 15) GEO_HIP_CHECK(hipMemsetAsync(lm_flags, 0, sizeof(int32_t), stream));   // the group's last level decides
                    cell_grow_kernel<<<geo::grid_for(n, 256, 2048), 256, 0, stream>>>(indptr, indices, n, label, level_of, level, lm_flags);
                }
                GEO_LAUNCH_CHECK();
                GEO_HIP_CHECK(hipMemcpyAsync(&hchanged, lm_flags, sizeof(int32_t), hipMemcpyDeviceToHost, stream));
                GEO_HIP_CHECK(hipStreamSynchronize(stream));
            }
            gather_i32_kernel<<<geo::grid_for(n_sources, 256, 64), 256, 0, stream>>>(level_of, sources, n_sources, reinterpret_cast<int32_t *>(lm_key));
            GEO_HIP_CHECK(hipMemcpyAsync(out.data(), lm_key, (size_t)n_sources * 4, hipMemcpyDeviceToHost, stream));
            GEO_HIP_CHECK(hipStreamSynchronize(stream));
            return GEO_OK;
        };
        if (int rc = bfs(0, ka)) return rc;
        int32_t far = 0, amax = 0;
        for (int32_t i = 0; i < n_sources; ++i)
            if (ka[i] > amax) { amax = ka[i]; far = i; }
        if (int rc = bfs(far, kb)) return rc;
        int32_t bmax = 0;
        for (int32_t i = 0; i < n_sources; ++i) bmax = std::max(bmax, kb[i]);
        std::vector<uint64_t> key(n_sources);
        for (int32_t i = 0; i < n_sources; ++i) {
            // sources the landmarks cannot reach (other components: level -1) sort last
            const uint32_t qa = ka[i] >= 0 && amax > 0 ? (uint32_t)(65535.0 * ka[i] / amax) : 65535u;
            const uint32_t qb = kb[i] >= 0 && bmax > 0 ? (uint32_t)(65535.0 * kb[i] / bmax) : 65535u;
            uint64_t m = 0;
            for (int bit = 15; bit >= 0; --bit) m = (m << 2) | (uint64_t)(((qa >> bit) & 1u) << 1) | ((qb >> bit) & 1u);
            key[i] = (m << 32) | (uint32_t)i;
        }
        std::sort(key.begin(), key.end());
        for (int32_t i = 0; i < n_sources; ++i) order[i] = (int32_t)(key[i] & 0xffffffffu);
        grouped = true;
        return GEO_OK;
    };
    // (b) along two landmark distances (round 2; `sssp_order=0`, the default): Morton order of (dist from sources[0], dist from the source
    //     farthest from it).  Also yields the eccentricity of sources[0], which tells the 32-bit solve early that it cannot fit.
    auto regroup_sources = [&]() -> int {
        if (opt.sssp_order == 2) return order_by_hops();
        if (opt.sssp_order == 1 && n_sources <= CELL_ORDER_MAX_SOURCES) return order_by_cells();
        int32_t lm_sweeps = 0;
        std::vector<float> ka(n_sources), kb(n_sources);
        const int gk = geo::grid_for(n_sources, 256, 64);
        if (int rc = solve_single(indptr, indices, weights, n, host_sources[0], lm_d, lm_flags, 16, stream, &lm_sweeps)) return rc;
        gather_f32_kernel<<<gk, 256, 0, stream>>>(lm_d, sources, n_sources, lm_key);
        unsigned long long *ecc_dev = reinterpret_cast<unsigned long long *>(lm_flags + 8), ecc_bits = 0ull;   // (8-byte aligned scratch)
        GEO_HIP_CHECK(hipMemsetAsync(ecc_dev, 0, sizeof(unsigned long long), stream));
        max_finite_kernel<<<geo::grid_for(n, 256, 1024), 256, 0, stream>>>(lm_d, n, ecc_dev);
        GEO_HIP_CHECK(hipMemcpyAsync(&ecc_bits, ecc_dev, sizeof(ecc_bits), hipMemcpyDeviceToHost, stream));
        GEO_HIP_CHECK(hipMemcpyAsync(ka.data(), lm_key, (size_t)n_sources * 4, hipMemcpyDeviceToHost, stream));
        GEO_HIP_CHECK(hipStreamSynchronize(stream));
        {
            double e;
            memcpy(&e, &ecc_bits, sizeof(e));
            landmark_ecc = e;                                  // farthest node from sources[0] (its component)
        }
        int32_t far = 0;
        float amax = 0.f;
        for (int32_t i = 0; i < n_sources; ++i)
            if (std::isfinite(ka[i]) && ka[i] > amax) { amax = ka[i]; far = i; }
        if (int rc = solve_single(indptr, indices, weights, n, host_sources[far], lm_d, lm_flags, 16, stream, &lm_sweeps)) return rc;
        gather_f32_kernel<<<gk, 256, 0, stream>>>(lm_d, sources, n_sources, lm_key);
        GEO_HIP_CHECK(hipMemcpyAsync(kb.data(), lm_key, (size_t)n_sources * 4, hipMemcpyDeviceToHost, stream));
        GEO_HIP_CHECK(hipStreamSynchronize(stream));
        float bmax = 0.f;
        for (int32_t i = 0; i < n_sources; ++i)
            if (std::isfinite(kb[i]) && kb[i] > bmax) bmax = kb[i];
        std::vector<uint64_t> key(n_sources);
        for (int32_t i = 0; i < n_sources; ++i) {
            // sources the landmarks cannot reach (other components) sort last
            const uint32_t qa = std::isfinite(ka[i]) && amax > 0.f ? (uint32_t)(65535.0f * ka[i] / amax) : 65535u;
            const uint32_t qb = std::isfinite(kb[i]) && bmax > 0.f ? (uint32_t)(65535.0f * kb[i] / bmax) : 65535u;
            uint64_t m = 0;
            for (int bit = 15; bit >= 0; --bit) m = (m << 2) | (uint64_t)(((qa >> bit) & 1u) << 1) | ((qb >> bit) & 1u);
            key[i] = (m << 32) | (uint32_t)i;
        }
        std::sort(key.begin(), key.end());
        for (int32_t i = 0; i < n_sources; ++i) order[i] = (int32_t)(key[i] & 0xffffffffu);
        grouped = true;
        return GEO_OK;
    };
    g_last_sweep_ms = 0.0;
    // ---- exact 32-bit fixed-point solve (32 sources per row) when the weights qualify; see sweep_chunk32u_kernel ----
    if (chunked && n_sources >= 32 && opt.sssp_u32 != 0 && opt.sssp_push != 2) {
        int shift = 0;
        bool eligible = true;
        if (weights) {
            const uint32_t init[4] = {0xffffffffu, 0u, 0u, 0u};
            uint32_t got[4];
            GEO_HIP_CHECK(hipMemcpyAsync(wrange, init, sizeof(init), hipMemcpyHostToDevice, stream));
            weight_range_kernel<<<geo::grid_for(nnz, 256 * 16, 512), 256, 0, stream>>>(weights, nnz, wrange);
            GEO_LAUNCH_CHECK();
            GEO_HIP_CHECK(hipMemcpyAsync(got, wrange, sizeof(got), hipMemcpyDeviceToHost, stream));
            GEO_HIP_CHECK(hipStreamSynchronize(stream));
            if (got[2]) eligible = false;                                   // negative, NaN or infinite weight
            else if (got[0] != 0xffffffffu) {                               // some positive weight
                const int e_min = (int)(got[0] >> 23) - 127, e_max = (int)(got[1] >> 23) - 127;
                eligible = (got[0] >> 23) != 0 && 24 + (e_max - e_min) <= 28;    // normal numbers, < 2^28 units each
                shift = 23 - e_min;
            }
        }
        if (eligible) {
            const int32_t nb32 = (n_sources + 31) / 32;
            const double unit = std::ldexp(1.0, -shift);
            uint32_t *dist32 = reinterpret_cast<uint32_t *>(w.dist);
            weight_units_kernel<<<geo::grid_for(nnz, 256, 2048), 256, 0, stream>>>(weights, nnz, shift, wunits);
            row_order_kernel<<<1, 1024, 0, stream>>>(chunk_cnt, n, row_order);
            GEO_LAUNCH_CHECK();
            std::vector<int32_t> hsrc32((size_t)nb32 * 32, -1), hrow32((size_t)nb32 * 32, -1), hf(nb32), hc(2 * (size_t)nb32);
            int32_t sweeps = 0, sweeps_before = 0;
            bool done = false, give_up = false;
            // a long-geodesics graph restarts ONCE with the sources ordered along the landmark distances (as the fp64
            // solve does): the fixed-point kernel then sweeps rows of 32 neighbouring sources
            for (int att32 = 0; att32 < 2 && !done && !give_up; ++att32) {
            bool restart = false;
            sweeps_before += sweeps;
            sweeps = 0;
            for (int32_t i = 0; i < n_sources; ++i) { hsrc32[i] = host_sources[order[i]]; hrow32[i] = order[i]; }
            GEO_HIP_CHECK(hipMemcpyAsync(w.src_pad, hsrc32.data(), hsrc32.size() * 4, hipMemcpyHostToDevice, stream));
            GEO_HIP_CHECK(hipMemcpyAsync(row_of, hrow32.data(), hrow32.size() * 4, hipMemcpyHostToDevice, stream));
            GEO_HIP_CHECK(hipMemsetAsync(w.flags, 0, 3 * (size_t)cs * sizeof(int32_t), stream));
            init_multi32_kernel<<<geo::grid_for((int64_t)nb32 * n * 32, 256 * 8), 256, 0, stream>>>(dist32, w.src_pad, n, nb32);
            GEO_HIP_CHECK(hipMemsetAsync(counts, 0, 4 * (size_t)cs * 4, stream));
            GEO_HIP_CHECK(hipMemsetAsync(bits, 0, 3 * (size_t)nb32 * words * 4, stream));
            source_need_kernel<<<geo::grid_for((int64_t)n_sources * 16, 256, 256), 256, 0, stream>>>(
                w.src_pad, n_sources, n, 32, words, indptr, indices, bits + 2 * (size_t)nb32 * words);
            GEO_LAUNCH_CHECK();
            const int gs32 = nb32 < 8 ? nb32 : 8, groups32 = (nb32 + gs32 - 1) / gs32;
            const int cap32_all = 65536 / (gs32 * groups32);
            // (grouped solves: many sweeps that touch few rows -- a smaller grid and the flagged-row body throughout)
            const int cap32 = (grouped && cap32_all > opt.sssp_grouped_cap) ? opt.sssp_grouped_cap : cap32_all;
            const int sdiv32 = grouped && opt.sssp_sparse_div <= 0 ? 1 : sparse_div;
            const int mdiv32 = grouped && opt.sssp_map_div <= 0 ? 1 : map_div;
            const int per_batch = geo::grid_for(n, 8, cap32 > 0 ? cap32 : 1);     // 8 row slots per block
            const unsigned grid = (unsigned)per_batch * (unsigned)gs32 * (unsigned)groups32;
            int group_len = SWEEP_GROUP;
            while (!done && !give_up && !restart) {
                int last_cur = 0;
                GEO_HIP_CHECK(hipEventRecord(g_ev0, stream));
                for (int g = 0; g < group_len; ++g, ++sweeps) {
                    const int cur = sweeps % 3, prev = (sweeps + 2) % 3, next = (sweeps + 1) % 3;
                    uint32_t *bcur = bits + (size_t)cur * nb32 * words, *bprev = bits + (size_t)prev * nb32 * words;
                    if (sweeps > 0) GEO_HIP_CHECK(hipMemsetAsync(bcur, 0, (size_t)nb32 * words * 4, stream));
                    sweep_chunk32u_kernel<<<grid, 256, 0, stream>>>(indptr, indices, wunits, n, nb32, row_order,
                                                                  per_batch, gs32, dist32, w.flags, counts, bprev,
                                                                  bcur, words, prev, cur, next, sweeps == 0, act_mode, sdiv32,
                                                                  mdiv32, (sweeps + 2) % 4, (sweeps + 3) % 4, sweeps % 4,
                                                                  (sweeps + 1) % 4, cs);
                    GEO_LAUNCH_CHECK();
                    if (opt.sssp_trace) {                              // experiment: sampled improvement counts per sweep
                        std::vector<int32_t> tc(nb32);
                        GEO_HIP_CHECK(hipMemcpy(tc.data(), counts + (size_t)(sweeps % 4) * cs, (size_t)nb32 * 4, hipMemcpyDeviceToHost));
                        long long tot = 0, neg = 0;
                        for (int32_t b = 0; b < nb32; ++b) { tot += tc[b] < 0 ? -tc[b] : tc[b]; neg += tc[b] < 0; }
                        fprintf(stderr, "[sssp-u32] sweep %d: ~%lld improved pairs (%.1f%% of pairs), %lld/%d batches without map\n", sweeps,
                                tot * 16, 100.0 * tot * 16 / ((double)nb32 * n * 32), neg, nb32);
                    }
                    last_cur = cur;
                }
                GEO_HIP_CHECK(hipEventRecord(g_ev1, stream));
                GEO_HIP_CHECK(hipMemcpyAsync(hf.data(), w.flags + (size_t)last_cur * cs, (size_t)nb32 * 4, hipMemcpyDeviceToHost, stream));
                GEO_HIP_CHECK(hipMemcpyAsync(hc.data(), counts + (size_t)((sweeps + 3) % 4) * cs, (size_t)nb32 * 4, hipMemcpyDeviceToHost, stream));
                GEO_HIP_CHECK(hipMemcpyAsync(hc.data() + nb32, counts + (size_t)((sweeps + 2) % 4) * cs, (size_t)nb32 * 4, hipMemcpyDeviceToHost, stream));
                GEO_HIP_CHECK(hipStreamSynchronize(stream));
                float ms = 0.f;
                GEO_HIP_CHECK(hipEventElapsedTime(&ms, g_ev0, g_ev1));
                g_last_sweep_ms += ms;
                done = true;
                for (int32_t b = 0; b < nb32; ++b) done = done && (hf[b] == 0);
                if (!done && att32 == 0 && !grouped && sweeps == SWEEP_GROUP && group_mode != 0) {
                    // long geodesics (few pairs moved so far): order the sources, start over
                    double moved = 0.0;
                    for (int32_t b = 0; b < 2 * nb32; ++b) moved += 16.0 * (hc[b] < 0 ? -hc[b] : hc[b]);
                    if (group_mode == 2 || moved < 0.25 * (double)nb32 * n * 32) {
                        if (int rc = regroup_sources()) return rc;
                        restart = true;
                        // some node is landmark_ecc away from sources[0]: if that alone does not fit 32 bits of units the
                        // fixed-point solve would only find out at its end -- the fp64 kernels take the (ordered) sources now
                        if (landmark_ecc / unit >= 4294967294.0) give_up = true;
                        // ordered by cells (no eccentricity known): long geodesics go to the near-far push solve, which is at
                        // least as fast as the ordered fixed-point sweeps and cannot overflow
                        if (opt.sssp_push != 0 && (opt.sssp_order != 0 || opt.sssp_push == 3)) give_up = true;
                    }
                }
                if (!done && sweeps > (int64_t)n + 2) give_up = true;
                if (sweeps >= 16 && group_len < 16) group_len *= 2;
            }
            }                                                  // att32
            sweeps += sweeps_before;
            if (done) {
                int32_t ovf = 0;
                GEO_HIP_CHECK(hipMemsetAsync(w.flags, 0, sizeof(int32_t), stream));
                overflow_scan_kernel<<<geo::grid_for((int64_t)nb32 * n * 32, 256, 2048), 256, 0, stream>>>(dist32, (int64_t)nb32 * n * 32, w.flags);
                GEO_HIP_CHECK(hipMemcpyAsync(&ovf, w.flags, sizeof(int32_t), hipMemcpyDeviceToHost, stream));
                GEO_HIP_CHECK(hipStreamSynchronize(stream));
                if (!ovf) {
                    if (sweeps_out) *sweeps_out = sweeps;
                    g_last_sweep_launches = sweeps;
                    g_last_layout = 2032;
                    const dim3 tg((unsigned)((n + 63) / 64), (unsigned)nb32);
                    if (D_out) transpose_out32_kernel<<<tg, 256, 0, stream>>>(dist32, D_out, n, n_sources, unit, row_of);
                    if (P_out) {
                        pred_multi32_kernel<<<geo::grid_for((int64_t)nb32 * n * 32, 256, 8192), 256, 0, stream>>>(
                            indptr, indices, wunits, n, nb32, dist32, w.src_pad, w.pred);
                        transpose_out_kernel<int32_t, int32_t><<<tg, 256, 0, stream>>>(w.pred, P_out, n, n_sources, 32, row_of);
                    }
                    if (dmin_out || argmin_out)
                        colmin32_kernel<<<(unsigned)((n + WAVES_PER_BLOCK - 1) / WAVES_PER_BLOCK), 256, 0, stream>>>(
                            dist32, n, n_sources, unit, row_of, dmin_out, argmin_out);
                    GEO_LAUNCH_CHECK();
                    GEO_HIP_CHECK(hipStreamSynchronize(stream));
                    return GEO_OK;
                }
            }
            g_last_sweep_ms = 0.0;                       // the fp64 solve below is the one that counts
        }
        // a graph this large took 16-source batches only for the fixed-point solve: the fp64 solve runs 64 per batch
        if (force_sb == 0 && opt.sssp_sb != 16 && opt.sssp_sb != 64 && (size_t)n * 128 > ((size_t)12 << 20))
            return sssp_multi_impl(indptr, indices, weights, n, nnz, sources, n_sources, D_out, P_out, dmin_out,
                                   argmin_out, ws, ws_bytes, sweeps_out, stream_, 64);
    }
    int32_t total_sweeps = 0;
    bool pushed = false;
    // Sources that lie close together are relaxed together: a row is evaluated whenever ANY of its batch's 16
    // sources moved a neighbour, so with 16 scattered sources every row is re-evaluated as each of 16 fronts
    // passes (and their corrections cascade), with 16 neighbouring sources the fronts pass as one.  On graphs
    // with short geodesics (a few sweeps) this does not matter; when the first sweeps reach only a small part
    // of the graph the solve restarts with the sources ordered along two landmark distances (Morton order of
    // (dist from sources[0], dist from the source farthest from it)).  Results do not depend on the grouping.
    for (int attempt = 0;; ++attempt) {
        hsrc.assign((size_t)nb * sb, -1);                           // -1 = no source in this slot
        std::vector<int32_t> hrow((size_t)nb * sb, -1);
        for (int32_t i = 0; i < n_sources; ++i) { hsrc[i] = host_sources[order[i]]; hrow[i] = order[i]; }
        GEO_HIP_CHECK(hipMemcpyAsync(w.src_pad, hsrc.data(), hsrc.size() * 4, hipMemcpyHostToDevice, stream));
        GEO_HIP_CHECK(hipMemcpyAsync(row_of, hrow.data(), hrow.size() * 4, hipMemcpyHostToDevice, stream));
        GEO_HIP_CHECK(hipMemsetAsync(w.flags, 0, 3 * (size_t)cs * sizeof(int32_t), stream));
        init_multi_kernel<<<geo::grid_for((int64_t)nb * n * sb, 256 * 8), 256, 0, stream>>>(w.dist, w.src_pad, n, nb, sb);
        GEO_LAUNCH_CHECK();
        if (chunked) {
            GEO_HIP_CHECK(hipMemsetAsync(counts, 0, 4 * (size_t)cs * 4, stream));
            GEO_HIP_CHECK(hipMemsetAsync(bits, 0, 3 * (size_t)nb * words * 4, stream));
            source_need_kernel<<<geo::grid_for((int64_t)n_sources * 16, 256, 256), 256, 0, stream>>>(
                w.src_pad, n_sources, n, sb, words, indptr, indices, bits + 2 * (size_t)nb * words);
            GEO_LAUNCH_CHECK();
        }
        if (chunked && opt.sssp_push != 0 && (grouped || opt.sssp_push == 2 || group_mode == 2)) {
            // ---- long geodesics: near-far push solve (see push_sweep_kernel) ----
            double mean_w = 1.0;
            if (weights && nnz > 0) {
                GEO_HIP_CHECK(hipMemsetAsync(wsum, 0, sizeof(double), stream));
                weight_sum_kernel<<<geo::grid_for(nnz, 256, 1024), 256, 0, stream>>>(weights, nnz, wsum);
                GEO_HIP_CHECK(hipMemcpyAsync(&mean_w, wsum, sizeof(double), hipMemcpyDeviceToHost, stream));
                GEO_HIP_CHECK(hipStreamSynchronize(stream));
                mean_w /= (double)nnz;
            }
            const double delta = (opt.sssp_delta > 0 ? (double)opt.sssp_delta : 4.0) * (mean_w > 0.0 ? mean_w : 1.0);
            auto push_reset = [&]() -> int {
                GEO_HIP_CHECK(hipMemsetAsync(push.near_bits, 0, 2 * (size_t)nb * push.words * 4, stream));
                GEO_HIP_CHECK(hipMemsetAsync(push.far_bits, 0, (size_t)nb * push.words * 4, stream));
                push_init_kernel<<<nb, 256, 0, stream>>>(w.src_pad, n, nb, push, delta);
                GEO_LAUNCH_CHECK();
                return GEO_OK;
            };
            if (int rc = push_reset()) return rc;
            const int64_t plimit = 64 * (int64_t)n + 64;
            const int bpb = opt.sssp_push_blocks > 0 ? opt.sssp_push_blocks : 64;
            const unsigned pgrid = (unsigned)(((nb + 7) / 8) * 8) * (unsigned)bpb;        // batch <-> XCD: see push_sweep_kernel
            int32_t sweeps = 0, hact = 1;
            int group_len = 8;
            while (hact) {
                int last = 0;
                GEO_HIP_CHECK(hipEventRecord(g_ev0, stream));
                for (int g = 0; g < group_len; ++g, ++sweeps) {
                    if (weights) push_sweep_kernel<true><<<pgrid, 256, 0, stream>>>(indptr, indices, weights, n, nb, bpb, w.dist, push, delta, sweeps);
                    else push_sweep_kernel<false><<<pgrid, 256, 0, stream>>>(indptr, indices, weights, n, nb, bpb, w.dist, push, delta, sweeps);
                    GEO_LAUNCH_CHECK();
                    if (opt.sssp_trace) {
                        std::vector<int32_t> hn(cs), hfc(3 * (size_t)cs);
                        GEO_HIP_CHECK(hipMemcpy(hn.data(), push.near_cnt + (size_t)((sweeps + 1) % 3) * cs, (size_t)cs * 4, hipMemcpyDeviceToHost));
                        GEO_HIP_CHECK(hipMemcpy(hfc.data(), push.far_cnt, 3 * (size_t)cs * 4, hipMemcpyDeviceToHost));
                        long long tn = 0, tf = 0, idle = 0;
                        for (int32_t b = 0; b < nb; ++b) { tn += hn[b]; long long f = std::max(hfc[b], std::max(hfc[cs + b], hfc[2 * cs + b])); tf += f; idle += (hn[b] == 0 && f == 0); }
                        fprintf(stderr, "[sssp-push] sweep %d: next near %lld, far pile <= %lld, idle batches %lld/%d\n", sweeps, tn, tf, idle, nb);
                    }
                    last = sweeps % 3;
                }
                GEO_HIP_CHECK(hipEventRecord(g_ev1, stream));
                GEO_HIP_CHECK(hipMemcpyAsync(&hact, push.active + last, sizeof(int32_t), hipMemcpyDeviceToHost, stream));
                GEO_HIP_CHECK(hipStreamSynchronize(stream));
                float ms = 0.f;
                GEO_HIP_CHECK(hipEventElapsedTime(&ms, g_ev0, g_ev1));
                g_last_sweep_ms += ms;
                if (hact && sweeps > plimit) {
                    geo::set_error("geo_sssp_multi: near-far solve did not finish within %d sweeps", sweeps);
                    return GEO_E_NOCONV;
                }
                if (sweeps >= 32 && group_len < 32) group_len *= 2;
            }
            total_sweeps += sweeps;
            pushed = true;
            break;
        }
        // grouped solves run many sweeps that touch few rows: a smaller grid keeps the idle launches cheap
        const int want_cap = opt.sssp_grouped_cap;
        const int cap_now = (grouped && cap > want_cap) ? want_cap : cap;
        // ... and stay with the flagged-row body whenever the need-map is there
        const int sdiv = grouped && opt.sssp_sparse_div <= 0 ? 1 : sparse_div;
        const int mdiv = grouped && opt.sssp_map_div <= 0 ? 1 : map_div;
        const int per_batch = chunked ? geo::grid_for(n_chunks, 16, cap_now > 0 ? cap_now : 1)
                                      : geo::grid_for(n, nodes_per_block, cap > 0 ? cap : 1);
        const unsigned grid = (unsigned)per_batch * (unsigned)gs * (unsigned)groups;
        int32_t sweeps = 0;
        bool done = false, regroup = false;
        const int64_t limit = (int64_t)n + 2;
        int group_len = SWEEP_GROUP;
        while (!done && !regroup) {
            int last_cur = 0;
            GEO_HIP_CHECK(hipEventRecord(g_ev0, stream));
            for (int g = 0; g < group_len; ++g, ++sweeps) {
                const int cur = sweeps % 3, prev = (sweeps + 2) % 3, next = (sweeps + 1) % 3;
#define GEO_SWEEP(SBT, WT)                                                                                          \
    sweep_multi_kernel<SBT, WT><<<grid, 256, 0, stream>>>(indptr, indices, weights, n, nb, per_batch, gs, w.dist, w.flags, \
                                                          prev, cur, next, sweeps == 0, cs)
                if (chunked) {
                    uint32_t *bcur = bits + (size_t)cur * nb * words, *bprev = bits + (size_t)prev * nb * words;
                    if (sweeps > 0) GEO_HIP_CHECK(hipMemsetAsync(bcur, 0, (size_t)nb * words * 4, stream));
                    if (weights)
                        sweep_chunk16_kernel<true><<<grid, 256, 0, stream>>>(indptr, indices, weights, n, nb, chunk_node, chunk_start,
                                                                             (int32_t)n_chunks, per_batch, gs, w.dist, w.flags,
                                                                             counts, bprev, bcur, words, prev, cur, next, sweeps == 0, act_mode, sdiv, mdiv,
                                                                             (sweeps + 2) % 4, (sweeps + 3) % 4, sweeps % 4, (sweeps + 1) % 4, cs);
                    else
                        sweep_chunk16_kernel<false><<<grid, 256, 0, stream>>>(indptr, indices, weights, n, nb, chunk_node, chunk_start,
                                                                              (int32_t)n_chunks, per_batch, gs, w.dist, w.flags,
                                                                              counts, bprev, bcur, words, prev, cur, next, sweeps == 0, act_mode, sdiv, mdiv,
                                                                             (sweeps + 2) % 4, (sweeps + 3) % 4, sweeps % 4, (sweeps + 1) % 4, cs);
                } else if (sb == 64) { if (weights) GEO_SWEEP(64, true); else GEO_SWEEP(64, false); }
                else                 { if (weights) GEO_SWEEP(16, true); else GEO_SWEEP(16, false); }
#undef GEO_SWEEP
                GEO_LAUNCH_CHECK();
                if (chunked && opt.sssp_trace) {            // experiment: sampled improvement counts per sweep
                    std::vector<int32_t> hc(nb);
                    GEO_HIP_CHECK(hipMemcpy(hc.data(), counts + (size_t)(sweeps % 4) * cs, (size_t)nb * 4, hipMemcpyDeviceToHost));
                    long long tot = 0, neg = 0;
                    for (int32_t b = 0; b < nb; ++b) { tot += hc[b] < 0 ? -hc[b] : hc[b]; neg += hc[b] < 0; }
                    fprintf(stderr, "[sssp] sweep %d: ~%lld improved pairs (%.1f%% of pairs), %lld/%d batches without map\n", sweeps,
                            tot * 16, 100.0 * tot * 16 / ((double)nb * n * 16), neg, nb);
                }
                last_cur = cur;
            }
            GEO_HIP_CHECK(hipEventRecord(g_ev1, stream));
            GEO_HIP_CHECK(hipMemcpyAsync(hflags.data(), w.flags + (size_t)last_cur * cs, (size_t)nb * sizeof(int32_t),
                                         hipMemcpyDeviceToHost, stream));
            if (chunked) {                       // improvement counts of the last two sweeps (4-slot ring)
                const int c_last = (sweeps + 3) % 4, c_before = (sweeps + 2) % 4;        // `sweeps` already counts them
                GEO_HIP_CHECK(hipMemcpyAsync(hcounts.data(), counts + (size_t)c_last * cs, (size_t)nb * sizeof(int32_t),
                                             hipMemcpyDeviceToHost, stream));
                GEO_HIP_CHECK(hipMemcpyAsync(hcounts.data() + nb, counts + (size_t)c_before * cs, (size_t)nb * sizeof(int32_t),
                                             hipMemcpyDeviceToHost, stream));
            }
            GEO_HIP_CHECK(hipStreamSynchronize(stream));
            float ms = 0.f;
            GEO_HIP_CHECK(hipEventElapsedTime(&ms, g_ev0, g_ev1));
            g_last_sweep_ms += ms;
            done = true;
            for (int32_t b = 0; b < nb; ++b) done = done && (hflags[b] == 0);
            if (!done && sweeps > limit) {
                geo::set_error("geo_sssp_multi: no fixed point after %d sweeps", sweeps);
                return GEO_E_NOCONV;
            }
            if (!done && chunked && !grouped && group_mode != 0 && attempt == 0 && sweeps == SWEEP_GROUP) {
                // after the first sweeps: how much of the (node, source) matrix moved in the last two?
                double moved = 0.0;
                for (int32_t b = 0; b < 2 * nb; ++b) moved += 16.0 * (hcounts[b] < 0 ? -hcounts[b] : hcounts[b]);
                regroup = group_mode == 2 || moved < 0.25 * (double)nb * n * 16;
            }
            if (sweeps >= 16 && group_len < 16) group_len *= 2;      // long solves: fewer host round trips
        }
        total_sweeps += sweeps;
        if (done) break;
        // ---- order the sources along two landmark distances, then start over ----
        if (int rc = regroup_sources()) return rc;
    }
    const int32_t sweeps = total_sweeps;
    if (sweeps_out) *sweeps_out = sweeps;
    g_last_sweep_launches = sweeps;
    g_last_layout = sb + (chunked ? 1000 : 0) + (pushed ? 3000 : 0);   // 4016 near-far push solve

    const dim3 tgrid((unsigned)((n + 63) / 64), (unsigned)nb);
    if (D_out) {
        transpose_out_kernel<double, float><<<tgrid, 256, 0, stream>>>(w.dist, D_out, n, n_sources, sb, row_of);
        GEO_LAUNCH_CHECK();
    }
    if (P_out) {
        const int pg = geo::grid_for((int64_t)nb * n * sb, 256, 8192);
        if (weights)
            pred_multi_kernel<true><<<pg, 256, 0, stream>>>(indptr, indices, weights, n, nb, sb, w.dist, w.src_pad, w.pred);
        else
            pred_multi_kernel<false><<<pg, 256, 0, stream>>>(indptr, indices, weights, n, nb, sb, w.dist, w.src_pad, w.pred);
        GEO_LAUNCH_CHECK();
        transpose_out_kernel<int32_t, int32_t><<<tgrid, 256, 0, stream>>>(w.pred, P_out, n, n_sources, sb, row_of);
        GEO_LAUNCH_CHECK();
    }
    if (dmin_out || argmin_out) {
        colmin_kernel<<<(unsigned)((n + WAVES_PER_BLOCK - 1) / WAVES_PER_BLOCK), 256, 0, stream>>>(
            w.dist, n, n_sources, sb, row_of, dmin_out, argmin_out);
        GEO_LAUNCH_CHECK();
    }
    GEO_HIP_CHECK(hipStreamSynchronize(stream));
    return GEO_OK;
}

extern "C" int geo_sssp_multi(const int32_t *indptr, const int32_t *indices, const float *weights, int32_t n,
                              int64_t nnz, const int32_t *sources, int32_t n_sources, float *D_out, int32_t *P_out,
                              float *dmin_out, int32_t *argmin_out, void *ws, size_t ws_bytes,
                              int32_t *sweeps_out, void *stream_) {
    return sssp_multi_impl(indptr, indices, weights, n, nnz, sources, n_sources, D_out, P_out, dmin_out, argmin_out, ws,
                           ws_bytes, sweeps_out, stream_, 0);
}

// ---- nearest source per node in ONE solve ("graph Voronoi") ------------------------------------------------------------
// assign_points_to_medoids (kmeans_optimized.py:77-106) needs of the K x N matrix only, per column, the minimum and the FIRST
// row attaining it -- taken AFTER dijkstra_multi_source has cast the matrix to float32 (geo_shortest_paths.py:50,
// kmeans_optimized.py:100).  In exact units (below) d(s,v) is an integer and f = float32 rounding is monotone, so the
// reference's answer for column v is
//     dmin(v) = f(min_s d(s,v)),      argmin(v) = the lowest row s with d(s,v) <= T(v),
// T(v) = the largest integer that rounds to the same float32 as min_s d(s,v).  Below 2^24 units every integer is a float32,
// T = the minimum itself and the rule is "lowest row among the exactly nearest"; above, a medoid that is farther by less than
// half an ulp ties with the nearest one, and a lower row wins (round-3 review: a 9-node graph where the exactly-nearest
// medoid is row 1 and the reference's code is 0).
//
// One label-carrying relaxation finds, per node, the TWO best (distance, row) keys with distinct rows:
//   key = (distance in units << 24) | row;  a node's pair = the two smallest keys with distinct rows over
//   {its own seeds} + {key(u) + (w(u,v) << 24) : u a neighbour, key in pair(u)}.
//   Exact: the second-best source s2 of v reaches v through a neighbour u with d(s2,u) + w = d(s2,v); were s2 not among the
//   two best of u, two other rows a, b would hold keys below (d(s2,u), s2) at u, hence (d(.,v), .) <= (d(.,u) + w, .) <
//   (d(s2,v), s2) at v -- s2 would be third at best.  Integer sums are exact and order-independent, stored keys are lengths
//   of real paths and only decrease, a sweep that changes nothing ends it.
// Then: second key beyond T(v) (or absent) -> the first key's row is the answer (ties at equal distance already went to the
// lower row).  Second key within T(v) and distances from 2^24 units on -> the node is SUSPECT: further sources may lie inside
// the window.  Suspects are resolved exactly: the graph is undirected (contract of this entry point: symmetric CSR), so
// d(s,v) = d(v,s): one multi-source solve FROM the suspects, read at the K medoids, float32 argmin with the lowest row on
// ties (the rule of colmin32_kernel).  More than NEAREST_MAX_SUSPECTS suspects (graphs built from a few distinct weights
// above 2^24 units), weights that do not fit 28 bits of the common unit, or a distance reaching 2^39 units: the call
// declines (status 1) and the caller runs geo_sssp_multi.
// Units: the largest power of two dividing every weight (lowest set mantissa bit over all weights), so unit weights are 1
// and hop counts stay below 2^24; never coarser than the 2^(e_min - 23) of the 32-bit solve.
constexpr uint64_t V_INF = ~0ull;
constexpr int V_LABEL_BITS = 24;
constexpr uint64_t V_LABEL_MASK = (1ull << V_LABEL_BITS) - 1;
constexpr uint64_t V_DIST_LIMIT = 1ull << 39;
constexpr int NEAREST_MAX_SUSPECTS = 32;

struct alignas(16) VPair { uint64_t k1, k2; };

// lo / hi bit patterns, bad flag, and the lowest set bit of any weight as (biased exponent + trailing zeros of the 24-bit
// mantissa): w = m * 2^(e - 150), lowest bit 2^(e - 150 + ctz(m))
__global__ __launch_bounds__(256) void weight_range_lowbit_kernel(const float *__restrict__ w, int64_t nnz, uint32_t *__restrict__ out) {
    uint32_t lo = 0xffffffffu, hi = 0u, bad = 0u, low = 0xffffffffu;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nnz; i += (int64_t)gridDim.x * blockDim.x) {
        const float x = w[i];
        const uint32_t b = __float_as_uint(x);
        if (!(x >= 0.0f) || b >= 0x7f800000u) bad = 1u;
        else if (b != 0u) {
            lo = b < lo ? b : lo; hi = b > hi ? b : hi;
            const uint32_t m = (b & 0x7fffffu) | 0x800000u;
            const uint32_t lb = (b >> 23) + (uint32_t)(__ffs((int)m) - 1);
            low = lb < low ? lb : low;
        }
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        const uint32_t l2 = __shfl_xor(lo, off, 64), h2 = __shfl_xor(hi, off, 64), w2 = __shfl_xor(low, off, 64);
        lo = l2 < lo ? l2 : lo; hi = h2 > hi ? h2 : hi; low = w2 < low ? w2 : low; bad |= __shfl_xor(bad, off, 64);
    }
    __shared__ uint32_t s_lo[4], s_hi[4], s_bad[4], s_low[4];
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { s_lo[wave] = lo; s_hi[wave] = hi; s_bad[wave] = bad; s_low[wave] = low; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int k = 1; k < 4; ++k) {
            lo = s_lo[k] < lo ? s_lo[k] : lo; hi = s_hi[k] > hi ? s_hi[k] : hi; low = s_low[k] < low ? s_low[k] : low; bad |= s_bad[k];
        }
        atomicMin(&out[0], lo);
        atomicMax(&out[1], hi);
        if (bad) atomicOr(&out[2], 1u);
        atomicMin(&out[3], low);
    }
}

__global__ __launch_bounds__(256) void voronoi_init_kernel(VPair *__restrict__ key, int32_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) key[i] = VPair{V_INF, V_INF};
}

// distance 0, label i; a node listed twice keeps its two lowest rows (pass 0 settles k1, pass 1 offers the others to k2)
__global__ __launch_bounds__(256) void voronoi_seed_kernel(VPair *__restrict__ key, const int32_t *__restrict__ src,
                                                          int32_t n_sources, int pass) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_sources) return;
    VPair *p = &key[src[i]];
    if (pass == 0) atomicMin(reinterpret_cast<unsigned long long *>(&p->k1), (unsigned long long)i);
    else if (p->k1 != (uint64_t)i) atomicMin(reinterpret_cast<unsigned long long *>(&p->k2), (unsigned long long)i);
}

// keep the two smallest keys with distinct labels
__device__ __forceinline__ void vpair_insert(uint64_t &b1, uint64_t &b2, uint64_t c) {
    if (c == V_INF) return;
    const uint64_t lc = c & V_LABEL_MASK;
    if (lc == (b1 & V_LABEL_MASK)) { b1 = c < b1 ? c : b1; }           // (V_INF carries label 0xffffff, no source has it)
    else if (c < b1) { b2 = b1; b1 = c; }
    else if (lc == (b2 & V_LABEL_MASK)) { b2 = c < b2 ? c : b2; }
    else if (c < b2) { b2 = c; }
}

// 16 lanes per node (pull): the slot's two best over the row's entries; ONE writer per node and sweep (plain stores), reads
// of a neighbour may see this sweep's or the last sweep's keys, or one of each -- every stored key is a real path's, and
// vpair_insert takes any mixture.
__global__ __launch_bounds__(256) void voronoi_sweep_kernel(const int32_t *__restrict__ indptr, const int32_t *__restrict__ indices,
                                                           const uint32_t *__restrict__ wunits, VPair *__restrict__ key,
                                                           int32_t n, int32_t *__restrict__ changed, int32_t *__restrict__ overflow) {
    const int lane16 = threadIdx.x & 15;
    const int64_t v = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 4;
    uint64_t b1 = V_INF, b2 = V_INF;
    bool over = false;
    if (v < n) {
        const int32_t e0 = indptr[v], e1 = indptr[v + 1];
        for (int32_t e = e0 + lane16; e < e1; e += 16) {
            const VPair ku = key[indices[e]];
            const uint64_t w = (uint64_t)wunits[e] << V_LABEL_BITS;
            if (ku.k1 != V_INF) {
                if ((ku.k1 >> V_LABEL_BITS) >= V_DIST_LIMIT) over = true;     // (checked BEFORE the add: the sum cannot wrap)
                else vpair_insert(b1, b2, ku.k1 + w);
            }
            if (ku.k2 != V_INF) {
                if ((ku.k2 >> V_LABEL_BITS) >= V_DIST_LIMIT) over = true;
                else vpair_insert(b1, b2, ku.k2 + w);
            }
        }
    }
#pragma unroll
    for (int off = 8; off >= 1; off >>= 1) {
        const uint64_t o1 = __shfl_xor((unsigned long long)b1, off, 64);
        const uint64_t o2 = __shfl_xor((unsigned long long)b2, off, 64);
        vpair_insert(b1, b2, o1);
        vpair_insert(b1, b2, o2);
    }
    if (over) *overflow = 1;
    if (v < n && lane16 == 0) {
        const VPair mine = key[v];
        vpair_insert(b1, b2, mine.k1);
        vpair_insert(b1, b2, mine.k2);
        if (b1 != mine.k1 || b2 != mine.k2) {
            key[v] = VPair{b1, b2};
            *changed = 1;                                       // (idempotent flag store)
        }
    }
}

// info[0] overflow, info[1] number of suspects; suspects[0 .. NEAREST_MAX_SUSPECTS) their nodes
__global__ __launch_bounds__(256) void voronoi_out_kernel(const VPair *__restrict__ key, int32_t n, double unit,
                                                         float *__restrict__ dmin, int32_t *__restrict__ argmin,
                                                         int32_t *__restrict__ info, int32_t *__restrict__ suspects) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const VPair k = key[i];
    if (k.k1 == V_INF) {
        if (dmin) dmin[i] = __uint_as_float(0x7f800000u);
        if (argmin) argmin[i] = 0;                              // an all-inf column keeps row 0 (numpy argmin)
        return;
    }
    const uint64_t d1 = k.k1 >> V_LABEL_BITS;
    if (d1 >= V_DIST_LIMIT) info[0] = 1;
    const float f1 = (float)((double)d1 * unit);
    if (dmin) dmin[i] = f1;
    if (argmin) argmin[i] = (int32_t)(k.k1 & V_LABEL_MASK);
    if (k.k2 != V_INF && d1 >= (1ull << 24)) {
        const uint64_t d2 = k.k2 >> V_LABEL_BITS;
        if ((float)((double)d2 * unit) == f1) {                 // the second source rounds to the same float32
            const int32_t slot = atomicAdd(&info[1], 1);
            if (slot < NEAREST_MAX_SUSPECTS) suspects[slot] = (int32_t)i;
        }
    }
}

// one wave per suspect: argmin over the K medoids of float32 d(suspect, medoid) = d(medoid, suspect), lowest row on ties
__global__ __launch_bounds__(64) void voronoi_resolve_kernel(const float *__restrict__ D, int32_t n, const int32_t *__restrict__ suspects,
                                                            const int32_t *__restrict__ src, int32_t n_sources,
                                                            float *__restrict__ dmin, int32_t *__restrict__ argmin) {
    const int lane = threadIdx.x;
    const float *row = D + (size_t)blockIdx.x * n;
    float best = __int_as_float(0x7f800000);
    int32_t barg = 0x7fffffff;
    for (int32_t j = lane; j < n_sources; j += 64) {
        const float val = row[src[j]];
        if (val < best) { best = val; barg = j; }               // (rows ascend per lane: strict < keeps the lowest)
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        const float ov = __shfl_xor(best, off, 64);
        const int32_t oi = __shfl_xor(barg, off, 64);
        if (ov < best || (ov == best && oi < barg)) { best = ov; barg = oi; }
    }
    if (lane == 0) {
        const int32_t v = suspects[blockIdx.x];
        if (argmin) argmin[v] = barg;
        if (dmin) dmin[v] = best;                               // (equal to what the relaxation wrote: f is monotone)
    }
}

extern "C" size_t geo_sssp_nearest_workspace_bytes(int32_t n, int64_t nnz) {
    if (n <= 0 || nnz < 0) return 1024;
    return geo::align_up((size_t)n * sizeof(VPair)) + geo::align_up((size_t)nnz * 4) + 4096 +
           geo::align_up((size_t)NEAREST_MAX_SUSPECTS * n * sizeof(float)) +
           geo_sssp_workspace_bytes(n, nnz, NEAREST_MAX_SUSPECTS) + 256;
}

extern "C" int geo_sssp_nearest_source(const int32_t *indptr, const int32_t *indices, const float *weights, int32_t n,
                                       int64_t nnz, const int32_t *sources, int32_t n_sources, float *dmin_out,
                                       int32_t *argmin_out, void *ws, size_t ws_bytes, int32_t *status_out,
                                       void *stream_) {
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    GEO_REQUIRE(indptr && indices && sources && ws && status_out, "geo_sssp_nearest_source: null pointer");
    GEO_REQUIRE(n > 0 && n_sources > 0, "geo_sssp_nearest_source: n=%d n_sources=%d", n, n_sources);
    status_out[0] = 1;                                         // declined unless everything below holds
    status_out[1] = 0;                                         // sweeps
    status_out[2] = 0;                                         // suspect nodes (float32 collisions) found
    status_out[3] = 0;                                         // why declined: 1 weights, 2 distance limit, 3 too many suspects, 4 K
    if (n_sources >= (1 << V_LABEL_BITS) - 1) { status_out[3] = 4; return GEO_OK; }
    geo::Arena ar(ws, ws_bytes);
    VPair *key = ar.take<VPair>((size_t)n);
    uint32_t *wunits = ar.take<uint32_t>((size_t)(nnz > 0 ? nnz : 1));
    uint32_t *scratch = ar.take<uint32_t>(128);
    float *Dsus = ar.take<float>((size_t)NEAREST_MAX_SUSPECTS * n);
    const size_t multi_bytes_needed = geo_sssp_workspace_bytes(n, nnz, NEAREST_MAX_SUSPECTS);
    char *multi_ws = ar.take<char>(multi_bytes_needed);
    if (!key || !wunits || !scratch || !Dsus || !multi_ws) {
        geo::set_error("geo_sssp_nearest_source: workspace %zu too small", ws_bytes);
        return GEO_E_WORKSPACE;
    }
    int shift = 0;
    if (weights && nnz > 0) {
        const uint32_t init[4] = {0xffffffffu, 0u, 0u, 0xffffffffu};
        uint32_t got[4];
        GEO_HIP_CHECK(hipMemcpyAsync(scratch, init, sizeof(init), hipMemcpyHostToDevice, stream));
        weight_range_lowbit_kernel<<<geo::grid_for(nnz, 256 * 16, 512), 256, 0, stream>>>(weights, nnz, scratch);
        GEO_LAUNCH_CHECK();
        GEO_HIP_CHECK(hipMemcpyAsync(got, scratch, sizeof(got), hipMemcpyDeviceToHost, stream));
        GEO_HIP_CHECK(hipStreamSynchronize(stream));
        status_out[3] = 1;
        if (got[2]) return GEO_OK;                              // negative, NaN or infinite weight: not ours
        if (got[0] != 0xffffffffu) {
            if ((got[0] >> 23) == 0) return GEO_OK;             // subnormal weights
            const int e_max = (int)(got[1] >> 23) - 127;
            const int low = (int)got[3] - 150;                  // every weight is a multiple of 2^low
            if (e_max + 1 - low > 28) return GEO_OK;            // a weight of 2^28 units or more
            shift = -low;
        }
        status_out[3] = 0;
        // (zero weights are 0 units: exact)
    }
    weight_units_kernel<<<geo::grid_for(nnz > 0 ? nnz : 1, 256, 2048), 256, 0, stream>>>(weights, nnz, shift, wunits);
    voronoi_init_kernel<<<geo::grid_for(n, 256), 256, 0, stream>>>(key, n);
    voronoi_seed_kernel<<<geo::grid_for(n_sources, 256), 256, 0, stream>>>(key, sources, n_sources, 0);
    voronoi_seed_kernel<<<geo::grid_for(n_sources, 256), 256, 0, stream>>>(key, sources, n_sources, 1);
    GEO_LAUNCH_CHECK();
    int32_t *flags = reinterpret_cast<int32_t *>(scratch);     // [0..7] one flag per launch of a batch, [8] overflow
    int32_t *info = flags + 16;                                 // [0] overflow at the end, [1] suspects
    int32_t *suspects = flags + 32;                             // NEAREST_MAX_SUSPECTS nodes
    constexpr int BATCH = 8;
    int32_t hflags[BATCH + 1];
    const unsigned grid = (unsigned)(((int64_t)n * 16 + 255) / 256);
    int sweeps = 0;
    bool done = false;
    for (int round = 0; round < 4096 && !done; ++round) {
        GEO_HIP_CHECK(hipMemsetAsync(flags, 0, sizeof(hflags), stream));
        for (int b = 0; b < BATCH; ++b)
            voronoi_sweep_kernel<<<grid, 256, 0, stream>>>(indptr, indices, wunits, key, n, flags + b, flags + BATCH);
        GEO_LAUNCH_CHECK();
        GEO_HIP_CHECK(hipMemcpyAsync(hflags, flags, sizeof(hflags), hipMemcpyDeviceToHost, stream));
        GEO_HIP_CHECK(hipStreamSynchronize(stream));
        if (hflags[BATCH]) { status_out[3] = 2; status_out[1] = sweeps + BATCH; return GEO_OK; }   // an upper bound reached 2^39 units
        for (int b = 0; b < BATCH; ++b) {
            ++sweeps;
            if (hflags[b] == 0) { done = true; break; }         // a sweep that moved nothing: the fixed point
        }
    }
    GEO_REQUIRE(done, "geo_sssp_nearest_source: no fixed point after %d sweeps", sweeps);
    status_out[1] = sweeps;
    GEO_HIP_CHECK(hipMemsetAsync(info, 0, 8, stream));
    voronoi_out_kernel<<<geo::grid_for(n, 256), 256, 0, stream>>>(key, n, std::ldexp(1.0, -shift), dmin_out, argmin_out, info, suspects);
    GEO_LAUNCH_CHECK();
    int32_t hinfo[2] = {0, 0};
    GEO_HIP_CHECK(hipMemcpyAsync(hinfo, info, 8, hipMemcpyDeviceToHost, stream));
    GEO_HIP_CHECK(hipStreamSynchronize(stream));
    status_out[2] = hinfo[1];
    if (hinfo[0]) { status_out[3] = 2; return GEO_OK; }
    if (hinfo[1] > NEAREST_MAX_SUSPECTS) { status_out[3] = 3; return GEO_OK; }
    if (hinfo[1] > 0 && argmin_out) {
        // d(suspect, .) rows: same weights, same exact sums, same float32 cast as the K x N matrix's column entries
        if (int rc = sssp_multi_impl(indptr, indices, weights, n, nnz, suspects, hinfo[1], Dsus, nullptr, nullptr, nullptr,
                                     multi_ws, multi_bytes_needed, nullptr, stream_, 0))
            return rc;
        voronoi_resolve_kernel<<<(unsigned)hinfo[1], 64, 0, stream>>>(Dsus, n, suspects, sources, n_sources, dmin_out, argmin_out);
        GEO_LAUNCH_CHECK();
        GEO_HIP_CHECK(hipStreamSynchronize(stream));
    }
    status_out[0] = 0;
    return GEO_OK;
}

extern "C" int geo_sssp_plan(int32_t n, int32_t n_sources) {
    if (n <= 0 || n_sources <= 0) return GEO_E_ARG;
    const int sb = choose_sb(n, n_sources, 0);
    const bool chunked = sb == 16 && n_sources > 16 && n < ROW128_MAX_NODES;
    const bool u32 = chunked && n_sources >= 32 && geo::options().sssp_u32 != 0;
    return sb + (chunked ? 1000 : 0) + (u32 ? 2000 : 0);
}

extern "C" int geo_sssp_last_profile(double *sweep_ms, int32_t *sweep_launches) {
    if (sweep_ms) *sweep_ms = g_last_sweep_ms;
    if (sweep_launches) *sweep_launches = g_last_sweep_launches;
    return g_last_layout;       // >= 0: sources per batch (+1000: 16-edge chunk work items), never an error
}

extern "C" int geo_sssp_single_update(const int32_t *indptr, const int32_t *indices, const float *weights, int32_t n,
                                      int32_t source, float *d_out, float *dmin_inout, int32_t *argmin_inout,
                                      int32_t center_pos, void *ws, size_t ws_bytes, int32_t *sweeps_out,
                                      void *stream_) {
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    GEO_REQUIRE(n > 0 && source >= 0 && source < n, "geo_sssp_single_update: bad n=%d source=%d", n, source);
    GEO_REQUIRE(indptr && indices && ws, "geo_sssp_single_update: null pointer");
    geo::Arena ar(ws, ws_bytes);
    double *d = ar.take<double>((size_t)n);
    int32_t *flags = ar.take<int32_t>(4);
    if (!d || !flags) {
        geo::set_error("geo_sssp_single_update: workspace too small");
        return GEO_E_WORKSPACE;
    }
    const int grid1 = geo::grid_for(n, 256, 2048);
    int32_t sweeps = 0;
    if (int rc = solve_single(indptr, indices, weights, n, source, d, flags, SWEEP_GROUP, stream, &sweeps)) return rc;
    if (sweeps_out) *sweeps_out = sweeps;
    finish_single_kernel<<<grid1, 256, 0, stream>>>(d, n, d_out, dmin_inout, argmin_inout, center_pos);
    GEO_LAUNCH_CHECK();
    GEO_HIP_CHECK(hipStreamSynchronize(stream));
    return GEO_OK;
}

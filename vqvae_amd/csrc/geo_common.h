// geo_common.h -- shared host-side helpers for libgeo_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>

#include "geo_hip.h"

namespace geo {

void set_error(const char *fmt, ...);

#define GEO_HIP_CHECK(expr)                                                                      \
    do {                                                                                         \
        hipError_t _e = (expr);                                                                  \
        if (_e != hipSuccess) {                                                                  \
            geo::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
            return GEO_E_HIP;                                                                    \
        }                                                                                        \
    } while (0)

#define GEO_REQUIRE(cond, ...)              \
    do {                                    \
        if (!(cond)) {                      \
            geo::set_error(__VA_ARGS__);    \
            return GEO_E_ARG;               \
        }                                   \
    } while (0)

#define GEO_LAUNCH_CHECK() GEO_HIP_CHECK(hipGetLastError())

// Experiment switches (DESIGN.md "Experiment switches").  Seeded ONCE from the environment when the library is
// loaded; afterwards only geo_set_option() changes them.  -1 = automatic / default behaviour.
struct Options {
    int sssp_sb = -1, sssp_act = 1, sssp_sparse_div = -1, sssp_map_div = -1, sssp_group = 1, sssp_grouped_cap = 256,
        sssp_trace = 0, sssp_u32 = 1, sssp_push = 1 /* near-far push solve for long geodesics; 2 = for every graph */,
        sssp_delta = 8 /* near-far bucket width in mean edge weights */, sssp_order = 0 /* sources ordered along 0: two exact landmark distances, 1: graph cells, 2: two landmark hop counts (1, 2: cheaper to compute, worse batches on the bench's swiss graph) */, sssp_push_blocks = 64, knn_filter = 1, kpp_grid = 256, kpp_profile = 0, jvp_mid = 0 /* 1 = f32 MFMA, 2 = per-chunk bf16 x 3 */,
        jvp_back_valu = 0, jvp_front_valu = 0, jvp_per_node = 1 /* fixed-statistics decoders: primal ConvT2 / ConvT3 once per latent */,
        jvp_pipe_grid = 1 /* workgroups of the persistent ConvT2 kernel per CU-count (its tiles are dealt round-robin; more, shorter workgroups fill in behind other builds' resident kernels) */,
        jvp_node_jacobian = 1 /* fixed statistics, d <= 16: decoder Jacobian once per latent, edge ends from its columns (1: when the graph has enough edges per latent, 2: always, 0: never) */;
};
Options &options();

static inline size_t align_up(size_t x, size_t a = 256) { return (x + a - 1) / a * a; }

// Bump allocator over the caller's workspace.
struct Arena {
    char *base;
    size_t size, off;
    Arena(void *p, size_t n) : base(static_cast<char *>(p)), size(n), off(0) {}
    template <typename T>
    T *take(size_t count) {
        size_t bytes = align_up(count * sizeof(T));
        if (off + bytes > size) return nullptr;
        T *r = reinterpret_cast<T *>(base + off);
        off += bytes;
        return r;
    }
};

// Exclusive prefix sum of int32 counts into int32 offsets (out[n] = total); returns total on host
// when total_host != nullptr (synchronises in that case).  tmp: >= scan_tmp_bytes(n).
size_t scan_tmp_bytes(int64_t n);
int exclusive_scan_i32(const int32_t *in, int32_t *out, int64_t n, void *tmp, size_t tmp_bytes,
                       int64_t *total_host, hipStream_t stream);

static inline int grid_for(int64_t work_items, int per_block, int cap = 8192) {
    int64_t g = (work_items + per_block - 1) / per_block;
    if (g < 1) g = 1;
    if (g > cap) g = cap;
    return static_cast<int>(g);
}

}  // namespace geo

// common.hip -- error reporting and the device prefix sum used by the graph kernels.
#include "geo_common.h"

#include <cstdlib>
#include <cstring>

namespace geo {

namespace {
struct OptionName { const char *name, *env; int Options::*field; };
const OptionName kOptionNames[] = {
    {"sssp_sb", "GEO_SSSP_SB", &Options::sssp_sb},
    {"sssp_act", "GEO_SSSP_ACT", &Options::sssp_act},
    {"sssp_sparse_div", "GEO_SSSP_SPARSE_DIV", &Options::sssp_sparse_div},
    {"sssp_map_div", "GEO_SSSP_MAP_DIV", &Options::sssp_map_div},
    {"sssp_group", "GEO_SSSP_GROUP", &Options::sssp_group},
    {"sssp_grouped_cap", "GEO_SSSP_GROUPED_CAP", &Options::sssp_grouped_cap},
    {"sssp_trace", "GEO_SSSP_TRACE", &Options::sssp_trace},
    {"sssp_u32", "GEO_SSSP_U32", &Options::sssp_u32},
    {"sssp_push", "GEO_SSSP_PUSH", &Options::sssp_push},
    {"sssp_order", "GEO_SSSP_ORDER", &Options::sssp_order},
    {"sssp_delta", "GEO_SSSP_DELTA", &Options::sssp_delta},
    {"sssp_push_blocks", "GEO_SSSP_PUSH_BLOCKS", &Options::sssp_push_blocks},
    {"knn_filter", "GEO_KNN_FILTER", &Options::knn_filter},
    {"kpp_grid", "GEO_KPP_GRID", &Options::kpp_grid},
    {"kpp_profile", "GEO_KPP_PROFILE", &Options::kpp_profile},
    {"jvp_mid", "GEO_JVP_MID", &Options::jvp_mid},
    {"jvp_back_valu", "GEO_JVP_BACK_VALU", &Options::jvp_back_valu},
    {"jvp_front_valu", "GEO_JVP_FRONT_VALU", &Options::jvp_front_valu},
    {"jvp_per_node", "GEO_JVP_PER_NODE", &Options::jvp_per_node},
    {"jvp_node_jacobian", "GEO_JVP_NODE_JACOBIAN", &Options::jvp_node_jacobian},
    {"jvp_pipe_grid", "GEO_JVP_PIPE_GRID", &Options::jvp_pipe_grid},
};
Options from_environment() {
    Options o;
    for (const OptionName &e : kOptionNames) {
        const char *v = getenv(e.env);
        if (!v) continue;
        if (e.field == &Options::jvp_mid) o.jvp_mid = v[0] == 'f' ? 1 : (v[0] == 'c' ? 2 : (v[0] == 'a' ? 3 : atoi(v)));
        else o.*(e.field) = atoi(v);
    }
    return o;
}
}  // namespace

Options &options() {
    static Options o = from_environment();       // the environment is read once, at the first call into the library
    return o;
}

static thread_local char g_err[512] = "";       // per host thread: concurrent calls (own stream, own workspace) do not mix messages

void set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

// ---------------------------------------------------------------------------------------------
// Exclusive scan: 256 threads x 8 items per block, block totals scanned recursively.
// ---------------------------------------------------------------------------------------------
constexpr int SCAN_THREADS = 256;
constexpr int SCAN_ITEMS = 8;
constexpr int SCAN_TILE = SCAN_THREADS * SCAN_ITEMS;

__global__ __launch_bounds__(SCAN_THREADS) void scan_tiles_kernel(const int32_t *__restrict__ in,
                                                                 int32_t *__restrict__ out,
                                                                 int32_t *__restrict__ tile_sums, int64_t n) {
    __shared__ int32_t wave_tot[SCAN_THREADS / 64];
    const int64_t base = (int64_t)blockIdx.x * SCAN_TILE + (int64_t)threadIdx.x * SCAN_ITEMS;
    int32_t v[SCAN_ITEMS];
    int32_t run = 0;
#pragma unroll
    for (int i = 0; i < SCAN_ITEMS; ++i) {
        int32_t x = (base + i < n) ? in[base + i] : 0;
        v[i] = run;
        run += x;
    }
    // inclusive scan of per-thread totals inside the wave
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int32_t inc = run;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        int32_t t = __shfl_up(inc, off, 64);
        if (lane >= off) inc += t;
    }
    if (lane == 63) wave_tot[wave] = inc;
    __syncthreads();
    int32_t wave_base = 0, total = 0;
#pragma unroll
    for (int w = 0; w < SCAN_THREADS / 64; ++w) {
        if (w < wave) wave_base += wave_tot[w];
        total += wave_tot[w];
    }
    const int32_t excl = wave_base + inc - run;
#pragma unroll
    for (int i = 0; i < SCAN_ITEMS; ++i)
        if (base + i < n) out[base + i] = excl + v[i];
    if (threadIdx.x == 0) tile_sums[blockIdx.x] = total;
}

__global__ __launch_bounds__(SCAN_THREADS) void scan_add_kernel(int32_t *__restrict__ out,
                                                               const int32_t *__restrict__ tile_offs, int64_t n,
                                                               int64_t n_tiles) {
    const int64_t base = (int64_t)blockIdx.x * SCAN_TILE + (int64_t)threadIdx.x * SCAN_ITEMS;
    const int32_t add = tile_offs[blockIdx.x];
#pragma unroll
    for (int i = 0; i < SCAN_ITEMS; ++i)
        if (base + i < n) out[base + i] += add;
    // out[n] = grand total (tile_offs has n_tiles + 1 entries)
    if (blockIdx.x == 0 && threadIdx.x == 0) out[n] = tile_offs[n_tiles];
}

__global__ void scan_single_total_kernel(int32_t *out, const int32_t *tile_sums, int64_t n) {
    if (threadIdx.x == 0 && blockIdx.x == 0) out[n] = tile_sums[0];
}

static int64_t tiles_of(int64_t n) { return (n + SCAN_TILE - 1) / SCAN_TILE; }

size_t scan_tmp_bytes(int64_t n) {
    size_t total = 0;
    int64_t m = n;
    while (true) {
        int64_t t = tiles_of(m > 0 ? m : 1);
        total += align_up((size_t)(t + 1) * sizeof(int32_t)) * 2;
        if (t <= 1) break;
        m = t;
    }
    return total + 256;
}

static int scan_rec(const int32_t *in, int32_t *out, int64_t n, Arena &ar, hipStream_t s) {
    const int64_t t = tiles_of(n > 0 ? n : 1);
    int32_t *sums = ar.take<int32_t>((size_t)t + 1);
    int32_t *offs = ar.take<int32_t>((size_t)t + 1);
    if (!sums || !offs) {
        set_error("exclusive_scan: workspace too small");
        return GEO_E_WORKSPACE;
    }
    scan_tiles_kernel<<<dim3((unsigned)t), dim3(SCAN_THREADS), 0, s>>>(in, out, sums, n);
    GEO_LAUNCH_CHECK();
    if (t == 1) {
        scan_single_total_kernel<<<1, 64, 0, s>>>(out, sums, n);
        GEO_LAUNCH_CHECK();
        return GEO_OK;
    }
    int rc = scan_rec(sums, offs, t, ar, s);
    if (rc != GEO_OK) return rc;
    scan_add_kernel<<<dim3((unsigned)t), dim3(SCAN_THREADS), 0, s>>>(out, offs, n, t);
    GEO_LAUNCH_CHECK();
    return GEO_OK;
}

int exclusive_scan_i32(const int32_t *in, int32_t *out, int64_t n, void *tmp, size_t tmp_bytes,
                       int64_t *total_host, hipStream_t stream) {
    Arena ar(tmp, tmp_bytes);
    int rc = scan_rec(in, out, n, ar, stream);
    if (rc != GEO_OK) return rc;
    if (total_host) {
        int32_t tot = 0;
        GEO_HIP_CHECK(hipMemcpyAsync(&tot, out + n, sizeof(int32_t), hipMemcpyDeviceToHost, stream));
        GEO_HIP_CHECK(hipStreamSynchronize(stream));
        *total_host = tot;
    }
    return GEO_OK;
}

}  // namespace geo

extern "C" int geo_version(void) { return 103; }   // 1.0.3: + geo_sssp_nearest_source, geo_jvp_edges_workspace_bytes, geo_pam_swap_deltas, prior kernels
extern "C" const char *geo_last_error(void) { return geo::g_err; }

extern "C" int geo_set_option(const char *name, int32_t value) {
    if (name)
        for (const geo::OptionName &e : geo::kOptionNames)
            if (strcmp(name, e.name) == 0) {
                geo::options().*(e.field) = value;
                return GEO_OK;
            }
    geo::set_error("geo_set_option: unknown option '%s'", name ? name : "(null)");
    return GEO_E_ARG;
}

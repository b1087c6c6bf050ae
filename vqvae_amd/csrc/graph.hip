// graph.hip -- kNN-list symmetrisation, upper-triangle edge list, connected components and CSR
// compaction on the GPU (gfx950).  All of it is integer/byte work bound by HBM / L2 traffic.
//
// Replaces, in the reference:
//   csr_matrix + W.maximum/minimum(W.T) + setdiag(0) + eliminate_zeros   src/geo/knn_graph_optimized.py:54-66
//   W.nonzero() / rows < cols, U + U^T                                   src/scripts/build_codebook.py:43-45,53-54
//   scipy.sparse.csgraph.connected_components                            src/geo/knn_graph_optimized.py:175,187
//   W[mask][:, mask]                                                     src/scripts/build_codebook.py:59
#include "geo_common.h"

namespace {

constexpr int WPB = 4;   // waves (= rows) per 256-thread block

// ------------------------------------------------------------------------------------ symmetrise
__global__ __launch_bounds__(256) void count_in_kernel(const int32_t *__restrict__ nbr, int64_t total, int32_t n,
                                                      int32_t *__restrict__ cnt_in) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int32_t j = nbr[i];
        if (j >= 0 && j < n) atomicAdd(&cnt_in[j], 1);
    }
}

__global__ __launch_bounds__(256) void fill_in_kernel(const int32_t *__restrict__ nbr, const float *__restrict__ w,
                                                     int64_t total, int32_t n, int32_t k,
                                                     const int32_t *__restrict__ off_in, int32_t *__restrict__ cursor,
                                                     int32_t *__restrict__ in_src, float *__restrict__ in_w) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int32_t j = nbr[i];
        if (j >= 0 && j < n) {
            const int32_t p = off_in[j] + atomicAdd(&cursor[j], 1);
            in_src[p] = (int32_t)(i / k);
            in_w[p] = w ? w[i] : 1.0f;
        }
    }
}

// weight of the directed entry a -> b in the kNN lists, 0 when absent (scipy: absent = 0)
__device__ __forceinline__ float directed_weight(const int32_t *__restrict__ nbr, const float *__restrict__ w,
                                                 int32_t k, int32_t a, int32_t b, bool *found) {
    const int32_t *row = nbr + (int64_t)a * k;
    for (int t = 0; t < k; ++t)
        if (row[t] == b) {
            *found = true;
            return w ? w[(int64_t)a * k + t] : 1.0f;
        }
    *found = false;
    return 0.0f;
}

// Element `t` of row r's candidate list: t < k -> out entry, else in entry.  Returns whether it is a
// kept primary of the symmetric matrix, with its column and value.
__device__ __forceinline__ bool sym_element(const int32_t *__restrict__ nbr, const float *__restrict__ w, int32_t n,
                                            int32_t k, int mode, const int32_t *__restrict__ off_in,
                                            const int32_t *__restrict__ in_src, const float *__restrict__ in_w,
                                            int32_t r, int32_t t, int32_t *col, float *val) {
    if (t < k) {
        const int32_t c = nbr[(int64_t)r * k + t];
        if (c < 0 || c >= n || c == r) return false;
        // a repeated column inside one list would be summed by scipy's COO->CSR; kNN lists have none
        const float wo = w ? w[(int64_t)r * k + t] : 1.0f;
        bool back;
        const float wb = directed_weight(nbr, w, k, c, r, &back);
        const float v = mode == 0 ? fmaxf(wo, wb) : fminf(wo, wb);
        *col = c;
        *val = v;
        return v != 0.0f;
    }
    if (mode != 0) return false;                       // mutual: an in-only entry has min(0, w) = 0
    const int32_t p = off_in[r] + (t - k);
    const int32_t c = in_src[p];
    if (c == r) return false;
    bool fwd;
    (void)directed_weight(nbr, w, k, r, c, &fwd);
    if (fwd) return false;                             // already produced by the out entry
    *col = c;
    *val = in_w[p];                                    // max(0, w)
    return *val != 0.0f;
}

__global__ __launch_bounds__(256) void sym_count_kernel(const int32_t *__restrict__ nbr, const float *__restrict__ w,
                                                       int32_t n, int32_t k, int mode,
                                                       const int32_t *__restrict__ off_in,
                                                       const int32_t *__restrict__ in_src,
                                                       const float *__restrict__ in_w, int32_t *__restrict__ row_cnt) {
    const int lane = threadIdx.x & 63;
    const int32_t r = blockIdx.x * WPB + (threadIdx.x >> 6);
    if (r >= n) return;
    const int32_t len = k + (off_in[r + 1] - off_in[r]);
    int32_t cnt = 0;
    for (int32_t t = lane; t < len; t += 64) {
        int32_t c;
        float v;
        cnt += sym_element(nbr, w, n, k, mode, off_in, in_src, in_w, r, t, &c, &v) ? 1 : 0;
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) cnt += __shfl_xor(cnt, off, 64);
    if (lane == 0) row_cnt[r] = cnt;
}

// pass A: kept primaries of row r, unsorted, into tmp at the row's final offsets
__global__ __launch_bounds__(256) void sym_scatter_kernel(const int32_t *__restrict__ nbr, const float *__restrict__ w,
                                                         int32_t n, int32_t k, int mode,
                                                         const int32_t *__restrict__ off_in,
                                                         const int32_t *__restrict__ in_src,
                                                         const float *__restrict__ in_w,
                                                         const int32_t *__restrict__ indptr,
                                                         int32_t *__restrict__ tmp_col, float *__restrict__ tmp_val) {
    const int lane = threadIdx.x & 63;
    const int32_t r = blockIdx.x * WPB + (threadIdx.x >> 6);
    if (r >= n) return;
    const int32_t len = k + (off_in[r + 1] - off_in[r]);
    int32_t base = indptr[r];
    for (int32_t t0 = 0; t0 < len; t0 += 64) {
        const int32_t t = t0 + lane;
        int32_t c = 0;
        float v = 0.f;
        const bool keep = t < len && sym_element(nbr, w, n, k, mode, off_in, in_src, in_w, r, t, &c, &v);
        const unsigned long long m = __ballot(keep);
        if (keep) {
            const int32_t p = base + __builtin_popcountll(m & ((1ull << lane) - 1ull));
            tmp_col[p] = c;
            tmp_val[p] = v;
        }
        base += __builtin_popcountll(m);
    }
}

// pass B: rank sort of every row segment by column (columns inside a row are distinct)
// rows of the symmetrised graph, sorted by column (rank by counting inside the row).  The work item is a 64-entry SEGMENT
// of a row, not a row: kNN graphs of high-dimensional clouds have hubs (in-degrees in the thousands at d = 64), and one
// wave ranking a 7 000-entry row alone took 4 ms of the 50 000 x 64 configuration's step.
__global__ __launch_bounds__(256) void seg_count_kernel(const int32_t *__restrict__ indptr, int32_t n, int32_t *__restrict__ seg_cnt) {
    for (int32_t r = blockIdx.x * blockDim.x + threadIdx.x; r < n; r += gridDim.x * blockDim.x)
        seg_cnt[r] = (indptr[r + 1] - indptr[r] + 63) >> 6;
}

__global__ __launch_bounds__(256) void sym_sort_kernel(int32_t n, const int32_t *__restrict__ indptr,
                                                      const int32_t *__restrict__ seg_off,
                                                      const int32_t *__restrict__ tmp_col,
                                                      const float *__restrict__ tmp_val, int32_t *__restrict__ indices,
                                                      float *__restrict__ data) {
    const int lane = threadIdx.x & 63;
    const int32_t w = blockIdx.x * WPB + (threadIdx.x >> 6);
    if (w >= seg_off[n]) return;
    int32_t lo = 0, hi = n;                                    // the row r with seg_off[r] <= w < seg_off[r + 1]
    while (hi - lo > 1) {
        const int32_t mid = (lo + hi) >> 1;
        if (seg_off[mid] <= w) lo = mid; else hi = mid;
    }
    const int32_t r = lo;
    const int32_t b = indptr[r], e = indptr[r + 1];
    const int32_t i = b + ((w - seg_off[r]) << 6) + lane;
    if (i < e) {
        const int32_t c = tmp_col[i];
        int32_t rank = 0;
        for (int32_t j = b; j < e; ++j) rank += tmp_col[j] < c ? 1 : 0;
        indices[b + rank] = c;
        data[b + rank] = tmp_val[i];
    }
}

struct SymWs {
    int32_t *cnt_in, *off_in, *cursor, *in_src, *row_cnt, *tmp_col;
    float *in_w, *tmp_val;
    void *scan_tmp;
    size_t scan_bytes;
};

bool carve_sym(void *ws, size_t ws_bytes, int32_t n, int32_t k, SymWs *o) {
    geo::Arena ar(ws, ws_bytes);
    const size_t nk = (size_t)n * k;
    o->cnt_in = ar.take<int32_t>((size_t)n + 1);
    o->off_in = ar.take<int32_t>((size_t)n + 1);
    o->cursor = ar.take<int32_t>((size_t)n + 1);
    o->row_cnt = ar.take<int32_t>((size_t)n + 1);
    o->in_src = ar.take<int32_t>(nk + 1);
    o->in_w = ar.take<float>(nk + 1);
    o->tmp_col = ar.take<int32_t>(2 * nk + 1);
    o->tmp_val = ar.take<float>(2 * nk + 1);
    o->scan_bytes = geo::scan_tmp_bytes((int64_t)n + 1);
    o->scan_tmp = ar.take<char>(o->scan_bytes);
    return o->scan_tmp != nullptr;
}

// ------------------------------------------------------------------------------------ upper edges
__device__ __forceinline__ int32_t first_greater(const int32_t *__restrict__ idx, int32_t b, int32_t e, int32_t key) {
    while (b < e) {                       // first position with idx[pos] > key (sorted row)
        const int32_t m = (b + e) >> 1;
        if (idx[m] > key) e = m; else b = m + 1;
    }
    return b;
}

__global__ __launch_bounds__(256) void upper_count_kernel(const int32_t *__restrict__ indptr,
                                                         const int32_t *__restrict__ indices, int32_t n,
                                                         int32_t *__restrict__ cnt) {
    for (int32_t r = blockIdx.x * blockDim.x + threadIdx.x; r < n; r += gridDim.x * blockDim.x)
        cnt[r] = indptr[r + 1] - first_greater(indices, indptr[r], indptr[r + 1], r);
}

__global__ __launch_bounds__(256) void upper_fill_kernel(const int32_t *__restrict__ indptr,
                                                        const int32_t *__restrict__ indices, int32_t n,
                                                        const int32_t *__restrict__ upper_ptr,
                                                        int32_t *__restrict__ src, int32_t *__restrict__ dst,
                                                        int32_t *__restrict__ entry_edge) {
    const int lane = threadIdx.x & 63;
    const int32_t r = blockIdx.x * WPB + (threadIdx.x >> 6);
    if (r >= n) return;
    const int32_t b = indptr[r], e = indptr[r + 1];
    const int32_t fu = e - (upper_ptr[r + 1] - upper_ptr[r]);        // first entry with col > r
    for (int32_t i = b + lane; i < e; i += 64) {
        const int32_t c = indices[i];
        int32_t eid = -1;
        if (c > r) {
            eid = upper_ptr[r] + (i - fu);
            src[eid] = r;
            dst[eid] = c;
        } else if (c < r) {
            const int32_t cb = indptr[c], ce = indptr[c + 1];
            const int32_t cfu = ce - (upper_ptr[c + 1] - upper_ptr[c]);
            const int32_t p = first_greater(indices, cfu, ce, r - 1);   // position of r in row c's upper part
            if (p < ce && indices[p] == r) eid = upper_ptr[c] + (p - cfu);
            (void)cb;
        }
        if (entry_edge) entry_edge[i] = eid;
    }
}

__global__ __launch_bounds__(256) void gather_weights_kernel(const float *__restrict__ len,
                                                            const int32_t *__restrict__ entry_edge, int64_t nnz,
                                                            float *__restrict__ out) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nnz; i += (int64_t)gridDim.x * blockDim.x) {
        const int32_t e = entry_edge[i];
        out[i] = e >= 0 ? len[e] : 0.0f;
    }
}

// ------------------------------------------------------------------------------------ components
__global__ __launch_bounds__(256) void cc_init_kernel(int32_t *__restrict__ label, int32_t n) {
    for (int32_t v = blockIdx.x * blockDim.x + threadIdx.x; v < n; v += gridDim.x * blockDim.x) label[v] = v;
}

// min-label hooking: a node takes the smallest label among itself and its neighbours and also
// pulls down the node its old label points to (so whole trees move at once).
__global__ __launch_bounds__(256) void cc_hook_kernel(const int32_t *__restrict__ indptr,
                                                     const int32_t *__restrict__ indices, int32_t n, int32_t *label,
                                                     int32_t *changed) {
    const int sub = threadIdx.x & 15;
    const int grp = (blockIdx.x * blockDim.x + threadIdx.x) >> 4;
    const int ngrp = (gridDim.x * blockDim.x) >> 4;
    bool any = false;
    for (int32_t v = grp; v < n; v += ngrp) {
        const int32_t old = label[v];
        int32_t m = old;
        for (int32_t e = indptr[v] + sub; e < indptr[v + 1]; e += 16) m = min(m, label[indices[e]]);
#pragma unroll
        for (int off = 8; off >= 1; off >>= 1) m = min(m, __shfl_xor(m, off, 16));
        if (sub == 0 && m < old) {
            atomicMin(&label[v], m);
            atomicMin(&label[old], m);
            any = true;
        }
    }
    if (any) *changed = 1;
}

__global__ __launch_bounds__(256) void cc_jump_kernel(int32_t *label, int32_t n) {
    for (int32_t v = blockIdx.x * blockDim.x + threadIdx.x; v < n; v += gridDim.x * blockDim.x) {
        int32_t l = label[v];
        int32_t ll = label[l];
        while (ll < l) { l = ll; ll = label[l]; }
        label[v] = l;
    }
}

__global__ __launch_bounds__(256) void cc_roots_kernel(const int32_t *__restrict__ label, int32_t n,
                                                      int32_t *__restrict__ is_root) {
    for (int32_t v = blockIdx.x * blockDim.x + threadIdx.x; v < n; v += gridDim.x * blockDim.x)
        is_root[v] = label[v] == v ? 1 : 0;
}

__global__ __launch_bounds__(256) void cc_number_kernel(const int32_t *__restrict__ label,
                                                       const int32_t *__restrict__ root_rank, int32_t n,
                                                       int32_t *__restrict__ out) {
    for (int32_t v = blockIdx.x * blockDim.x + threadIdx.x; v < n; v += gridDim.x * blockDim.x)
        out[v] = root_rank[label[v]];
}

// ------------------------------------------------------------------------------------ compaction
__global__ __launch_bounds__(256) void keep_flags_kernel(const uint8_t *__restrict__ keep, int32_t n,
                                                        int32_t *__restrict__ flag) {
    for (int32_t v = blockIdx.x * blockDim.x + threadIdx.x; v < n; v += gridDim.x * blockDim.x)
        flag[v] = (!keep || keep[v]) ? 1 : 0;
}

__global__ __launch_bounds__(256) void new_index_kernel(const int32_t *__restrict__ flag,
                                                       const int32_t *__restrict__ pos, int32_t n,
                                                       int32_t *__restrict__ new_index) {
    for (int32_t v = blockIdx.x * blockDim.x + threadIdx.x; v < n; v += gridDim.x * blockDim.x)
        new_index[v] = flag[v] ? pos[v] : -1;
}

__global__ __launch_bounds__(256) void compact_count_kernel(const int32_t *__restrict__ indptr,
                                                           const int32_t *__restrict__ indices,
                                                           const float *__restrict__ data, int32_t n,
                                                           const int32_t *__restrict__ new_index, int drop_zero,
                                                           int32_t *__restrict__ row_cnt_new) {
    const int lane = threadIdx.x & 63;
    const int32_t r = blockIdx.x * WPB + (threadIdx.x >> 6);
    if (r >= n) return;
    const int32_t nr = new_index[r];
    if (nr < 0) return;
    int32_t cnt = 0;
    for (int32_t i = indptr[r] + lane; i < indptr[r + 1]; i += 64)
        cnt += (new_index[indices[i]] >= 0 && !(drop_zero && data && data[i] == 0.0f)) ? 1 : 0;
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) cnt += __shfl_xor(cnt, off, 64);
    if (lane == 0) row_cnt_new[nr] = cnt;
}

__global__ __launch_bounds__(256) void compact_fill_kernel(const int32_t *__restrict__ indptr,
                                                          const int32_t *__restrict__ indices,
                                                          const float *__restrict__ data, int32_t n,
                                                          const int32_t *__restrict__ new_index, int drop_zero,
                                                          const int32_t *__restrict__ indptr_new,
                                                          int32_t *__restrict__ indices_out,
                                                          float *__restrict__ data_out) {
    const int lane = threadIdx.x & 63;
    const int32_t r = blockIdx.x * WPB + (threadIdx.x >> 6);
    if (r >= n) return;
    const int32_t nr = new_index[r];
    if (nr < 0) return;
    int32_t base = indptr_new[nr];
    const int32_t b = indptr[r], e = indptr[r + 1];
    for (int32_t i0 = b; i0 < e; i0 += 64) {
        const int32_t i = i0 + lane;
        int32_t nc = -1;
        bool keep = false;
        if (i < e) {
            nc = new_index[indices[i]];
            keep = nc >= 0 && !(drop_zero && data && data[i] == 0.0f);
        }
        const unsigned long long m = __ballot(keep);
        if (keep) {
            const int32_t p = base + __builtin_popcountll(m & ((1ull << lane) - 1ull));
            indices_out[p] = nc;
            if (data_out) data_out[p] = data ? data[i] : 1.0f;
        }
        base += __builtin_popcountll(m);
    }
}

unsigned rows_grid(int32_t n) { return (unsigned)((n + WPB - 1) / WPB); }

}  // namespace

// ============================================================================================ ABI
extern "C" size_t geo_symmetrize_workspace_bytes(int32_t n, int32_t k) {
    if (n <= 0 || k <= 0) return 1024;
    const size_t nk = (size_t)n * k;
    return 4 * geo::align_up(((size_t)n + 1) * 4) + 2 * geo::align_up((nk + 1) * 4) +
           2 * geo::align_up((2 * nk + 1) * 4) + geo::align_up(geo::scan_tmp_bytes((int64_t)n + 1)) + 4096;
}

extern "C" int geo_symmetrize_count(const int32_t *nbr_idx, const float *nbr_w, int32_t n, int32_t k, int32_t mode,
                                    int32_t *indptr_out, int64_t *nnz_out, void *ws, size_t ws_bytes, void *stream_) {
    hipStream_t s = static_cast<hipStream_t>(stream_);
    GEO_REQUIRE(nbr_idx && indptr_out && nnz_out && ws, "geo_symmetrize_count: null pointer");
    GEO_REQUIRE(n > 0 && k > 0 && (int64_t)n * k < ((int64_t)1 << 30), "geo_symmetrize_count: bad n=%d k=%d", n, k);
    GEO_REQUIRE(mode == 0 || mode == 1, "geo_symmetrize_count: mode must be 0 (union) or 1 (mutual)");
    SymWs w;
    if (!carve_sym(ws, ws_bytes, n, k, &w)) {
        geo::set_error("geo_symmetrize_count: workspace %zu too small", ws_bytes);
        return GEO_E_WORKSPACE;
    }
    const int64_t total = (int64_t)n * k;
    GEO_HIP_CHECK(hipMemsetAsync(w.cnt_in, 0, ((size_t)n + 1) * 4, s));
    GEO_HIP_CHECK(hipMemsetAsync(w.cursor, 0, ((size_t)n + 1) * 4, s));
    count_in_kernel<<<geo::grid_for(total, 256), 256, 0, s>>>(nbr_idx, total, n, w.cnt_in);
    GEO_LAUNCH_CHECK();
    int rc = geo::exclusive_scan_i32(w.cnt_in, w.off_in, n, w.scan_tmp, w.scan_bytes, nullptr, s);
    if (rc) return rc;
    fill_in_kernel<<<geo::grid_for(total, 256), 256, 0, s>>>(nbr_idx, nbr_w, total, n, k, w.off_in, w.cursor, w.in_src, w.in_w);
    GEO_LAUNCH_CHECK();
    sym_count_kernel<<<rows_grid(n), 256, 0, s>>>(nbr_idx, nbr_w, n, k, mode, w.off_in, w.in_src, w.in_w, w.row_cnt);
    GEO_LAUNCH_CHECK();
    return geo::exclusive_scan_i32(w.row_cnt, indptr_out, n, w.scan_tmp, w.scan_bytes, nnz_out, s);
}

extern "C" int geo_symmetrize_fill(const int32_t *nbr_idx, const float *nbr_w, int32_t n, int32_t k, int32_t mode,
                                   const int32_t *indptr, int32_t *indices_out, float *data_out, void *ws,
                                   size_t ws_bytes, void *stream_) {
    hipStream_t s = static_cast<hipStream_t>(stream_);
    GEO_REQUIRE(nbr_idx && indptr && indices_out && data_out && ws, "geo_symmetrize_fill: null pointer");
    SymWs w;
    if (!carve_sym(ws, ws_bytes, n, k, &w)) {
        geo::set_error("geo_symmetrize_fill: workspace %zu too small", ws_bytes);
        return GEO_E_WORKSPACE;
    }
    sym_scatter_kernel<<<rows_grid(n), 256, 0, s>>>(nbr_idx, nbr_w, n, k, mode, w.off_in, w.in_src, w.in_w, indptr,
                                                    w.tmp_col, w.tmp_val);
    GEO_LAUNCH_CHECK();
    // (cnt_in / off_in are free again after the scatter: segment counts and their offsets)
    seg_count_kernel<<<geo::grid_for(n, 256, 1024), 256, 0, s>>>(indptr, n, w.cnt_in);
    GEO_LAUNCH_CHECK();
    if (int rc = geo::exclusive_scan_i32(w.cnt_in, w.off_in, n, w.scan_tmp, w.scan_bytes, nullptr, s)) return rc;
    const int64_t max_segs = (int64_t)n + (2 * (int64_t)n * k) / 64 + 1;          // nnz <= 2 n k (union)
    sym_sort_kernel<<<(unsigned)((max_segs + WPB - 1) / WPB), 256, 0, s>>>(n, indptr, w.off_in, w.tmp_col, w.tmp_val,
                                                                          indices_out, data_out);
    GEO_LAUNCH_CHECK();
    return GEO_OK;
}

extern "C" int geo_upper_edges_count(const int32_t *indptr, const int32_t *indices, int32_t n, int32_t *upper_ptr_out,
                                     int64_t *n_edges_out, void *ws, size_t ws_bytes, void *stream_) {
    hipStream_t s = static_cast<hipStream_t>(stream_);
    GEO_REQUIRE(indptr && indices && upper_ptr_out && n_edges_out && ws && n > 0, "geo_upper_edges_count: bad argument");
    geo::Arena ar(ws, ws_bytes);
    int32_t *cnt = ar.take<int32_t>((size_t)n + 1);
    const size_t sb = geo::scan_tmp_bytes((int64_t)n + 1);
    void *st = ar.take<char>(sb);
    if (!cnt || !st) {
        geo::set_error("geo_upper_edges_count: workspace %zu too small", ws_bytes);
        return GEO_E_WORKSPACE;
    }
    upper_count_kernel<<<geo::grid_for(n, 256), 256, 0, s>>>(indptr, indices, n, cnt);
    GEO_LAUNCH_CHECK();
    return geo::exclusive_scan_i32(cnt, upper_ptr_out, n, st, sb, n_edges_out, s);
}

extern "C" int geo_upper_edges_fill(const int32_t *indptr, const int32_t *indices, int32_t n, const int32_t *upper_ptr,
                                    int32_t *src_out, int32_t *dst_out, int32_t *entry_edge_out, void *stream_) {
    hipStream_t s = static_cast<hipStream_t>(stream_);
    GEO_REQUIRE(indptr && indices && upper_ptr && src_out && dst_out && n > 0, "geo_upper_edges_fill: bad argument");
    upper_fill_kernel<<<rows_grid(n), 256, 0, s>>>(indptr, indices, n, upper_ptr, src_out, dst_out, entry_edge_out);
    GEO_LAUNCH_CHECK();
    return GEO_OK;
}

extern "C" int geo_gather_edge_weights(const float *len, const int32_t *entry_edge, int64_t nnz, float *data_out,
                                       void *stream_) {
    hipStream_t s = static_cast<hipStream_t>(stream_);
    GEO_REQUIRE(len && entry_edge && data_out && nnz >= 0, "geo_gather_edge_weights: bad argument");
    if (nnz == 0) return GEO_OK;
    gather_weights_kernel<<<geo::grid_for(nnz, 256), 256, 0, s>>>(len, entry_edge, nnz, data_out);
    GEO_LAUNCH_CHECK();
    return GEO_OK;
}

extern "C" size_t geo_cc_workspace_bytes(int32_t n) {
    if (n <= 0) return 1024;
    return 3 * geo::align_up(((size_t)n + 1) * 4) + geo::align_up(geo::scan_tmp_bytes((int64_t)n + 1)) + 4096;
}

extern "C" int geo_connected_components(const int32_t *indptr, const int32_t *indices, int32_t n, int32_t *labels_out,
                                        int32_t *n_components_out, void *ws, size_t ws_bytes, void *stream_) {
    hipStream_t s = static_cast<hipStream_t>(stream_);
    GEO_REQUIRE(indptr && indices && labels_out && n_components_out && ws && n > 0,
                "geo_connected_components: bad argument");
    geo::Arena ar(ws, ws_bytes);
    int32_t *label = ar.take<int32_t>((size_t)n + 1);
    int32_t *is_root = ar.take<int32_t>((size_t)n + 1);
    int32_t *rank = ar.take<int32_t>((size_t)n + 1);
    int32_t *changed = ar.take<int32_t>(64);
    const size_t sb = geo::scan_tmp_bytes((int64_t)n + 1);
    void *st = ar.take<char>(sb);
    if (!label || !is_root || !rank || !changed || !st) {
        geo::set_error("geo_connected_components: workspace %zu too small", ws_bytes);
        return GEO_E_WORKSPACE;
    }
    const int g = geo::grid_for(n, 256, 2048);
    const int gh = geo::grid_for(n, 16, 2048);
    cc_init_kernel<<<g, 256, 0, s>>>(label, n);
    GEO_LAUNCH_CHECK();
    int32_t hchanged = 1;
    int64_t rounds = 0;
    while (hchanged) {
        GEO_HIP_CHECK(hipMemsetAsync(changed, 0, sizeof(int32_t), s));
        cc_hook_kernel<<<gh, 256, 0, s>>>(indptr, indices, n, label, changed);
        GEO_LAUNCH_CHECK();
        cc_jump_kernel<<<g, 256, 0, s>>>(label, n);
        GEO_LAUNCH_CHECK();
        GEO_HIP_CHECK(hipMemcpyAsync(&hchanged, changed, sizeof(int32_t), hipMemcpyDeviceToHost, s));
        GEO_HIP_CHECK(hipStreamSynchronize(s));
        if (++rounds > (int64_t)n + 2) {
            geo::set_error("geo_connected_components: no fixed point after %lld rounds", (long long)rounds);
            return GEO_E_NOCONV;
        }
    }
    cc_roots_kernel<<<g, 256, 0, s>>>(label, n, is_root);
    GEO_LAUNCH_CHECK();
    int64_t ncomp = 0;
    int rc = geo::exclusive_scan_i32(is_root, rank, n, st, sb, &ncomp, s);
    if (rc) return rc;
    cc_number_kernel<<<g, 256, 0, s>>>(label, rank, n, labels_out);
    GEO_LAUNCH_CHECK();
    GEO_HIP_CHECK(hipStreamSynchronize(s));
    *n_components_out = (int32_t)ncomp;
    return GEO_OK;
}

extern "C" size_t geo_csr_compact_workspace_bytes(int32_t n) {
    if (n <= 0) return 1024;
    return 3 * geo::align_up(((size_t)n + 1) * 4) + geo::align_up(geo::scan_tmp_bytes((int64_t)n + 1)) + 4096;
}

extern "C" int geo_csr_compact_count(const int32_t *indptr, const int32_t *indices, const float *data, int32_t n,
                                     const uint8_t *keep_node, int32_t drop_zero, int32_t *new_index_out,
                                     int32_t *indptr_out, int32_t *n_out, int64_t *nnz_out, void *ws, size_t ws_bytes,
                                     void *stream_) {
    hipStream_t s = static_cast<hipStream_t>(stream_);
    GEO_REQUIRE(indptr && indices && new_index_out && indptr_out && n_out && nnz_out && ws && n > 0,
                "geo_csr_compact_count: bad argument");
    geo::Arena ar(ws, ws_bytes);
    int32_t *flag = ar.take<int32_t>((size_t)n + 1);
    int32_t *pos = ar.take<int32_t>((size_t)n + 1);
    int32_t *row_cnt = ar.take<int32_t>((size_t)n + 1);
    const size_t sb = geo::scan_tmp_bytes((int64_t)n + 1);
    void *st = ar.take<char>(sb);
    if (!flag || !pos || !row_cnt || !st) {
        geo::set_error("geo_csr_compact_count: workspace %zu too small", ws_bytes);
        return GEO_E_WORKSPACE;
    }
    const int g = geo::grid_for(n, 256, 2048);
    keep_flags_kernel<<<g, 256, 0, s>>>(keep_node, n, flag);
    GEO_LAUNCH_CHECK();
    int64_t n_new = 0;
    int rc = geo::exclusive_scan_i32(flag, pos, n, st, sb, &n_new, s);
    if (rc) return rc;
    new_index_kernel<<<g, 256, 0, s>>>(flag, pos, n, new_index_out);
    GEO_LAUNCH_CHECK();
    *n_out = (int32_t)n_new;
    if (n_new == 0) {
        *nnz_out = 0;
        GEO_HIP_CHECK(hipMemsetAsync(indptr_out, 0, sizeof(int32_t), s));
        return GEO_OK;
    }
    GEO_HIP_CHECK(hipMemsetAsync(row_cnt, 0, ((size_t)n + 1) * 4, s));
    compact_count_kernel<<<rows_grid(n), 256, 0, s>>>(indptr, indices, data, n, new_index_out, drop_zero, row_cnt);
    GEO_LAUNCH_CHECK();
    return geo::exclusive_scan_i32(row_cnt, indptr_out, n_new, st, sb, nnz_out, s);
}

extern "C" int geo_csr_compact_fill(const int32_t *indptr, const int32_t *indices, const float *data, int32_t n,
                                    const uint8_t *keep_node, int32_t drop_zero, const int32_t *new_index,
                                    const int32_t *indptr_new, int32_t *indices_out, float *data_out, void *stream_) {
    hipStream_t s = static_cast<hipStream_t>(stream_);
    (void)keep_node;
    GEO_REQUIRE(indptr && indices && new_index && indptr_new && indices_out && n > 0, "geo_csr_compact_fill: bad argument");
    compact_fill_kernel<<<rows_grid(n), 256, 0, s>>>(indptr, indices, data, n, new_index, drop_zero, indptr_new,
                                                     indices_out, data_out);
    GEO_LAUNCH_CHECK();
    return GEO_OK;
}

/*
 * oracle/geo_oracle.c -- CPU restatement of the numeric kernels of the geodesic-codebook path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under vqvae_amd/ may link, import or call this file.
 * It is the checker for the HIP path (tests/, __graft_entry__.smoke(), bench.py cpu_baseline).
 *
 * What each function restates (reference = /root/reference, third-party = the pinned versions
 * named in oracle/__init__.py):
 *
 *   oracle_sssp   scipy.sparse.csgraph.dijkstra as called from
 *                 src/geo/geo_shortest_paths.py:36-49 (fp64 path sums accumulated source-outward,
 *                 directed=False relaxes along csr and csr^T, unreachable = +inf,
 *                 predecessor sentinel -9999).  Binary heap instead of scipy's Fibonacci heap:
 *                 the distance fixed point is unique, so the distances are bit-identical.
 *   oracle_knn    sklearn NearestNeighbors(algorithm="auto").kneighbors as called from
 *                 src/geo/knn_graph_optimized.py:40-42: exact (k+1)-NN ranked on fp64 squared
 *                 distances.  form=1 is the brute-force expansion |x|^2 - 2 x.y + |y|^2 clamped
 *                 at 0 (sklearn _argkmin.pyx.tp:492-502, chosen by "auto" for d > 15), form=0 the
 *                 direct sum of squared differences (kd-tree rdist, chosen for d <= 15).
 *                 Ties are ordered (distance, index).
 *   oracle_cc     scipy.sparse.csgraph.connected_components(directed=False) as called from
 *                 src/geo/knn_graph_optimized.py:175,187: labels numbered in order of the lowest
 *                 node index of each component.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------ SSSP */

typedef struct { double key; int32_t node; } heap_item;

typedef struct {
    heap_item *a;
    int64_t size, cap;
} heap_t;

static int item_less(const heap_item *x, const heap_item *y) {
    if (x->key != y->key) return x->key < y->key;
    return x->node < y->node;
}

static int heap_push(heap_t *h, double key, int32_t node) {
    if (h->size == h->cap) {
        int64_t ncap = h->cap ? 2 * h->cap : 1024;
        heap_item *na = (heap_item *)realloc(h->a, (size_t)ncap * sizeof(heap_item));
        if (!na) return -1;
        h->a = na; h->cap = ncap;
    }
    int64_t i = h->size++;
    heap_item it = { key, node };
    while (i > 0) {
        int64_t p = (i - 1) >> 1;
        if (!item_less(&it, &h->a[p])) break;
        h->a[i] = h->a[p];
        i = p;
    }
    h->a[i] = it;
    return 0;
}

static heap_item heap_pop(heap_t *h) {
    heap_item top = h->a[0];
    heap_item last = h->a[--h->size];
    int64_t i = 0;
    for (;;) {
        int64_t l = 2 * i + 1, r = l + 1, m = i;
        const heap_item *best = &last;
        if (l < h->size && item_less(&h->a[l], best)) { m = l; best = &h->a[l]; }
        if (r < h->size && item_less(&h->a[r], best)) { m = r; best = &h->a[r]; }
        if (m == i) break;
        h->a[i] = h->a[m];
        i = m;
    }
    if (h->size > 0) h->a[i] = last;
    return top;
}

static void relax_row(const int32_t *indptr, const int32_t *indices, const double *w,
                      int32_t u, double du, double *dist, int32_t *pred, heap_t *h,
                      const uint8_t *done) {
    for (int32_t e = indptr[u]; e < indptr[u + 1]; ++e) {
        int32_t v = indices[e];
        if (done[v]) continue;
        double cand = du + w[e];           /* one fp64 rounding per hop, source outward */
        if (cand < dist[v]) {
            dist[v] = cand;
            if (pred) pred[v] = u;
            heap_push(h, cand, v);
        }
    }
}

/* D_out: [nsrc][n] fp64.  P_out may be NULL.  When directed == 0 the transpose arrays are also
 * relaxed (pass the same arrays twice for a symmetric matrix -- harmless). */
int oracle_sssp(int32_t n,
                const int32_t *indptr, const int32_t *indices, const double *w,
                const int32_t *indptrT, const int32_t *indicesT, const double *wT,
                int directed, int32_t nsrc, const int64_t *sources,
                double *D_out, int32_t *P_out) {
    uint8_t *done = (uint8_t *)malloc((size_t)(n > 0 ? n : 1));
    heap_t h = { 0, 0, 0 };
    if (!done) return -1;
    for (int32_t s = 0; s < nsrc; ++s) {
        double *dist = D_out + (size_t)s * n;
        int32_t *pred = P_out ? P_out + (size_t)s * n : 0;
        for (int32_t i = 0; i < n; ++i) { dist[i] = INFINITY; if (pred) pred[i] = -9999; }
        memset(done, 0, (size_t)n);
        int64_t src = sources[s];
        if (src < 0 || src >= n) { free(done); free(h.a); return -2; }
        dist[src] = 0.0;
        h.size = 0;
        heap_push(&h, 0.0, (int32_t)src);
        while (h.size > 0) {
            heap_item it = heap_pop(&h);
            int32_t u = it.node;
            if (done[u] || it.key > dist[u]) continue;   /* stale entry */
            done[u] = 1;
            relax_row(indptr, indices, w, u, it.key, dist, pred, &h, done);
            if (!directed)
                relax_row(indptrT, indicesT, wT, u, it.key, dist, pred, &h, done);
        }
    }
    free(done);
    free(h.a);
    return 0;
}

/* ------------------------------------------------------------------ kNN */

/* keeps the kq smallest (d2, idx) pairs of one row, sorted ascending, by insertion */
static void topk_insert(double *bd, int64_t *bi, int32_t kq, int32_t *cnt, double d2, int64_t j) {
    int32_t c = *cnt;
    if (c == kq) {
        if (d2 > bd[c - 1] || (d2 == bd[c - 1] && j > bi[c - 1])) return;
        c--;                               /* drop the current worst */
    }
    int32_t p = c;
    while (p > 0 && (bd[p - 1] > d2 || (bd[p - 1] == d2 && bi[p - 1] > j))) {
        bd[p] = bd[p - 1]; bi[p] = bi[p - 1]; --p;
    }
    bd[p] = d2; bi[p] = j;
    *cnt = c + 1;
}

/* z: [N][d] f32 (queries = corpus).  rows [row0,row1) are computed.  idx_out / d2_out: [row1-row0][kq] */
int oracle_knn(const float *z, int64_t N, int32_t d, int32_t kq, int32_t form,
               int64_t row0, int64_t row1, int64_t *idx_out, double *d2_out) {
    double *nrm = (double *)malloc((size_t)(N > 0 ? N : 1) * sizeof(double));
    if (!nrm) return -1;
    for (int64_t i = 0; i < N; ++i) {
        double s = 0.0;
        for (int32_t c = 0; c < d; ++c) { double x = z[i * d + c]; s = fma(x, x, s); }
        nrm[i] = s;
    }
#pragma omp parallel for schedule(dynamic, 16)
    for (int64_t i = row0; i < row1; ++i) {
        double *bd = d2_out + (size_t)(i - row0) * kq;
        int64_t *bi = idx_out + (size_t)(i - row0) * kq;
        int32_t cnt = 0;
        const float *xi = z + i * d;
        for (int64_t j = 0; j < N; ++j) {
            const float *yj = z + j * d;
            double d2;
            if (form) {
                double dot = 0.0;
                for (int32_t c = 0; c < d; ++c) dot = fma((double)xi[c], (double)yj[c], dot);
                d2 = (nrm[i] + (-2.0 * dot)) + nrm[j];
                if (!(d2 > 0.0)) d2 = 0.0;
            } else {
                d2 = 0.0;
                for (int32_t c = 0; c < d; ++c) {
                    double t = (double)xi[c] - (double)yj[c];
                    d2 = fma(t, t, d2);
                }
            }
            topk_insert(bd, bi, kq, &cnt, d2, j);
        }
    }
    free(nrm);
    return 0;
}

/* ------------------------------------------------------------------ connected components */

/* Undirected components of a structurally arbitrary CSR (edges used in both directions, so the
 * transpose must be supplied as well).  Returns the number of components. */
int32_t oracle_cc(int32_t n, const int32_t *indptr, const int32_t *indices,
                  const int32_t *indptrT, const int32_t *indicesT, int32_t *labels) {
    int32_t *stack = (int32_t *)malloc((size_t)(n > 0 ? n : 1) * sizeof(int32_t));
    if (!stack) return -1;
    for (int32_t i = 0; i < n; ++i) labels[i] = -1;
    int32_t ncomp = 0;
    for (int32_t root = 0; root < n; ++root) {
        if (labels[root] >= 0) continue;
        int32_t top = 0;
        stack[top++] = root;
        labels[root] = ncomp;
        while (top > 0) {
            int32_t u = stack[--top];
            for (int32_t e = indptr[u]; e < indptr[u + 1]; ++e) {
                int32_t v = indices[e];
                if (labels[v] < 0) { labels[v] = ncomp; stack[top++] = v; }
            }
            for (int32_t e = indptrT[u]; e < indptrT[u + 1]; ++e) {
                int32_t v = indicesT[e];
                if (labels[v] < 0) { labels[v] = ncomp; stack[top++] = v; }
            }
        }
        ++ncomp;
    }
    free(stack);
    return ncomp;
}

int oracle_version(void) { return 1; }

/*
 * geo_hip.h -- C ABI of libgeo_hip.so, the MI355X (gfx950) implementation of the geodesic-codebook
 * hot path of m4rch1n0/vqvae (src/geo + src/scripts/build_codebook.py).
 *
 * The reference has no FFI boundary of its own: its boundary is the Python API of src/geo, whose
 * numerics are executed by scipy / scikit-learn / torch.autograd on the CPU.  Each entry point below
 * replaces one of those third-party call sites (cited as reference file:line) and is what a binding
 * in the reference's src/geo modules would call (see INTEGRATION.md for the ctypes stubs).
 *
 * Conventions
 *   - every data pointer is a DEVICE pointer unless the parameter is marked [host];
 *   - `stream` is a hipStream_t passed as void*; work is enqueued on it.  Functions marked
 *     "synchronises" call hipStreamSynchronize(stream) before returning (they need a host decision);
 *   - no allocation crosses the ABI: the caller owns inputs, outputs and the workspace `ws`
 *     (size from the matching *_workspace_bytes query, any 256-byte aligned device buffer);
 *   - return value: 0 = ok, negative = error (GEO_E_*); geo_last_error() gives the text;
 *   - concurrency: calls from DIFFERENT host threads may run at the same time provided each uses its own stream, its own
 *     workspace and its own output buffers (bench.py pipelines two builds that way); geo_last_error's buffer, the sweep
 *     profile of geo_sssp_last_profile and its HIP events are per host thread.  The options of geo_set_option are
 *     process-global: change them only while no call is in flight.
 */
#ifndef GEO_HIP_H
#define GEO_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GEO_OK 0
#define GEO_E_ARG (-1)      /* invalid argument (shape, null pointer, unsupported size) */
#define GEO_E_WORKSPACE (-2) /* workspace too small */
#define GEO_E_HIP (-3)      /* a HIP runtime call failed */
#define GEO_E_NOCONV (-4)   /* iteration limit reached without convergence */

int geo_version(void);
const char *geo_last_error(void);

/* Experiment switches ("sssp_group", "knn_filter", "kpp_profile", ...: the table in DESIGN.md).  They are seeded from
 * the GEO_* environment variables ONCE, when the library is first called; afterwards only this call changes them.
 * Process-wide and not synchronised (like the rest of the library: one call at a time).  GEO_E_ARG: unknown name. */
int geo_set_option(const char *name, int32_t value);

/* ------------------------------------------------------------------------------------------
 * Shortest paths.  Replaces scipy.sparse.csgraph.dijkstra as called from
 * src/geo/geo_shortest_paths.py:36-49 (dijkstra_multi_source) and, through it,
 * src/geo/kmeans_optimized.py:43,97,125.
 *
 * The graph is a "pull" CSR: row v lists the nodes u with an edge u->v and its weight.  For an
 * undirected solve on a symmetric matrix that is the matrix itself.  Path sums are accumulated in
 * fp64 source-outward, one rounding per hop, and the label-correcting iteration is run to its fixed
 * point, so the fp64 distances equal Dijkstra's; they are rounded to f32 on output exactly like
 * geo_shortest_paths.py:50.  weights == NULL means unit weights (unweighted=True, :32-34).
 * Unreachable = +inf; predecessor sentinel -9999.
 * ------------------------------------------------------------------------------------------ */
size_t geo_sssp_workspace_bytes(int32_t n, int64_t nnz, int32_t n_sources);

/* S sources -> any of: D_out f32 [S][n]; P_out i32 [S][n]; dmin_out f32 [n] + argmin_out i32 [n]
 * (column minimum over the S rows of the f32 matrix and the FIRST row index attaining it,
 * = D.argmin(axis=0) of kmeans_optimized.py:100; all-inf column -> 0).  Any output may be NULL.
 * sweeps_out [host, may be NULL] receives the number of relaxation sweeps launched.
 * Internally the sources may be relaxed in another order than given (batches of neighbouring sources on graphs
 * with long geodesics); every output is in the caller's order and does not depend on that.
 * Synchronises. */
int geo_sssp_multi(const int32_t *indptr, const int32_t *indices, const float *weights,
                   int32_t n, int64_t nnz, const int32_t *sources, int32_t n_sources,
                   float *D_out, int32_t *P_out, float *dmin_out, int32_t *argmin_out,
                   void *ws, size_t ws_bytes, int32_t *sweeps_out, void *stream);

/* Nearest source per node in ONE label-carrying solve (what assign_points_to_medoids, kmeans_optimized.py:77-106, keeps of the
 * K x N matrix): dmin_out[v] = min_s D[s][v] and argmin_out[v] = D.argmin(axis=0)[v] of the FLOAT32 matrix dijkstra_multi_source
 * returns (geo_shortest_paths.py:50 casts before kmeans_optimized.py:100 compares): the lowest source row whose distance ROUNDS
 * to the column's float32 minimum -- not necessarily the exactly nearest one.  An unreachable node gets (+inf, 0).
 * The CSR must be symmetric (undirected graph, what the reference's callers pass): nodes whose two nearest sources round to the
 * same float32 ("suspects") are resolved by a solve from those nodes read at the sources.
 * status_out (host, int32 [4]): [0] = 0 answered / 1 declined (nothing usable written: call geo_sssp_multi with dmin_out /
 * argmin_out instead), [1] = sweeps, [2] = suspect nodes found, [3] = reason when declined (1 weights outside 28 bits of their
 * common power-of-two unit, negative or non-finite; 2 a distance reached 2^39 units; 3 more than 32 suspects; 4 n_sources >= 2^24 - 1).
 * Either output may be NULL.  Synchronises. */
size_t geo_sssp_nearest_workspace_bytes(int32_t n, int64_t nnz);
int geo_sssp_nearest_source(const int32_t *indptr, const int32_t *indices, const float *weights, int32_t n, int64_t nnz,
                            const int32_t *sources, int32_t n_sources, float *dmin_out, int32_t *argmin_out, void *ws,
                            size_t ws_bytes, int32_t *status_out, void *stream);

/* Device time (ms, HIP events on the call's stream) spent in the relaxation sweeps of the last
 * geo_sssp_multi call, and how many sweep kernels it launched.  Used by bench.py's roofline.
 * Returns the layout that call used: sources per batch (16 or 64), +1000 for the chunked 16-source kernel,
 * 2032 for the exact 32-bit fixed-point kernel (32 sources per row), 4016 for the near-far push solve
 * (long geodesics: sources ordered along landmark distances, 16 per batch, delta-stepping buckets; one launch per sweep). */
int geo_sssp_last_profile(double *sweep_ms, int32_t *sweep_launches);

/* The layout geo_sssp_multi would START with for a graph of n nodes and n_sources sources (host arithmetic only, no GPU
 * call): sources per batch (16 or 64), +1000 when the 16-edge-chunk kernels run, +2000 more when the exact 32-bit
 * fixed-point solve is tried first.  The 16- / 32-source row layouts address a batch with 32-bit byte offsets and are
 * therefore never chosen for n >= 2^25 nodes.  Negative on bad arguments. */
int geo_sssp_plan(int32_t n, int32_t n_sources);

/* One source; fused k-means++ bookkeeping of kmeans_optimized.py:43-44 and the single-pass
 * assignment: d32 = f32(dist(source, .)); where d32 < dmin: dmin = d32, argmin = center_pos.
 * d_out f32 [n] may be NULL, dmin_inout/argmin_inout may be NULL.  Synchronises. */
int geo_sssp_single_update(const int32_t *indptr, const int32_t *indices, const float *weights,
                           int32_t n, int32_t source, float *d_out,
                           float *dmin_inout, int32_t *argmin_inout, int32_t center_pos,
                           void *ws, size_t ws_bytes, int32_t *sweeps_out, void *stream);

/* ------------------------------------------------------------------------------------------
 * k-means++ seeding chain, device resident.  Replaces the loop of src/geo/kmeans_optimized.py:40-71:
 * iteration t in [it0, it1) solves from centers[t], folds the f32 distances into dmin/argmin (position t)
 * and, when t+1 < n_centers_total, draws centers[t+1] with numpy's legacy RandomState.choice semantics
 * (float32 D^2 weights, float32 add.reduce, fp64 cdf, searchsorted) from the uniform deviate u_host[t]
 * the caller took from the same RandomState stream.  centers i32 [n_centers_total] (centers[it0] set by
 * the caller), is_center u8 [n] (set for centers[0..it0]), dmin f32 [n] / argmin i32 [n] carried state.
 * sweeps_per_solve relaxation sweeps are enqueued per solve (they exit early once converged); 0 (needs
 * assume_finite) runs the chain as one step kernel launched until done: no budget, reason 1 only past 4094 sweeps;
 * -1 (needs assume_finite and n <= geo_kpp_resident_max_nodes()) runs iterations [it0, it1) inside ONE resident
 * workgroup in a single launch (solve in an LDS hash table, incremental float32 reduction tree, draw): the mode for
 * small cells; a centre whose cell outgrows the table is run by the step kernel inside the call.
 * assume_finite != 0 promises that d_min has no inf entry left (status_out[2] of an earlier call): the
 * per-iteration maximum pass is skipped.
 * status_out [host, 4 ints]: {abort_iter or -1, reason, inf entries of d_min at the last maximum pass,
 * most sweeps any solve of this call needed (step kernel: launches that did work)}:
 * reason 1 = solve not converged (nothing of that iteration is applied), 2 = u too close to a cdf boundary,
 * 3 = degenerate weights (for 2 and 3 the solve of that iteration IS applied, the draw is not).  The caller
 * repeats that step another way and resumes.  Resident mode: status_out[3] = centres handed to the step kernel.
 * One synchronisation at the end.
 * ------------------------------------------------------------------------------------------ */
size_t geo_kpp_workspace_bytes(int32_t n);
int32_t geo_kpp_resident_max_nodes(void);
int geo_kpp_chain(const int32_t *indptr, const int32_t *indices, const float *weights, int32_t n,
                  int32_t *centers, uint8_t *is_center, float *dmin, int32_t *argmin, const double *u_host,
                  int32_t it0, int32_t it1, int32_t n_centers_total, int32_t sweeps_per_solve,
                  int32_t assume_finite,
                  void *ws, size_t ws_bytes, int32_t *status_out, void *stream);

/* ------------------------------------------------------------------------------------------
 * Medoid update over a resident all-pairs matrix (extension: the reference stops after seeding + assignment,
 * src/geo/kmeans_optimized.py:141-183).  D f32 [n][ld] holds geodesic distances (rows filled by
 * geo_sssp_multi).  geo_cluster_costs: cost_out[i] = sum over the members j of i's cluster of D[i][j]^power
 * (power 1 or 2, fp64; members of cluster c = order[offsets[c] .. offsets[c+1]), `assign` i32 [n]).
 * geo_rows_argmin: for every column j the smallest D[rows[m]][j] and the first m attaining it (np.argmin's tie rule),
 * i.e. the re-assignment to the medoids `rows`.
 * ------------------------------------------------------------------------------------------ */
int geo_cluster_costs(const float *D, int64_t ld, const int32_t *assign, const int32_t *order,
                      const int32_t *offsets, int32_t n, int32_t power, double *cost_out, void *stream);
int geo_rows_argmin(const float *D, int64_t ld, const int32_t *rows, int32_t n_rows, int32_t n,
                    float *dmin_out, int32_t *argmin_out, void *stream);
/* geo_pam_swap_deltas: PAM's SWAP evaluation over the resident matrix (extension; FastPAM1 form).  For every
 * non-medoid candidate x: the medoid whose replacement by x lowers the total cost sum_j D[nearest(j)][j]^power most (first
 * medoid on ties) and that change.  nearest i32 [n]: position (0..K-1) of every node's nearest medoid; d1 / d2 f32 [n]:
 * distance to its nearest / second-nearest medoid; base f64 [K]: sum over the nodes of medoid i of (d2^power - d1^power);
 * is_medoid u8 [n].  K >= 2.  best_delta_out f64 [n] (+inf for medoids),
 * best_medoid_out i32 [n] (position in 0..K-1).  Reads D exactly once, row by row (n^2 * 4 bytes): HBM-bound.  K <= 3584. */
int geo_pam_swap_deltas(const float *D, int64_t ld, const int32_t *nearest, const float *d1, const float *d2,
                        const double *base, const uint8_t *is_medoid, int32_t n, int32_t K, int32_t power, double *best_delta_out,
                        int32_t *best_medoid_out, void *stream);
/* geo_attach_argmin: geodesic assignment of points outside the graph (the step the reference's notes call
 * assign_codes_val_geodesic.py, docs/results/cifar10_quantization_analysis.md:147; not in its repository).  Point v is
 * joined to graph nodes nbr[v][0..k) by edges of length len[v][0..k) (nbr < 0: no edge); Dt f32 [n][ld] holds the medoids'
 * distance rows transposed ([node][medoid]).  dist_out[v] = min over medoids m and edges u of len[v][u] + Dt[nbr[v][u]][m],
 * arg_out[v] = the first medoid attaining it (0 if none is reachable). */
int geo_attach_argmin(const float *Dt, int64_t ld, int32_t K, const int32_t *nbr, const float *len, int32_t k,
                      int64_t n_new, float *dist_out, int32_t *arg_out, void *stream);

/* ------------------------------------------------------------------------------------------
 * kNN search.  Replaces sklearn NearestNeighbors.kneighbors as called from
 * src/geo/knn_graph_optimized.py:40-42: exact n_neighbors nearest corpus rows (self included) of the
 * query rows [row0,row1) of z, ranked on fp64 squared distances, ties ordered by index.
 * form = 1: |x|^2 - 2 x.y + |y|^2 clamped at 0 (sklearn brute force, d > 15); form = 0: sum of squared
 * differences (sklearn kd-tree, d <= 15).  idx_out i32 [rows][n_neighbors], d2_out f64 likewise,
 * both sorted ascending.  n_neighbors <= GEO_KNN_MAX_NEIGHBORS (lists of up to 64 entries live one
 * per lane of the query's wave, longer ones two or four per lane; the float32 pre-filter serves
 * lists <= 64), d <= 128.
 * ------------------------------------------------------------------------------------------ */
#define GEO_KNN_MAX_NEIGHBORS 256
size_t geo_knn_workspace_bytes(int64_t n, int32_t d);
int geo_knn_topk(const float *z, int64_t n, int32_t d, int32_t n_neighbors, int32_t form,
                 int64_t row0, int64_t row1, int32_t *idx_out, double *d2_out,
                 void *ws, size_t ws_bytes, void *stream);

/* ------------------------------------------------------------------------------------------
 * Symmetrisation.  Replaces csr_matrix(...) + W.maximum/minimum(W.T) + setdiag(0) +
 * eliminate_zeros() of src/geo/knn_graph_optimized.py:54-66.
 * nbr_idx i32 [n][k] / nbr_w f32 [n][k] (NULL = connectivity, all ones) are the directed lists.
 * mode 0 = union (max), 1 = mutual (min).  Output is canonical CSR (columns ascending, no diagonal,
 * no stored zeros).  Two calls: geo_symmetrize_count fills indptr_out [n+1] and returns nnz through
 * nnz_out [host] (synchronises); the caller allocates indices/data and calls geo_symmetrize_fill
 * with the same workspace (its contents carry over).
 * ------------------------------------------------------------------------------------------ */
size_t geo_symmetrize_workspace_bytes(int32_t n, int32_t k);
int geo_symmetrize_count(const int32_t *nbr_idx, const float *nbr_w, int32_t n, int32_t k, int32_t mode,
                         int32_t *indptr_out, int64_t *nnz_out, void *ws, size_t ws_bytes, void *stream);
int geo_symmetrize_fill(const int32_t *nbr_idx, const float *nbr_w, int32_t n, int32_t k, int32_t mode,
                        const int32_t *indptr, int32_t *indices_out, float *data_out,
                        void *ws, size_t ws_bytes, void *stream);

/* ------------------------------------------------------------------------------------------
 * Upper-triangle edge list in row-major order.  Replaces W.nonzero() + rows<cols of
 * src/scripts/build_codebook.py:43-45.  upper_ptr_out i32 [n+1] = exclusive count of entries with
 * col > row; n_edges through [host] pointer (synchronises).  geo_upper_edges_fill writes
 * src/dst i32 [E] and entry_edge i32 [nnz]: for every stored entry the index of its undirected
 * edge (so that W_geo = U + U^T of build_codebook.py:53-54 is a gather).
 * ------------------------------------------------------------------------------------------ */
int geo_upper_edges_count(const int32_t *indptr, const int32_t *indices, int32_t n,
                          int32_t *upper_ptr_out, int64_t *n_edges_out, void *ws, size_t ws_bytes, void *stream);
int geo_upper_edges_fill(const int32_t *indptr, const int32_t *indices, int32_t n, const int32_t *upper_ptr,
                         int32_t *src_out, int32_t *dst_out, int32_t *entry_edge_out, void *stream);

/* ------------------------------------------------------------------------------------------
 * Connected components.  Replaces scipy.sparse.csgraph.connected_components(directed=False) of
 * src/geo/knn_graph_optimized.py:175,187 for a structurally symmetric CSR.
 * labels_out i32 [n]: component numbers in order of each component's lowest node (scipy's order).
 * n_components through [host] pointer.  Synchronises.
 * ------------------------------------------------------------------------------------------ */
size_t geo_cc_workspace_bytes(int32_t n);
int geo_connected_components(const int32_t *indptr, const int32_t *indices, int32_t n,
                             int32_t *labels_out, int32_t *n_components_out,
                             void *ws, size_t ws_bytes, void *stream);

/* ------------------------------------------------------------------------------------------
 * CSR filtering / compaction.  Replaces the scipy.sparse arithmetic of build_codebook.py:53-54
 * (entries that sum to exactly zero vanish) and the LCC sub-matrix W[mask][:, mask] (:59).
 * keep_node u8 [n] (NULL = all) selects rows/columns, entries with weight == 0 are dropped when
 * drop_zero != 0.  new_index_out i32 [n] = position of each kept node (-1 otherwise).
 * Two calls like symmetrize (count synchronises).
 * ------------------------------------------------------------------------------------------ */
size_t geo_csr_compact_workspace_bytes(int32_t n);
int geo_csr_compact_count(const int32_t *indptr, const int32_t *indices, const float *data, int32_t n,
                          const uint8_t *keep_node, int32_t drop_zero, int32_t *new_index_out,
                          int32_t *indptr_out, int32_t *n_out, int64_t *nnz_out,
                          void *ws, size_t ws_bytes, void *stream);
int geo_csr_compact_fill(const int32_t *indptr, const int32_t *indices, const float *data, int32_t n,
                         const uint8_t *keep_node, int32_t drop_zero, const int32_t *new_index,
                         const int32_t *indptr_new, int32_t *indices_out, float *data_out, void *stream);

/* ------------------------------------------------------------------------------------------
 * Decoder pull-back edge lengths.  Replaces torch.autograd.functional.jvp through SpatialDecoder
 * (src/geo/riemannian_metric.py:12-35,37-66; src/models/spatial_vae.py:47-81):
 *   len[e] = 0.5 * ( |J(z[src[e]]) dz| + |J(z[dst[e]]) dz| ),  dz = z[dst[e]] - z[src[e]],
 * J = Jacobian of sigmoid(decoder(.)) at a 1x1 latent, edges processed in chunks of `batch_size`
 * consecutive edges (each endpoint side of a chunk is one BatchNorm batch when bn_train != 0).
 * ------------------------------------------------------------------------------------------ */
typedef struct geo_decoder_desc {
    int32_t latent_dim;        /* d */
    int32_t c0, c1, c2;        /* dec_channels */
    int32_t out_channels;      /* image channels */
    int32_t out_size;          /* 28 or 32 */
    int32_t norm;              /* 0 none, 1 batch, 2 group */
    int32_t bn_train;          /* batch statistics (decoder.training) */
    int32_t groups1, groups2;  /* GroupNorm group counts */
    float eps;
    /* device pointers, f32, torch layouts */
    const float *w_in, *b_in;              /* conv_in.weight [c0][d], bias [c0] */
    const float *w1, *b1;                  /* deconv_layers.0 weight [c0][c1][4][4], bias [c1] */
    const float *g1, *be1, *rm1, *rv1;     /* deconv_layers.1 (norm) weight, bias, running_mean, running_var */
    const float *w2, *b2;                  /* deconv_layers.3 weight [c1][c2][4][4], bias [c2] */
    const float *g2, *be2, *rm2, *rv2;     /* deconv_layers.4 */
    const float *w3, *b3;                  /* deconv_layers.6 weight [c2][out_channels][4][4], bias */
    /* train-mode BatchNorm also folds every batch into its running statistics (torch: momentum 0.1), once per decoder
     * call of riemannian_metric.py:57-58, i.e. per (chunk, start | end side) in order.  update_running != 0 (with
     * norm == 1, bn_train != 0 and the four rm / rv pointers non-null) does the same IN PLACE on rm1, rv1, rm2, rv2
     * (unbiased batch variance); num_batches_tracked (+2 per chunk) is the caller's to advance. */
    int32_t update_running;
    float momentum;
} geo_decoder_desc;

size_t geo_jvp_workspace_bytes(const geo_decoder_desc *dec, int64_t n_edges, int32_t batch_size);
/* Workspace of geo_decoder_jvp_edges including the per-latent buffers (primal ConvT2 output and output sigmoids of every
 * latent) that decoders with fixed statistics use: the primal pass then runs once per latent instead of once per edge end
 * (what riemannian_metric.py:57-58 recomputes for every edge).  With fixed statistics, latent_dim <= 16 and at least
 * 0.75 * latent_dim edges per latent the call goes one step further and computes the decoder Jacobian once per latent (its
 * latent_dim columns, n_nodes * latent_dim * 32 or 192 floats of this workspace), each edge end from those columns: same
 * quantity, another summation order (within 1e-6 of the per-edge-end lengths; option jvp_node_jacobian = 0 turns it off).
 * A workspace of geo_jvp_workspace_bytes() still works: the call then takes the per-edge-end path. */
size_t geo_jvp_edges_workspace_bytes(const geo_decoder_desc *dec, int64_t n_nodes, int64_t n_edges, int32_t batch_size);
int geo_decoder_jvp_edges(const geo_decoder_desc *dec, const float *z, int64_t n_nodes,
                          const int32_t *src, const int32_t *dst, int64_t n_edges, int32_t batch_size,
                          float *len_out, void *ws, size_t ws_bytes, void *stream);

/* Same quantity for explicit endpoint arrays (edge_lengths_riemannian's own signature):
 * z_start / z_end f32 [E][d]. */
int geo_decoder_jvp_pairs(const geo_decoder_desc *dec, const float *z_start, const float *z_end,
                          int64_t n_edges, int32_t batch_size, float *len_out,
                          void *ws, size_t ws_bytes, void *stream);

/* Gather: data_out[e] = len[entry_edge[e]] for every stored entry (W_geo = U + U^T). */
int geo_gather_edge_weights(const float *len, const int32_t *entry_edge, int64_t nnz, float *data_out, void *stream);

/* ---- the code prior's training step (src/models/transformer.py:98-133, src/scripts/train_transformer.py:39-66) ----
 * Fused causal multi-head attention for sequences of at most 16 tokens.  qkv f32 [B][T][3][H][head_dim] (the c_attn
 * projection's output), head_dim in {16, 32, 64}; out f32 [B][T][H*head_dim]; probs f32 [B][H][T][T] receives the softmax
 * rows (kept for the backward pass); keep u8 [B][H][T][T] (1 = kept) or NULL applies dropout to them with
 * keep_scale = 1 / (1 - p).  Replaces q @ k^T * scale -> masked_fill(-inf) -> softmax -> dropout -> @ v
 * (transformer.py:121-129).  Asynchronous on `stream`. */
int geo_prior_attention_fwd(const float *qkv, const uint8_t *keep, float keep_scale, int32_t B, int32_t T, int32_t H,
                            int32_t head_dim, float *out, float *probs, void *stream);
/* Gradient of the above with respect to qkv: dqkv f32 [B][T][3][H][head_dim] from dout f32 [B][T][H*head_dim]. */
int geo_prior_attention_bwd(const float *qkv, const float *probs, const uint8_t *keep, float keep_scale, const float *dout,
                            int32_t B, int32_t T, int32_t H, int32_t head_dim, float *dqkv, void *stream);
/* torch.optim.AdamW's update (train_transformer.py:39-43,64-66: decoupled weight decay, bias-corrected moments) over ONE
 * flat buffer of n floats in a single pass.  lr_dev f32 [1] and step_dev i64 [1] (the 1-based step count of THIS update) are
 * read on the device, so the launch does not change from step to step.  Asynchronous on `stream`. */
int geo_prior_adamw(float *param, const float *grad, float *exp_avg, float *exp_avg_sq, int64_t n, const float *lr_dev,
                    const int64_t *step_dev, float beta1, float beta2, float eps, float weight_decay, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* GEO_HIP_H */

// Temporary: entry points not implemented yet report an error (replaced file by file).
#include "geo_common.h"
#define NOT_YET(name) geo::set_error(name ": not implemented"); return GEO_E_ARG
extern "C" {
size_t geo_knn_workspace_bytes(int64_t, int32_t) { return 0; }
int geo_knn_topk(const float *, int64_t, int32_t, int32_t, int32_t, int64_t, int64_t, int32_t *, double *, void *, size_t, void *) { NOT_YET("geo_knn_topk"); }
size_t geo_symmetrize_workspace_bytes(int32_t, int32_t) { return 0; }
int geo_symmetrize_count(const int32_t *, const float *, int32_t, int32_t, int32_t, int32_t *, int64_t *, void *, size_t, void *) { NOT_YET("geo_symmetrize_count"); }
int geo_symmetrize_fill(const int32_t *, const float *, int32_t, int32_t, int32_t, const int32_t *, int32_t *, float *, void *, size_t, void *) { NOT_YET("geo_symmetrize_fill"); }
int geo_upper_edges_count(const int32_t *, const int32_t *, int32_t, int32_t *, int64_t *, void *, size_t, void *) { NOT_YET("geo_upper_edges_count"); }
int geo_upper_edges_fill(const int32_t *, const int32_t *, int32_t, const int32_t *, int32_t *, int32_t *, int32_t *, void *) { NOT_YET("geo_upper_edges_fill"); }
size_t geo_cc_workspace_bytes(int32_t) { return 0; }
int geo_connected_components(const int32_t *, const int32_t *, int32_t, int32_t *, int32_t *, void *, size_t, void *) { NOT_YET("geo_connected_components"); }
size_t geo_csr_compact_workspace_bytes(int32_t) { return 0; }
int geo_csr_compact_count(const int32_t *, const int32_t *, const float *, int32_t, const uint8_t *, int32_t, int32_t *, int32_t *, int32_t *, int64_t *, void *, size_t, void *) { NOT_YET("geo_csr_compact_count"); }
int geo_csr_compact_fill(const int32_t *, const int32_t *, const float *, int32_t, const uint8_t *, int32_t, const int32_t *, const int32_t *, int32_t *, float *, void *) { NOT_YET("geo_csr_compact_fill"); }
size_t geo_jvp_workspace_bytes(const geo_decoder_desc *, int64_t, int32_t) { return 0; }
int geo_decoder_jvp_edges(const geo_decoder_desc *, const float *, int64_t, const int32_t *, const int32_t *, int64_t, int32_t, float *, void *, size_t, void *) { NOT_YET("geo_decoder_jvp_edges"); }
int geo_decoder_jvp_pairs(const geo_decoder_desc *, const float *, const float *, int64_t, int32_t, float *, void *, size_t, void *) { NOT_YET("geo_decoder_jvp_pairs"); }
int geo_gather_edge_weights(const float *, const int32_t *, int64_t, float *, void *) { NOT_YET("geo_gather_edge_weights"); }
}

"""ctypes binding of libgeo_hip.so (include/geo_hip.h).  There is no fallback: importing a kernel
entry point without the built library raises, and every call checks the returned status."""
import ctypes
import os

import torch  # noqa: F401  -- must come first: libgeo_hip.so has to bind to the HIP runtime PyTorch-ROCm loads,
#                              so that device pointers and streams are shared (two runtimes = "no device")

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libgeo_hip.so")

c_p = ctypes.c_void_p
i32, i64, sz = ctypes.c_int32, ctypes.c_int64, ctypes.c_size_t


class DecoderDesc(ctypes.Structure):
    """geo_decoder_desc of include/geo_hip.h."""
    _fields_ = ([(n, i32) for n in ("latent_dim", "c0", "c1", "c2", "out_channels", "out_size", "norm",
                                    "bn_train", "groups1", "groups2")]
                + [("eps", ctypes.c_float)]
                + [(n, c_p) for n in ("w_in", "b_in", "w1", "b1", "g1", "be1", "rm1", "rv1",
                                      "w2", "b2", "g2", "be2", "rm2", "rv2", "w3", "b3")]
                + [("update_running", i32), ("momentum", ctypes.c_float)])


_SIGNATURES = {
    "geo_version": (ctypes.c_int, []),
    "geo_last_error": (ctypes.c_char_p, []),
    "geo_set_option": (ctypes.c_int, [ctypes.c_char_p, i32]),
    "geo_sssp_workspace_bytes": (sz, [i32, i64, i32]),
    "geo_sssp_multi": (ctypes.c_int, [c_p, c_p, c_p, i32, i64, c_p, i32, c_p, c_p, c_p, c_p, c_p, sz, c_p, c_p]),
    "geo_sssp_nearest_workspace_bytes": (sz, [i32, i64]),
    "geo_sssp_nearest_source": (ctypes.c_int, [c_p, c_p, c_p, i32, i64, c_p, i32, c_p, c_p, c_p, sz, c_p, c_p]),
    "geo_sssp_last_profile": (ctypes.c_int, [c_p, c_p]),
    "geo_sssp_plan": (ctypes.c_int, [ctypes.c_int32, ctypes.c_int32]),
    "geo_prior_attention_fwd": (ctypes.c_int, [c_p, c_p, ctypes.c_float, i32, i32, i32, i32, c_p, c_p, c_p]),
    "geo_prior_attention_bwd": (ctypes.c_int, [c_p, c_p, c_p, ctypes.c_float, c_p, i32, i32, i32, i32, c_p, c_p]),
    "geo_prior_adamw": (ctypes.c_int, [c_p, c_p, c_p, c_p, i64, c_p, c_p, ctypes.c_float, ctypes.c_float, ctypes.c_float,
                                       ctypes.c_float, c_p]),
    "geo_sssp_single_update": (ctypes.c_int, [c_p, c_p, c_p, i32, i32, c_p, c_p, c_p, i32, c_p, sz, c_p, c_p]),
    "geo_kpp_workspace_bytes": (sz, [i32]),
    "geo_cluster_costs": (ctypes.c_int, [c_p, i64, c_p, c_p, c_p, i32, i32, c_p, c_p]),
    "geo_rows_argmin": (ctypes.c_int, [c_p, i64, c_p, i32, i32, c_p, c_p, c_p]),
    "geo_pam_swap_deltas": (ctypes.c_int, [c_p, i64, c_p, c_p, c_p, c_p, c_p, i32, i32, i32, c_p, c_p, c_p]),
    "geo_attach_argmin": (ctypes.c_int, [c_p, i64, i32, c_p, c_p, i32, i64, c_p, c_p, c_p]),
    "geo_kpp_resident_max_nodes": (i32, []),
    "geo_kpp_chain": (ctypes.c_int, [c_p, c_p, c_p, i32, c_p, c_p, c_p, c_p, c_p, i32, i32, i32, i32, i32, c_p, sz, c_p, c_p]),
    "geo_knn_workspace_bytes": (sz, [i64, i32]),
    "geo_knn_topk": (ctypes.c_int, [c_p, i64, i32, i32, i32, i64, i64, c_p, c_p, c_p, sz, c_p]),
    "geo_symmetrize_workspace_bytes": (sz, [i32, i32]),
    "geo_symmetrize_count": (ctypes.c_int, [c_p, c_p, i32, i32, i32, c_p, c_p, c_p, sz, c_p]),
    "geo_symmetrize_fill": (ctypes.c_int, [c_p, c_p, i32, i32, i32, c_p, c_p, c_p, c_p, sz, c_p]),
    "geo_upper_edges_count": (ctypes.c_int, [c_p, c_p, i32, c_p, c_p, c_p, sz, c_p]),
    "geo_upper_edges_fill": (ctypes.c_int, [c_p, c_p, i32, c_p, c_p, c_p, c_p, c_p]),
    "geo_cc_workspace_bytes": (sz, [i32]),
    "geo_connected_components": (ctypes.c_int, [c_p, c_p, i32, c_p, c_p, c_p, sz, c_p]),
    "geo_csr_compact_workspace_bytes": (sz, [i32]),
    "geo_csr_compact_count": (ctypes.c_int, [c_p, c_p, c_p, i32, c_p, i32, c_p, c_p, c_p, c_p, c_p, sz, c_p]),
    "geo_csr_compact_fill": (ctypes.c_int, [c_p, c_p, c_p, i32, c_p, i32, c_p, c_p, c_p, c_p, c_p]),
    "geo_jvp_workspace_bytes": (sz, [ctypes.POINTER(DecoderDesc), i64, i32]),
    "geo_jvp_edges_workspace_bytes": (sz, [ctypes.POINTER(DecoderDesc), i64, i64, i32]),
    "geo_decoder_jvp_edges": (ctypes.c_int, [ctypes.POINTER(DecoderDesc), c_p, i64, c_p, c_p, i64, i32, c_p, c_p, sz, c_p]),
    "geo_decoder_jvp_pairs": (ctypes.c_int, [ctypes.POINTER(DecoderDesc), c_p, c_p, i64, i32, c_p, c_p, sz, c_p]),
    "geo_gather_edge_weights": (ctypes.c_int, [c_p, c_p, i64, c_p, c_p]),
}

EXPORTS = tuple(_SIGNATURES)
_lib = None


class GeoHipError(RuntimeError):
    pass


def load() -> ctypes.CDLL:
    """Load libgeo_hip.so (built by vqvae_amd/csrc/Makefile or __graft_entry__.build())."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise GeoHipError(
                f"{LIB_PATH} is missing: build it with `make -C vqvae_amd/csrc` "
                "(the geodesic-codebook path has no CPU fallback)")
        L = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in _SIGNATURES.items():
            fn = getattr(L, name)           # AttributeError if the library lacks a declared symbol
            fn.restype, fn.argtypes = res, args
        _lib = L
    return _lib


def check(status: int, what: str) -> None:
    if status != 0:
        raise GeoHipError(f"{what} failed with status {status}: {load().geo_last_error().decode()}")

"""CPU-side checks: the C-ABI library loads and exports every symbol include/geo_hip.h declares, the host
logic of the reference-API wrappers (validation, pull structure, k-means++ draw, decoder export, CLI flags)
and the rule that the product never touches oracle/ and fails loudly without a GPU."""
import os
import re
import subprocess
import sys

import numpy as np
import pytest
import torch
from scipy import sparse

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HAS_GPU = torch.cuda.is_available()


def header_symbols():
    text = open(os.path.join(ROOT, "include", "geo_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(geo_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    from vqvae_amd import _lib
    lib = _lib.load()
    declared = header_symbols()
    assert len(declared) >= 20
    assert sorted(_lib.EXPORTS) == declared
    for name in declared:
        assert hasattr(lib, name), name
    assert lib.geo_version() >= 100
    out = subprocess.run(["nm", "-D", "--defined-only", _lib.LIB_PATH], capture_output=True, text=True).stdout
    exported = set(re.findall(r" T (geo_[a-z0-9_]+)", out))
    assert set(declared) <= exported


def test_abi_has_no_torch_types():
    text = open(os.path.join(ROOT, "include", "geo_hip.h")).read()
    assert 'extern "C"' in text
    code = re.sub(r"/\*.*?\*/", "", text, flags=re.S)           # signatures only, comments stripped
    assert "torch" not in code.lower() and "at::" not in code and "Tensor" not in code


def test_product_never_imports_oracle():
    for base, _, files in os.walk(os.path.join(ROOT, "vqvae_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                src = open(os.path.join(base, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle", src, flags=re.M), f
                assert "liboracle" not in src, f


@pytest.mark.skipif(HAS_GPU, reason="checks the no-GPU failure mode")
def test_fails_loudly_without_gpu():
    from vqvae_amd import _lib
    from vqvae_amd.geo import build_knn_graph, dijkstra_multi_source
    W = sparse.csr_matrix((np.ones(2, np.float32), ([0, 1], [1, 0])), shape=(2, 2))
    with pytest.raises(_lib.GeoHipError):
        dijkstra_multi_source(W, [0])
    with pytest.raises(_lib.GeoHipError):
        build_knn_graph(np.random.RandomState(0).randn(10, 4).astype(np.float32), k=3)


def test_validation_errors_match_reference_classes():
    from vqvae_amd.geo.geo_shortest_paths import dijkstra_multi_source, distances_between, ensure_valid_graph
    from vqvae_amd.geo.kmeans_optimized import fit_kmedoids_optimized
    W = sparse.csr_matrix((np.ones(2, np.float32), ([0, 1], [1, 0])), shape=(2, 2))
    with pytest.raises(TypeError):
        ensure_valid_graph(np.zeros((3, 3)))
    with pytest.raises(ValueError):
        ensure_valid_graph(sparse.csr_matrix((3, 4)))
    Wn = W.copy()
    Wn.data[0] = -1
    with pytest.raises(ValueError):
        ensure_valid_graph(Wn)
    assert sparse.isspmatrix_csr(ensure_valid_graph(W.tocoo()))
    with pytest.raises(ValueError):
        dijkstra_multi_source(W, [])
    with pytest.raises(ValueError):
        distances_between(W, [0], [])
    with pytest.raises(ValueError):
        fit_kmedoids_optimized(W, K=1, init="bogus")


def test_knn_host_special_cases_need_no_gpu():
    from vqvae_amd.geo.knn_graph_optimized import build_knn_graph, build_knn_graph_auto
    W, nb = build_knn_graph(np.empty((0, 8), np.float32), k=10)
    assert W.shape == (0, 0) and nb["distances"].shape == (0, 0) and nb["indices"].shape == (0, 0)
    W, nb = build_knn_graph(np.zeros((1, 8), np.float32), k=10)
    assert W.shape == (1, 1) and nb["indices"].shape == (1, 0)
    W, nb = build_knn_graph(np.zeros((25, 4), np.float32), k=0)
    assert W.nnz == 0 and nb["indices"].shape == (25, 0) and W.dtype == np.float32
    with pytest.raises(RuntimeError):
        build_knn_graph_auto(np.zeros((5, 2), np.float32), k=1, force_method="faiss")
    with pytest.raises(AssertionError):
        build_knn_graph(np.zeros(4, np.float32))


def test_pull_structure():
    from vqvae_amd.geo.geo_shortest_paths import _pull_structure
    W = sparse.csr_matrix((np.array([5.0, 1.0, 2.0], np.float32), ([0, 1, 2], [1, 0, 0])), shape=(3, 3))
    G = _pull_structure(W, directed=True)                   # row v = edges into v
    assert G[1, 0] == 5.0 and G[0, 1] == 1.0 and G[0, 2] == 2.0 and G.nnz == 3
    U = _pull_structure(W, directed=False)                  # both directions, duplicates kept as parallel edges
    assert U.indptr[-1] == 6
    rows = np.repeat(np.arange(3), np.diff(U.indptr))
    pairs = sorted(zip(rows, U.indices, U.data))
    assert pairs == [(0, 1, 1.0), (0, 1, 5.0), (0, 2, 2.0), (1, 0, 1.0), (1, 0, 5.0), (2, 0, 2.0)]
    S = sparse.csr_matrix((np.ones(2, np.float32), ([0, 1], [1, 0])), shape=(2, 2))
    assert _pull_structure(S, directed=False) is not None and _pull_structure(S, directed=False).nnz == 2


def test_kpp_draw_matches_oracle_draw():
    from oracle import kmedoids as ok
    from vqvae_amd.geo.kmeans_optimized import _next_center
    r = np.random.RandomState(0)
    d = r.rand(500).astype(np.float32)
    d[[3, 77]] = np.inf
    centers = [5, 9]
    a = _next_center(np.random.RandomState(7), 500, d.copy(), list(centers))
    probs = ok._seeding_probs(d.copy(), list(centers))
    b = ok._draw_next(np.random.RandomState(7), 500, probs, list(centers))
    assert a == b
    assert _next_center(np.random.RandomState(1), 3, np.zeros(3, np.float32), [0, 1, 2]) is None
    assert _next_center(np.random.RandomState(1), 3, np.zeros(3, np.float32), [0]) in (1, 2)


def test_host_draw_with_given_deviate_equals_randomstate_choice():
    """The draw the host makes when the device declines one (`_draw_with_u`: no copy for finite distances, float64 cumulative sum
    taken directly, the division by cdf[-1] only around the answer) is RandomState.choice(N, p=probs) with that deviate -- also
    for deviates sitting exactly on, just below and just above a step of the cdf, with inf / zero distances, N = 1."""
    from vqvae_amd.geo.kmeans_optimized import _draw_with_u

    def choice_with_u(N, d_min, centers, u):                    # kmeans_optimized.py:47-61 + numpy's legacy choice, spelled out
        finite = np.isfinite(d_min)
        safe = np.where(finite, d_min, np.max(d_min[finite]) * 2.0) if finite.any() else np.ones_like(d_min)
        probs = safe ** 2
        probs[centers] = 0.0
        total = probs.sum()
        if not total > 0:
            return None
        probs /= total
        cdf = probs.astype(np.float64).cumsum()
        cdf /= cdf[-1]
        return int(cdf.searchsorted(u, side="right"))

    r = np.random.RandomState(0)
    for trial in range(120):
        N = int(r.choice([1, 2, 5, 100, 1000, 8193, 50000]))
        d = np.abs(r.randn(N)).astype(np.float32)
        kind = trial % 5
        if kind == 1:
            d[r.rand(N) < 0.3] = np.inf
        elif kind == 2:
            d[:] = np.inf
        elif kind == 3:
            d[r.rand(N) < 0.7] = 0
        elif kind == 4:
            d[:] = 0
        centers = list(r.choice(N, size=min(N, int(r.randint(1, 5))), replace=False))
        deviates = [float(r.rand()), 0.0, 1.0 - 1e-16]
        if choice_with_u(N, d.copy(), centers, 0.5) is not None:
            f = np.isfinite(d)
            safe = np.where(f, d, np.max(d[f]) * 2.0) if f.any() else np.ones_like(d)
            p = safe ** 2
            p[centers] = 0
            p /= p.sum()
            cdf = p.astype(np.float64).cumsum()
            cdf /= cdf[-1]
            for j in r.randint(0, N, 4):
                deviates += [u for u in (float(cdf[j]), float(np.nextafter(cdf[j], 0)), float(np.nextafter(cdf[j], 2))) if u < 1]
        for u in deviates:
            assert _draw_with_u(N, d.copy(), centers, u) == choice_with_u(N, d.copy(), centers, u), (N, kind, u)


def test_decoder_module_and_export_on_cpu():
    from oracle import metric as om
    from vqvae_amd.spatial_decoder import DecoderExport, SpatialDecoder, looks_like_spatial_decoder, make_norm
    for norm, code in (("batch", 1), ("group", 2), ("none", 0)):
        sd = om.make_decoder_state(1, 16, 1, norm_type=norm)
        dec = SpatialDecoder(1, (256, 128, 64), 16, 28, norm)
        dec.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()})      # same key names
        assert looks_like_spatial_decoder(dec)
        ex = DecoderExport(dec, torch.device("cpu"))
        assert (ex.desc.latent_dim, ex.desc.c0, ex.desc.c1, ex.desc.c2, ex.desc.out_channels) == (16, 256, 128, 64, 1)
        assert ex.desc.out_size == 28 and ex.desc.norm == code
        assert ex.desc.bn_train == (1 if norm == "batch" else 0)
        dec.eval()
        assert DecoderExport(dec, torch.device("cpu")).desc.bn_train == 0
        # the module computes what the oracle's closed form differentiates
        x = torch.randn(4, 16, 1, 1)
        assert dec(x).shape == (4, 1, 4, 4)
    assert SpatialDecoder(3, (256, 128, 64), 32, 32, "batch")(torch.randn(2, 32, 1, 1)).shape == (2, 3, 8, 8)
    assert isinstance(make_norm("group", 48), torch.nn.GroupNorm) and make_norm("group", 48).num_groups == 24
    assert not looks_like_spatial_decoder(torch.nn.Sequential(torch.nn.Linear(4, 4)))
    with pytest.raises(ValueError):
        SpatialDecoder(1, (8, 8, 8), 4, 64, "none")


def test_generic_decoder_path_on_cpu_matches_reference_properties():
    """The reference's tests/test_riemannian_metric.py:25-62 on the Linear test decoder (no kernel exists
    for arbitrary modules: torch.func.jvp on the decoder's own device)."""
    from vqvae_amd.geo.riemannian_metric import decoder_logits_to_img, edge_lengths_riemannian

    class DummyDec(torch.nn.Module):
        def __init__(self, d=16, hw=784):
            super().__init__()
            self.lin = torch.nn.Linear(d, hw)

        def forward(self, z):
            return self.lin(z).view(z.size(0), 1, 28, 28)

    dec = DummyDec().eval()
    g = torch.Generator().manual_seed(42)
    zi = torch.randn(127, 16, generator=g)
    zj = zi + 0.1 * torch.randn(127, 16, generator=g)
    L = edge_lengths_riemannian(dec, zi, zj, batch_size=32)
    assert L.shape == (127,) and L.dtype == torch.float32 and torch.all(L >= 0)
    assert torch.allclose(L, edge_lengths_riemannian(dec, zj, zi, batch_size=64), rtol=1e-4, atol=1e-6)
    assert torch.allclose(edge_lengths_riemannian(dec, zi, zj, batch_size=16),
                          edge_lengths_riemannian(dec, zi, zj, batch_size=1024), rtol=1e-5, atol=1e-7)
    Lh = edge_lengths_riemannian(dec, zi, zi + 0.5 * (zj - zi), batch_size=64)
    assert torch.all(Lh <= L + 1e-6) and 0.3 < (Lh / (L + 1e-8)).mean().item() < 0.7
    with pytest.raises(AssertionError):
        edge_lengths_riemannian(dec, zi, zj[:5])
    assert torch.equal(decoder_logits_to_img(torch.zeros(2)), torch.full((2,), 0.5))


def test_cli_flag_set_is_the_reference_one():
    from vqvae_amd.scripts.build_codebook import make_parser
    args = make_parser().parse_args([
        "--latents_path", "z.pt", "--out_dir", "o", "--vae_ckpt_path", "b.pt", "--in_channels", "1",
        "--output_image_size", "28", "--latent_dim", "16", "--enc_channels", "64", "128", "256",
        "--dec_channels", "256", "128", "64", "--recon_loss", "mse", "--norm_type", "batch"])
    assert (args.k, args.sym, args.K, args.init, args.seed, args.batch_size) == (20, "union", 512, "kpp", 42, 512)
    assert args.mse_use_sigmoid is False and args.dec_channels == [256, 128, 64]
    with pytest.raises(SystemExit):
        make_parser().parse_args(["--latents_path", "z.pt"])


def test_sssp_plan_keeps_32bit_row_offsets_below_2_pow_25_nodes():
    """The 16- / 32-source row kernels gather through 32-bit byte offsets (node << 7): geo_sssp_plan (the choice
    geo_sssp_multi makes, host arithmetic only) must never pick them once n * 128 bytes reaches 2^32."""
    from vqvae_amd import _lib
    plan = _lib.load().geo_sssp_plan
    assert plan(60000, 512) == 16 + 1000 + 2000                  # C2: chunked rows, fixed point tried first
    assert plan(1_000_000, 1024) == 16 + 1000 + 2000             # C4
    assert plan((1 << 25) - 1, 512) == 16 + 1000 + 2000
    assert plan(1 << 25, 512) == 64                              # 64-bit indexing only from here on
    assert plan(40_000_000, 1024) == 64
    assert plan(1 << 25, 8) == 16                                # few sources: the 16-source NODE kernel (size_t indices)
    assert plan(2048, 64) == 64 and plan(0, 4) < 0

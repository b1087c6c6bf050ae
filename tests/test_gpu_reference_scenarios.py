"""The scenarios of the reference's own test-suite that the other GPU test files do not restate yet (reference
tests/test_kmeans_optimized.py:201, tests/test_riemannian_metric.py:42, tests/test_knn_graph.py, the FAISS guard of
src/geo/knn_graph_optimized.py:73-74), run through the drop-in API on the HIP path and checked against the oracle."""
import numpy as np
import pytest
import torch
from scipy import sparse

from conftest import latents

pytestmark = pytest.mark.gpu


def _two_component_graph():
    from oracle import knn as okn
    A, _ = okn.build_knn_graph(latents(300, 8, 31), k=6, mode="distance", sym="union")
    B, _ = okn.build_knn_graph(latents(200, 8, 32), k=6, mode="distance", sym="union")
    return sparse.block_diag((A, B), format="csr", dtype=np.float32)


def test_kmedoids_with_connectivity_check_connected_and_disconnected():
    from oracle import kmedoids as okm
    from oracle import knn as okn
    from vqvae_amd.geo.kmeans_optimized import fit_kmedoids_with_connectivity_check
    W, _ = okn.build_knn_graph(latents(600, 8, 30), k=8, mode="distance", sym="union")
    med, assign, qe, meta = fit_kmedoids_with_connectivity_check(W, K=12, init="kpp", seed=7)
    mo, ao, qo = okm.fit_kmedoids_optimized(W, K=12, init="kpp", seed=7)
    np.testing.assert_array_equal(med, mo)
    np.testing.assert_array_equal(assign, ao)
    assert qe == qo and med.dtype == int and assign.dtype == int
    assert meta["n_components"] == 1 and meta["n_nodes"] == 600 and meta["n_edges"] == W.nnz
    assert meta["largest_component_size"] == 600 and meta["n_medoids"] == 12 and meta["quantization_error"] == qe
    W2 = _two_component_graph()
    med, assign, qe, meta = fit_kmedoids_with_connectivity_check(W2, K=9, init="kpp", seed=3)
    mo, ao, qo = okm.fit_kmedoids_optimized(W2, K=9, init="kpp", seed=3)
    np.testing.assert_array_equal(med, mo)
    np.testing.assert_array_equal(assign, ao)
    assert qe == qo and meta["n_components"] == 2 and meta["largest_component_size"] == 300
    assert np.isfinite(qe)                                         # unreachable pairs do not enter the error


def test_random_init_and_bad_init():
    from oracle import kmedoids as okm
    from oracle import knn as okn
    from vqvae_amd.geo.kmeans_optimized import fit_kmedoids_optimized
    W, _ = okn.build_knn_graph(latents(400, 8, 33), k=6, mode="distance", sym="union")
    med, assign, qe = fit_kmedoids_optimized(W, K=10, init="random", seed=11)
    mo, ao, qo = okm.fit_kmedoids_optimized(W, K=10, init="random", seed=11)
    np.testing.assert_array_equal(med, mo)
    np.testing.assert_array_equal(assign, ao)
    assert qe == qo
    with pytest.raises(ValueError):
        fit_kmedoids_optimized(W, K=4, init="farthest", seed=0)


def test_half_step_gives_about_half_the_length_on_the_hip_decoder():
    """Edge lengths are first order in the step (reference test_scaling_step_cpu, there on a linear decoder): on the
    SpatialDecoder kernels too -- eval-mode BatchNorm, so samples are independent."""
    from oracle import metric as om
    from vqvae_amd.geo.riemannian_metric import edge_lengths_riemannian
    from vqvae_amd.spatial_decoder import SpatialDecoder
    sd = om.make_decoder_state(10, 16, 1, norm_type="batch")
    dec = SpatialDecoder(1, (256, 128, 64), 16, 28, "batch")
    dec.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()})
    dec = dec.cuda().eval()
    g = torch.Generator().manual_seed(0)
    zi = torch.randn(256, 16, generator=g)
    v = 0.05 * torch.randn(256, 16, generator=g)
    L1 = edge_lengths_riemannian(dec, zi, zi + v, batch_size=64)
    Lh = edge_lengths_riemannian(dec, zi, zi + 0.5 * v, batch_size=64)
    ratio = (Lh / (L1 + 1e-8)).mean().item()
    assert L1.is_cuda and torch.all(Lh <= L1 + 1e-6) and 0.45 < ratio < 0.55
    Ls = edge_lengths_riemannian(dec, zi + v, zi, batch_size=100)      # swapping the endpoints changes nothing
    assert torch.allclose(L1, Ls, rtol=1e-5, atol=1e-7)


def test_faiss_entry_points_behave_like_the_reference_without_faiss():
    from vqvae_amd.geo.knn_graph_optimized import build_knn_graph_auto, build_knn_graph_faiss
    z = latents(128, 8, 34)
    with pytest.raises(RuntimeError):
        build_knn_graph_faiss(z, k=5)
    with pytest.raises(RuntimeError):
        build_knn_graph_auto(z, k=5, force_method="faiss")
    W, info = build_knn_graph_auto(z, k=5, force_method="sklearn", mode="connectivity", sym="mutual")
    assert W.shape == (128, 128) and (W != W.T).nnz == 0 and set(np.unique(W.data)) <= {1.0}
    assert info["indices"].shape == (128, 5) and info["indices"].dtype == np.int64

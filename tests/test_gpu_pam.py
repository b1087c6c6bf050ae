"""Extension without a reference implementation: PAM swaps over the resident all-pairs geodesic matrix.
csrc/medoid.hip's pam_swap_kernel evaluates every (medoid, candidate) exchange in ONE pass over the N x N matrix (FastPAM1
form); parity is against the brute-force definition restated in oracle/kmedoids.py (every pair tried, total cost
recomputed), and at the 60 000-latent size against the change of the total cost measured independently."""
import numpy as np
import pytest
import torch

from conftest import latents

pytestmark = pytest.mark.gpu


def _graph(n, d, seed, k=8):
    from oracle import knn as okn
    W, _ = okn.build_knn_graph(latents(n, d, seed), k=k, mode="distance", sym="union")
    return W.tocsr()


@pytest.mark.parametrize("power", [1, 2])
def test_swap_pass_equals_the_definition(power):
    from oracle import kmedoids as ok
    from vqvae_amd._device import DeviceCSR, device
    from vqvae_amd.geo.geo_shortest_paths import all_pairs_geodesic_device
    from vqvae_amd.geo.kmeans_optimized import pam_swap_pass_device
    W = _graph(260, 8, 3)
    D = all_pairs_geodesic_device(DeviceCSR.from_scipy(W, device()))
    Dh = D.cpu().numpy()
    np.testing.assert_array_equal(Dh, ok.all_pairs(W))
    rs = np.random.RandomState(5)
    for K in (1, 2, 7):
        med = rs.choice(260, K, replace=False)
        delta, i, x, total = pam_swap_pass_device(D, torch.from_numpy(med.astype(np.int32)).to(D.device), power)
        d_o, i_o, x_o = ok.pam_swap_pass(Dh, med, power)
        assert (i, x) == (i_o, x_o)
        assert abs(delta - d_o) <= 1e-9 * abs(total) and abs(total - ok.total_cost(Dh, med, power)) <= 1e-9 * total


def test_pam_converges_like_the_restatement_and_never_raises_the_cost():
    from oracle import kmedoids as ok
    from vqvae_amd.geo.kmeans_optimized import fit_kmedoids_optimized, fit_kmedoids_pam
    W = _graph(300, 6, 11)
    med0, _, _ = fit_kmedoids_optimized(W, K=6, init="kpp", seed=42)
    med, assign, qe, hist = fit_kmedoids_pam(W, K=6, init="kpp", seed=42, max_swaps=40, power=2)
    Dh = ok.all_pairs(W)
    med_o, assign_o, hist_o = ok.pam(Dh, med0, power=2, max_swaps=40)
    np.testing.assert_array_equal(med, med_o)
    np.testing.assert_array_equal(assign, assign_o)
    np.testing.assert_allclose(hist, hist_o, rtol=1e-9)
    assert all(b < a for a, b in zip(hist, hist[1:])) and len(hist) >= 2          # every applied swap lowers the cost
    assert abs(qe - hist[-1]) <= 1e-5 * qe                                           # qe (f32 sum) == total cost (fp64)


def test_one_swap_pass_at_the_60000_latent_size():
    """N = 60 000, K = 512: the swap the kernel picks changes the total cost by exactly the change it predicted (the new
    cost is recomputed from the matrix rows, independently of the kernel's formula); the pass reads 14.4 GB once."""
    import time
    from oracle import synthetic as syn
    from vqvae_amd._device import device
    from vqvae_amd.geo.geo_shortest_paths import all_pairs_geodesic_device
    from vqvae_amd.geo.kmeans_optimized import assign_from_rows_device, fit_kmedoids_optimized, pam_swap_pass_device
    from vqvae_amd.geo.knn_graph_optimized import knn_graph_device
    dev = device()
    z = torch.from_numpy(syn.gauss_latents(60000, 16, 0)).to(dev)
    G, _, _ = knn_graph_device(z, 20, mode="distance", sym="union")
    med, _, _ = fit_kmedoids_optimized(G, K=512, init="kpp", seed=42)
    D = all_pairs_geodesic_device(G)
    m = torch.from_numpy(med.astype(np.int32)).to(dev)
    pam_swap_pass_device(D, m, 2)                                           # warm-up
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    delta, i, x, total = pam_swap_pass_device(D, m, 2)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) * 1e3
    assert delta < 0
    m2 = m.clone()
    m2[i] = x
    dmin, _ = assign_from_rows_device(D, m2)
    new_total = float((dmin.double() ** 2).sum())
    assert abs((new_total - total) - delta) <= 1e-9 * total
    print(f"swap pass over 60000 x 60000: {ms:.2f} ms ({60000 * 60000 * 4 / ms / 1e6:.0f} GB/s of matrix bytes), "
          f"best swap lowers the total cost {total:.1f} by {-delta:.3f}")
    del D
    torch.cuda.empty_cache()

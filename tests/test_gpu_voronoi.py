"""Extension (no reference implementation): all-pairs matrix, medoid update and re-assignment on the
GPU against the numpy restatement oracle/kmedoids.py (voronoi_iteration)."""
import numpy as np
import pytest
import torch

from conftest import latents

pytestmark = pytest.mark.gpu


def _graph(n, d, k, seed):
    from oracle import knn as okn
    W, _ = okn.build_knn_graph(latents(n, d, seed), k=k, mode="distance", sym="union")
    mask = okn.largest_connected_component(W)
    return W[mask][:, mask].tocsr()


def test_all_pairs_matrix_equals_scipy_rows():
    from oracle import kmedoids as okm
    from vqvae_amd.geo.geo_shortest_paths import all_pairs_geodesic_device
    from vqvae_amd.geo.kmeans_optimized import _to_device_graph
    W = _graph(1500, 8, 8, 3)
    D = all_pairs_geodesic_device(_to_device_graph(W), block=400)           # ragged last block
    ref = okm.all_pairs(W)
    np.testing.assert_array_equal(D.cpu().numpy(), ref)
    with pytest.raises(ValueError):
        all_pairs_geodesic_device(_to_device_graph(W), max_bytes=1 << 20)


@pytest.mark.parametrize("power", [2, 1])
def test_medoid_update_and_reassignment_equal_the_oracle(power):
    from oracle import kmedoids as okm
    from vqvae_amd._device import device
    from vqvae_amd.geo.kmeans_optimized import assign_from_rows_device, medoid_update_device
    W = _graph(2500, 16, 10, 5)
    n, K = W.shape[0], 40
    D = okm.all_pairs(W)
    med0 = np.random.RandomState(1).choice(n, K, replace=False)
    assign0 = np.argmin(D[med0], axis=0)
    new_o, cost_o = okm.medoid_update(D, assign0, med0, power)
    dev = device()
    Dd = torch.from_numpy(D).to(dev)
    new_g, cost_g = medoid_update_device(Dd, torch.from_numpy(assign0.astype(np.int32)).to(dev),
                                         torch.from_numpy(med0.astype(np.int32)).to(dev), power)
    np.testing.assert_array_equal(cost_g.cpu().numpy(), cost_o)             # same summation tree: bit-equal fp64
    np.testing.assert_array_equal(new_g.cpu().numpy(), new_o)
    dmin, arg = assign_from_rows_device(Dd, new_g)
    np.testing.assert_array_equal(arg.cpu().numpy(), np.argmin(D[new_o], axis=0))
    np.testing.assert_array_equal(dmin.cpu().numpy(), D[new_o].min(axis=0))
    # a cluster without members keeps its medoid; ties between rows go to the first medoid
    dup = new_o.copy()
    dup[7] = dup[3]                                                         # medoid 7 duplicates medoid 3: cluster 7 is empty
    dmin2, arg2 = assign_from_rows_device(Dd, torch.from_numpy(dup.astype(np.int32)).to(dev))
    a2 = arg2.cpu().numpy()
    np.testing.assert_array_equal(a2, np.argmin(D[dup], axis=0))
    assert not (a2 == 7).any()
    kept, _ = medoid_update_device(Dd, arg2, torch.from_numpy(dup.astype(np.int32)).to(dev), power)
    ref_kept, _ = okm.medoid_update(D, a2, dup, power)
    np.testing.assert_array_equal(kept.cpu().numpy(), ref_kept)
    assert kept.cpu().numpy()[7] == dup[7]


@pytest.mark.parametrize("d,init,min_updates", [(16, "kpp", 1), (2, "kpp", 5), (2, "random", 5)])
def test_voronoi_iteration_end_to_end_equals_the_oracle(d, init, min_updates):
    """16-dimensional clouds settle after one update; planar ones keep moving for many."""
    from oracle import kmedoids as okm
    from vqvae_amd.geo.kmeans_optimized import fit_kmedoids_voronoi
    W = _graph(3000, d, 10 if d == 16 else 8, 9)
    med0, _, _ = okm.fit_kmedoids_optimized(W, K=64, init=init, seed=42)
    med_o, assign_o, qe_o, hist_o = okm.voronoi_iteration(W, med0, max_iter=12)
    med_g, assign_g, qe_g, hist_g = fit_kmedoids_voronoi(W, K=64, init=init, seed=42, max_iter=12)
    np.testing.assert_array_equal(med_g, med_o)
    np.testing.assert_array_equal(assign_g, assign_o)
    assert hist_g == hist_o and qe_g == qe_o
    assert len(hist_g) >= 1 + min_updates                                   # the updates did move the medoids
    assert all(b <= a for a, b in zip(hist_g, hist_g[1:]))                  # QE (squared distances) never increases

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


def test_codes_of_latents_outside_the_graph_vs_oracle():
    """Held-out latents joined to the training graph by k attachment edges (Euclidean lengths here: exact comparison)."""
    from oracle import kmedoids as okm
    from oracle import knn as okn
    from oracle import pipeline as opl
    from vqvae_amd._device import device
    from vqvae_amd.geo.kmeans_optimized import _to_device_graph
    from vqvae_amd.training.assign_codes_val_geodesic import assign_codes_geodesic, attach_neighbors_device
    z = latents(2600, 12, 4)
    W, _ = okn.build_knn_graph(z[:2000], k=10, mode="distance", sym="union")
    mask = okn.largest_connected_component(W)
    W = W[mask][:, mask].tocsr()
    zg, znew = z[:2000][mask], z[2000:].copy()
    znew[:7] = zg[100:107]                                                  # some held-out latents coincide with nodes
    med, assign_tr, _ = okm.fit_kmedoids_optimized(W, K=32, init="kpp", seed=42)
    codes_o, dist_o, idx_o, len_o = opl.assign_new_latents(znew, zg, W, med, k=12)
    dev = device()
    out = assign_codes_geodesic(torch.from_numpy(znew), torch.from_numpy(zg).to(dev), _to_device_graph(W), med, k=12)
    np.testing.assert_array_equal(out["neighbors"].cpu().numpy(), idx_o)
    np.testing.assert_array_equal(out["lengths"].cpu().numpy(), len_o)
    np.testing.assert_array_equal(out["dist"].cpu().numpy(), dist_o)
    np.testing.assert_array_equal(out["codes"].cpu().numpy(), codes_o)
    # a latent equal to a graph node, attached only along edges the graph has (5 nearest of a 10-NN union graph), gets
    # that node's training assignment: no attachment can be a shortcut
    D = okm.all_pairs(W)
    near = assign_codes_geodesic(torch.from_numpy(znew[:7]), torch.from_numpy(zg).to(dev), _to_device_graph(W), med, k=5)
    np.testing.assert_array_equal(near["codes"].cpu().numpy(), np.argmin(D[np.asarray(med)][:, 100:107], axis=0))
    np.testing.assert_array_equal(near["dist"].cpu().numpy(), D[np.asarray(med)][:, 100:107].min(axis=0))
    # blocks of any size give the same neighbours; k larger than the graph is clipped; an empty set is an empty answer
    i2, _ = attach_neighbors_device(torch.from_numpy(znew).to(dev), torch.from_numpy(zg).to(dev), 12, rows_per_block=37)
    np.testing.assert_array_equal(i2.cpu().numpy(), idx_o)
    small = assign_codes_geodesic(torch.from_numpy(znew[:5]), torch.from_numpy(zg).to(dev), _to_device_graph(W), med, k=10 ** 6)
    assert small["neighbors"].shape == (5, zg.shape[0])
    empty = assign_codes_geodesic(torch.from_numpy(znew[:0]), torch.from_numpy(zg).to(dev), _to_device_graph(W), med, k=12)
    assert empty["codes"].shape == (0,)


def test_codes_of_held_out_latents_with_pull_back_lengths():
    """Attachment edges measured through the decoder (the training metric): distances within the length tolerance of the
    float32 CPU restatement, codes equal except where two medoids are that close."""
    from oracle import kmedoids as okm
    from oracle import knn as okn
    from oracle import metric as om
    from oracle import pipeline as opl
    from vqvae_amd._device import device
    from vqvae_amd.geo.kmeans_optimized import _to_device_graph
    from vqvae_amd.spatial_decoder import SpatialDecoder
    from vqvae_amd.training.assign_codes_val_geodesic import assign_codes_geodesic
    z = latents(1500, 16, 8)
    W, _ = okn.build_knn_graph(z[:1200], k=8, mode="distance", sym="union")
    mask = okn.largest_connected_component(W)
    W = W[mask][:, mask].tocsr()
    zg, znew = z[:1200][mask], z[1200:]
    med, _, _ = okm.fit_kmedoids_optimized(W, K=16, init="kpp", seed=1)
    sd = om.make_decoder_state(3, 16, 1, norm_type="batch")
    dec = SpatialDecoder(1, (256, 128, 64), 16, 28, "batch")
    dec.load_state_dict({k_: torch.from_numpy(np.asarray(v)) for k_, v in sd.items()})
    dev = device()
    out = assign_codes_geodesic(torch.from_numpy(znew), torch.from_numpy(zg).to(dev), _to_device_graph(W), med, k=6,
                                decoder=dec.to(dev).eval(), batch_size=256)
    idx = out["neighbors"].cpu().numpy()
    V, kk = idx.shape
    ref_len = om.edge_lengths(sd, "batch", 28, np.repeat(znew, kk, axis=0), zg[idx.reshape(-1)], 256, False).numpy().reshape(V, kk)
    np.testing.assert_allclose(out["lengths"].cpu().numpy(), ref_len, rtol=2e-5)
    codes_o, dist_o, _, _ = opl.assign_new_latents(znew, zg, W, med, k=6, lengths=ref_len.astype(np.float32))
    np.testing.assert_allclose(out["dist"].cpu().numpy(), dist_o, rtol=2e-5)
    assert (out["codes"].cpu().numpy() == codes_o).mean() > 0.99


"""GPU parity of csrc/knn.hip + csrc/graph.hip through vqvae_amd.geo.knn_graph_optimized against the
golden vectors (reference outputs), the oracle, and the reference's structural tests
(reference tests/test_knn_graph.py, tests/test_integration_knn_geo.py)."""
import numpy as np
import pytest
import scipy.sparse as sp

from conftest import clustered_latents, csr_from_golden, latents

pytestmark = pytest.mark.gpu

CASES = {"g16": (2048, 16, 0), "g32": (512, 32, 1), "g64": (300, 64, 2), "g8": (400, 8, 3)}


def _ulp_diff(a, b):
    return np.max(np.abs(a.astype(np.float64) - b.astype(np.float64)) / np.spacing(np.abs(b).astype(np.float32)))


@pytest.mark.parametrize("name", list(CASES))
def test_golden_graph_structure_and_weights(golden, name):
    from vqvae_amd.geo.knn_graph_optimized import build_knn_graph_auto
    g = golden("knn")
    N, d, seed = CASES[name]
    z = latents(N, d, seed)
    for k in (1, 5, 20):
        for mode in ("connectivity", "distance"):
            for sym in ("union", "mutual"):
                tag = f"{name}/k{k}/{mode}/{sym}"
                W, info = build_knn_graph_auto(z, k=k, mode=mode, sym=sym)
                assert isinstance(W, sp.csr_matrix) and W.dtype == np.float32 and W.shape == (N, N)
                assert W.has_sorted_indices
                np.testing.assert_array_equal(W.indptr, g[f"{tag}/indptr"], err_msg=tag)
                np.testing.assert_array_equal(W.indices, g[f"{tag}/indices"], err_msg=tag)
                if mode == "distance":
                    assert _ulp_diff(W.data, g[f"{tag}/data"]) <= 1.0, tag       # <= 1 ulp f32 (SURVEY 8a gate)
                else:
                    assert (W.data == 1.0).all()
        # neighbour lists: same sets, same order (distinct distances), distances within 1 ulp
        assert info["indices"].dtype == np.int64 and info["distances"].dtype == np.float32
        np.testing.assert_array_equal(info["indices"], g[f"{name}/k{k}/nbr_indices"])
        assert _ulp_diff(info["distances"], g[f"{name}/k{k}/nbr_distances"]) <= 1.0


def test_duplicates_match_modulo_ties(golden):
    """Exactly tied fp64 distances (duplicate latents) are ordered by heap accident in sklearn; ours are
    ordered by index.  Everything not involving a tie must still agree with the reference."""
    from vqvae_amd.geo.knn_graph_optimized import build_knn_graph_auto
    from oracle import knn as okn
    g = golden("knn")
    z = clustered_latents(256, 16, 4)
    dup_nodes = {3, 5, 15, 16, 17, 128, 255}
    for k in (1, 5, 20):
        for mode in ("connectivity", "distance"):
            for sym in ("union", "mutual"):
                W, info = build_knn_graph_auto(z, k=k, mode=mode, sym=sym)
                Wo, info_o = okn.build_knn_graph_auto(z, k=k, mode=mode, sym=sym)
                # bit-identical to the oracle (same tie rule) ...
                np.testing.assert_array_equal(W.indptr, Wo.indptr)
                np.testing.assert_array_equal(W.indices, Wo.indices)
                np.testing.assert_array_equal(info["indices"], info_o["indices"])
                # ... and equal to the reference away from the duplicated points
                Wr = csr_from_golden(g, f"dup16/k{k}/{mode}/{sym}", 256, with_data=False)
                diff = (abs(sp.csr_matrix((np.ones(W.nnz), W.indices, W.indptr), shape=W.shape) - Wr)).tocoo()
                touched = set(diff.row[diff.data != 0]) | set(diff.col[diff.data != 0])
                assert all((r in dup_nodes) or (c in dup_nodes) for r, c in zip(diff.row, diff.col)), touched
        ref_d = g[f"dup16/k{k}/nbr_distances"]
        assert _ulp_diff(info["distances"] + 1, ref_d + 1) <= 64       # same distance multisets row by row


def test_reference_edge_cases():
    from vqvae_amd.geo.knn_graph_optimized import build_knn_graph
    W, nb = build_knn_graph(np.empty((0, 8), np.float32), k=10)
    assert isinstance(W, sp.csr_matrix) and W.shape == (0, 0)
    assert nb["distances"].shape == (0, 0) and nb["indices"].shape == (0, 0)
    W, nb = build_knn_graph(latents(1, 8, 0), k=10)
    assert W.shape == (1, 1) and nb["distances"].shape == (1, 0) and nb["indices"].shape == (1, 0)
    W, nb = build_knn_graph(latents(25, 4, 0), k=0)
    assert W.nnz == 0 and nb["indices"].shape == (25, 0)
    W, nb = build_knn_graph(latents(5, 4, 0), k=10)
    assert nb["distances"].shape == (5, 4) and nb["indices"].shape == (5, 4) and W.shape == (5, 5)
    with pytest.raises(ValueError):
        build_knn_graph(latents(20, 4, 0), k=3, sym="bogus")
    with pytest.raises(AssertionError):
        build_knn_graph(np.zeros(7, np.float32), k=3)
    from vqvae_amd.geo.knn_graph_optimized import build_knn_graph_auto
    with pytest.raises(RuntimeError):
        build_knn_graph_auto(latents(20, 4, 0), k=3, force_method="faiss")


def test_structural_properties():
    from vqvae_amd.geo.knn_graph_optimized import build_knn_graph
    z = latents(200, 8, 0)
    W, nb = build_knn_graph(z, k=10)
    assert (W - W.T).nnz == 0 and float(W.diagonal().sum()) == 0.0
    rows = np.arange(200)[:, None]
    assert not np.any(nb["indices"] == rows)
    Wd, _ = build_knn_graph(latents(120, 6, 0), k=8, mode="distance")
    assert np.all(Wd.data >= 0.0)


def test_ragged_sizes_vs_oracle():
    """N not a multiple of the 64-lane step, odd d (zero padded), k+1 up to the 64-lane list."""
    from vqvae_amd.geo.knn_graph_optimized import build_knn_graph
    from oracle import knn as okn
    for N, d, k in ((65, 3, 7), (130, 17, 63), (1000, 33, 20), (777, 100, 5), (64, 16, 63)):
        z = latents(N, d, N + d)
        W, info = build_knn_graph(z, k=k, mode="distance", sym="union")
        Wo, info_o = okn.build_knn_graph(z, k=k, mode="distance", sym="union")
        np.testing.assert_array_equal(info["indices"], info_o["indices"], err_msg=str((N, d, k)))
        np.testing.assert_array_equal(W.indices, Wo.indices)
        np.testing.assert_array_equal(W.indptr, Wo.indptr)
        np.testing.assert_array_equal(W.data, Wo.data)


def test_lists_longer_than_one_wave_vs_oracle():
    """k + 1 > 64 (two and four list registers per lane): the reference's sklearn search has no such limit
    (src/geo/knn_graph_optimized.py:40); ties from duplicated rows are ordered by index like the oracle's."""
    from vqvae_amd.geo.knn_graph_optimized import MAX_NEIGHBORS, build_knn_graph
    from oracle import knn as okn
    for N, d, k in ((700, 8, 64), (1500, 16, 99), (900, 33, 127), (1100, 16, 128), (2000, 5, 200), (300, 24, 255)):
        z = latents(N, d, N + k)
        z[N // 2:N // 2 + 40] = z[:40]                              # exact ties across the register boundary
        W, info = build_knn_graph(z, k=k, mode="distance", sym="mutual")
        Wo, info_o = okn.build_knn_graph(z, k=k, mode="distance", sym="mutual")
        np.testing.assert_array_equal(info["distances"], info_o["distances"], err_msg=str((N, d, k)))
        np.testing.assert_array_equal(info["indices"], info_o["indices"], err_msg=str((N, d, k)))
        np.testing.assert_array_equal(W.indices, Wo.indices)
        np.testing.assert_array_equal(W.data, Wo.data)
        np.testing.assert_array_equal(W.indptr, Wo.indptr)
    assert MAX_NEIGHBORS == 256
    with pytest.raises(ValueError):
        build_knn_graph(latents(400, 8, 1), k=MAX_NEIGHBORS)


def _oracle_rows(z, n_neighbors, form, r0, r1):
    import ctypes
    from oracle import _clib
    io = np.empty((r1 - r0, n_neighbors), np.int64)
    do = np.empty((r1 - r0, n_neighbors), np.float64)
    _clib.lib().oracle_knn(ctypes.c_void_p(z.ctypes.data), z.shape[0], z.shape[1], n_neighbors, form, r0, r1,
                           ctypes.c_void_p(io.ctypes.data), ctypes.c_void_p(do.ctypes.data))
    return io, do


@pytest.mark.parametrize("d,filt", [(16, "1"), (8, "1"), (24, "1"), (64, "1"), (16, "0")])
def test_large_corpus_filter_path_vs_oracle(d, filt):
    """Above 40 000 rows the search runs behind the float32 matrix-core filter (subset thresholds, MFMA scan, exact
    fp64 refinement of the kept candidates); lists and fp64 keys must equal the oracle's, as without the filter --
    expansion form (d > 15), direct form (d <= 15) and a padded dimension (24 -> 32)."""
    import torch
    from vqvae_amd._device import device
    from vqvae_amd.geo.knn_graph_optimized import knn_search_device
    from vqvae_amd import _lib
    n, kq = 45000, 21
    z = latents(n, d, 21)
    _lib.check(_lib.load().geo_set_option(b"knn_filter", int(filt)), "geo_set_option")
    try:
        idx, d2 = knn_search_device(torch.from_numpy(z).to(device()), kq)
    finally:
        _lib.load().geo_set_option(b"knn_filter", 1)
    idx, d2 = idx.cpu().numpy(), d2.cpu().numpy()
    for r0, r1 in ((0, 48), (22000, 22048), (n - 48, n)):
        io, do = _oracle_rows(z, kq, 1 if d > 15 else 0, r0, r1)
        np.testing.assert_array_equal(idx[r0:r1], io)
        np.testing.assert_array_equal(d2[r0:r1], do)
    assert (idx[:, 0] == np.arange(n)).all() and (np.diff(d2, axis=1) >= 0).all()


@pytest.mark.parametrize("n", [40960, 40961, 41215])
def test_filter_path_at_tile_boundaries_vs_oracle(n):
    """The scan reads the corpus in whole tiles of 256 candidates from a zero-padded copy whose padding rows carry a threshold
    nothing falls below: a corpus that is an exact multiple of the tile, one row more, one row less than the next multiple."""
    import torch
    from vqvae_amd._device import device
    from vqvae_amd.geo.knn_graph_optimized import knn_search_device
    kq = 21
    z = latents(n, 16, 31)
    idx, d2 = knn_search_device(torch.from_numpy(z).to(device()), kq)
    idx, d2 = idx.cpu().numpy(), d2.cpu().numpy()
    for r0, r1 in ((0, 32), (n - 300, n - 268), (n - 32, n)):
        io, do = _oracle_rows(z, kq, 1, r0, r1)
        np.testing.assert_array_equal(idx[r0:r1], io)
        np.testing.assert_array_equal(d2[r0:r1], do)
    assert (idx[:, 0] == np.arange(n)).all() and (idx >= 0).all() and (idx < n).all()


@pytest.mark.parametrize("filt", [1, 2])
def test_filter_margins_hold_far_from_the_origin(filt):
    """Latents shifted by +40 in every coordinate: norms (25 600) dwarf the neighbour distances (~10), so the filter's
    rounding margin -- relative to |x|^2 + |y|^2, wider for the bf16 scan (1) than for the float32 one (2) -- decides
    what is kept; the lists must still be the exact ones, and a coordinate scale of 1e3 must not change that either."""
    import torch
    from vqvae_amd._device import device
    from vqvae_amd.geo.knn_graph_optimized import knn_search_device
    from vqvae_amd import _lib
    n, kq = 42000, 21
    for scale, shift in ((1.0, 40.0), (1000.0, 0.0), (1e-3, 0.05)):
        z = (latents(n, 16, 5) * np.float32(scale) + np.float32(shift)).astype(np.float32)
        _lib.check(_lib.load().geo_set_option(b"knn_filter", int(filt)), "geo_set_option")
        try:
            idx, d2 = knn_search_device(torch.from_numpy(z).to(device()), kq)
        finally:
            _lib.load().geo_set_option(b"knn_filter", 1)
        idx, d2 = idx.cpu().numpy(), d2.cpu().numpy()
        for r0, r1 in ((0, 40), (30000, 30040)):
            io, do = _oracle_rows(z, kq, 1, r0, r1)
            np.testing.assert_array_equal(idx[r0:r1], io, err_msg=str((scale, shift)))
            np.testing.assert_array_equal(d2[r0:r1], do)


def test_large_corpus_row_ranges_equal_the_full_search():
    """Query-row shards (the multi-rank layout, parallel.sharded_knn) of a corpus that takes the filter path."""
    import torch
    from vqvae_amd._device import device
    from vqvae_amd.geo.knn_graph_optimized import knn_search_device
    n, kq = 45000, 21
    z = torch.from_numpy(latents(n, 16, 23)).to(device())
    idx, d2 = knn_search_device(z, kq)
    for r0, r1 in ((0, 11250), (11250, 30001), (30001, n)):
        i_s, d_s = knn_search_device(z, kq, r0, r1)
        assert torch.equal(i_s, idx[r0:r1]) and torch.equal(d_s, d2[r0:r1])


def test_large_corpus_with_masses_of_duplicates_falls_back():
    """3 000 copies of one point overflow the candidate lists of the filter (cap 1024): the search must notice and
    answer with the exact scan.  Ties among exact duplicates: (distance, index) order."""
    import torch
    from vqvae_amd._device import device
    from vqvae_amd.geo.knn_graph_optimized import knn_search_device
    n, d, kq = 42000, 16, 21
    z = latents(n, d, 22)
    z[5000:8000] = z[5000]
    idx, d2 = knn_search_device(torch.from_numpy(z).to(device()), kq)
    idx, d2 = idx.cpu().numpy(), d2.cpu().numpy()
    for r0, r1 in ((0, 32), (5000, 5032), (7968, 8000)):
        io, do = _oracle_rows(z, kq, 1, r0, r1)
        np.testing.assert_array_equal(d2[r0:r1], do)
        np.testing.assert_array_equal(idx[r0:r1], io)
    assert (d2[5000:8000] == 0).all() and (idx[5000:8000, 0] == 5000).all()     # lowest indices among the copies first


@pytest.mark.parametrize("copies", [30, 100])
def test_filter_path_with_ties_at_the_kq_th_distance(copies):
    """Groups of identical latents small enough for the candidate lists (cap 1 024) but tying at the kq-th distance: the
    refinement's selection (quickselect for the kq-th distance, then the entries at or below it ordered by (distance, index))
    must cut the tie by index like the oracle -- 30 copies: more entries at the threshold than kq, one per lane; 100 copies:
    more than the 64 lanes hold, the one-by-one extraction takes over."""
    import torch
    from vqvae_amd._device import device
    from vqvae_amd.geo.knn_graph_optimized import knn_search_device
    n, d, kq = 42000, 16, 21
    z = latents(n, d, 29)
    for g in range(20):                                         # 20 groups of `copies` identical rows, scattered
        rows = 1000 + 2000 * g + 7 * np.arange(copies)
        z[rows] = z[rows[0]]
    idx, d2 = knn_search_device(torch.from_numpy(z).to(device()), kq)
    idx, d2 = idx.cpu().numpy(), d2.cpu().numpy()
    for r0, r1 in ((990, 1022), (1000 + 7 * (copies - 1) - 8, 1000 + 7 * (copies - 1) + 8), (39000, 39032), (n - 32, n)):
        io, do = _oracle_rows(z, kq, 1, r0, r1)
        np.testing.assert_array_equal(d2[r0:r1], do)
        np.testing.assert_array_equal(idx[r0:r1], io)
    first = 1000 + 7 * np.arange(kq)
    np.testing.assert_array_equal(idx[1000 + 7 * (copies - 1)], first)          # the last copy: the kq lowest-indexed copies
    assert (d2[1000 + 7 * (copies - 1)] == 0).all()


def test_lcc_and_connectivity(golden):
    from vqvae_amd.geo.knn_graph_optimized import (analyze_graph_connectivity, build_knn_graph,
                                                   largest_connected_component)
    from oracle import knn as okn
    gs = golden("sssp")
    W, _ = build_knn_graph(latents(240, 12, 1), k=1, mode="distance", sym="mutual")
    mask = largest_connected_component(W)
    assert mask.dtype == bool
    np.testing.assert_array_equal(mask, gs["disc/lcc"])
    stats = analyze_graph_connectivity(W)
    ref = okn.analyze_graph_connectivity(W)
    for key in ("n_nodes", "n_edges", "n_components", "largest_component_size"):
        assert stats[key] == ref[key], key
    Wc, _ = build_knn_graph(latents(300, 8, 5), k=10, sym="union")
    assert largest_connected_component(Wc).all()
    # scipy label order: components numbered by their lowest node
    from vqvae_amd.geo.knn_graph_optimized import _undirected_structure, connected_components_device
    from vqvae_amd._device import device
    n, labels = connected_components_device(_undirected_structure(W, device()))
    no, lo = okn.connected_components(W)
    assert n == no
    np.testing.assert_array_equal(labels.cpu().numpy(), lo)


def test_upper_edges_and_reweight_vs_oracle():
    import torch
    from oracle import pipeline as op
    from vqvae_amd._device import DeviceCSR, device
    from vqvae_amd.geo.knn_graph_optimized import (build_knn_graph, compact_device, reweight_device,
                                                   upper_edges_device)
    W, _ = build_knn_graph(latents(500, 16, 9), k=6, mode="connectivity", sym="union")
    dev = device()
    G = DeviceCSR.from_scipy(W, dev)
    src, dst, entry_edge = upper_edges_device(G)
    edges = op.upper_edges(W)
    np.testing.assert_array_equal(src.cpu().numpy(), edges[:, 0])
    np.testing.assert_array_equal(dst.cpu().numpy(), edges[:, 1])
    lengths = np.random.RandomState(0).rand(len(edges)).astype(np.float32)
    lengths[::17] = 0.0                                     # zero-length edges vanish from U + U^T
    Wg = reweight_device(G, entry_edge, torch.from_numpy(lengths).to(dev))
    Wz, _ = compact_device(Wg, None, drop_zero=True)
    ref = op.reweighted_graph(500, edges, lengths)
    ref.sort_indices()
    got = Wz.to_scipy()
    np.testing.assert_array_equal(got.indptr, ref.indptr)
    np.testing.assert_array_equal(got.indices, ref.indices)
    np.testing.assert_array_equal(got.data, ref.data)
    # node compaction = W[mask][:, mask]
    mask = np.random.RandomState(1).rand(500) > 0.3
    Wm, new_index = compact_device(Wz, torch.from_numpy(mask).to(dev), drop_zero=False)
    refm = ref[mask][:, mask]
    refm.sort_indices()
    gotm = Wm.to_scipy()
    np.testing.assert_array_equal(gotm.indptr, refm.indptr)
    np.testing.assert_array_equal(gotm.indices, refm.indices)
    np.testing.assert_array_equal(gotm.data, refm.data)


def test_cosine_and_other_metrics_vs_reference(golden):
    """metric != "euclidean" ("never an error"): cosine on the HIP search over the unit rows, manhattan /
    chebyshev through torch.cdist on the GPU; reference graphs in tests/golden/knn_metrics.npz.  An unknown metric
    name is a ValueError, as sklearn raises."""
    from vqvae_amd.geo.knn_graph_optimized import build_knn_graph_auto
    g = golden("knn_metrics")
    for name, (N, d, seed) in {"g16": (2048, 16, 0), "g32": (512, 32, 1)}.items():
        z = latents(N, d, seed)
        for metric in ("cosine", "manhattan", "chebyshev"):
            if metric != "cosine" and name != "g32":
                continue
            for mode, sym in (("distance", "union"), ("connectivity", "mutual")):
                tag = f"{name}/{metric}/{mode}/{sym}"
                W, info = build_knn_graph_auto(z, k=20, metric=metric, mode=mode, sym=sym)
                W.sort_indices()
                np.testing.assert_array_equal(W.indptr, g[f"{tag}/indptr"], err_msg=tag)
                np.testing.assert_array_equal(W.indices, g[f"{tag}/indices"], err_msg=tag)
                if mode == "distance":
                    ref = g[f"{tag}/data"]
                    assert np.abs(W.data - ref).max() <= 5e-7 * max(1.0, float(ref.max())), tag
            np.testing.assert_array_equal(info["indices"], g[f"{name}/{metric}/nbr_indices"])
            assert info["distances"].dtype == np.float32
    with pytest.raises(ValueError):
        build_knn_graph_auto(latents(64, 8, 0), k=5, metric="no-such-metric")


@pytest.mark.parametrize("metric", ["euclidean", "cosine"])
def test_faiss_semantics_opt_in_equals_the_restated_indexflat(metric):
    """a reference WITH faiss takes build_knn_graph_faiss at every headline size (N >= 50 000).  Its
    semantics -- squared float32 L2 distances / 1 - <x^, y^>, self column dropped only when it leads every row -- on the HIP
    search, opt-in (vqvae_amd.geo.knn_graph_optimized.FAISS_SEMANTICS), against the numpy restatement of IndexFlatL2 /
    IndexFlatIP in oracle/knn.py.  PARITY UNPINNED (faiss is not installed where the fixtures are made)."""
    from oracle import knn as okn
    from vqvae_amd.geo import knn_graph_optimized as kg
    z = latents(700, 16, 41)
    kg.FAISS_SEMANTICS = True
    try:
        for mode, sym in (("distance", "union"), ("connectivity", "mutual")):
            W, info = kg.build_knn_graph_auto(z, k=12, metric=metric, mode=mode, sym=sym, force_method="faiss")
            Wo, io = okn.build_knn_graph_faiss_semantics(z, k=12, metric=metric, mode=mode, sym=sym)
            W.sort_indices(); Wo.sort_indices()
            np.testing.assert_array_equal(info["indices"], io["indices"])
            assert info["distances"].dtype == np.float32 and info["indices"].dtype == np.int64
            np.testing.assert_allclose(info["distances"], io["distances"], rtol=2e-6, atol=2e-7)
            np.testing.assert_array_equal(W.indptr, Wo.indptr)
            np.testing.assert_array_equal(W.indices, Wo.indices)
            np.testing.assert_allclose(W.data, Wo.data, rtol=2e-6, atol=2e-7)
        # the automatic choice follows the reference's size threshold once the semantics are switched on
        Wa, ia = kg.build_knn_graph_auto(z, k=5, metric=metric, mode="distance", sym="union", size_threshold=500)
        Wf, _ = kg.build_knn_graph_faiss(z, k=5, metric=metric, mode="distance", sym="union")
        assert (Wa != Wf).nnz == 0
        Ws, _ = kg.build_knn_graph_auto(z, k=5, metric=metric, mode="distance", sym="union", size_threshold=5000)
        assert abs(Ws.data.max() - np.sqrt(Wa.data.max())) < 1e-5 if metric == "euclidean" else True
        with pytest.raises(ValueError):
            kg.build_knn_graph_faiss(z, k=5, metric="manhattan")
    finally:
        kg.FAISS_SEMANTICS = False
    with pytest.raises(RuntimeError):
        kg.build_knn_graph_faiss(z, k=5)


def test_cosine_graph_with_a_zero_latent_follows_sklearn(golden):
    """A latent that is exactly zero is at cosine distance 1.0 from every row in the reference (sklearn leaves a zero
    row un-normalised), not at 0.5 (what |x^ - y^|^2 / 2 on unit rows would say).  Fixture: the reference's own lists for
    every row but the zero one (tests/golden/knn_cosine_zero.npz)."""
    from oracle import synthetic as syn
    from vqvae_amd.geo.knn_graph_optimized import build_knn_graph_auto
    g = golden("knn_cosine_zero")
    z = syn.gauss_latents(400, 16, 31)
    z[17] = 0.0
    W, info = build_knn_graph_auto(z, k=10, metric="cosine", mode="distance", sym="union")
    rows = g["rows"]
    np.testing.assert_array_equal(info["indices"][rows], g["nbr_indices"])
    np.testing.assert_allclose(info["distances"][rows], g["nbr_distances"], rtol=0, atol=3e-7)     # f32 cosine: 1 ulp near 1
    assert not (info["indices"][rows] == 17).any()                       # nobody is near the zero latent
    np.testing.assert_allclose(info["distances"][17], g["zero_row_distances"], atol=1e-7)      # all 1.0
    assert W[17].nnz == 10 and np.allclose(W[17].data, 1.0)


@pytest.mark.parametrize("d,shift", [(16, 0.0), (32, 0.0), (16, 25.0)])
def test_two_level_thresholds_from_200k_rows_vs_oracle_and_one_level(d, shift):
    """From 200 000 rows on the filter's thresholds come from a filtered pass themselves (every 256th row exact, then the
    matrix-core scan + fp64 refinement over the every-16th-row subset): the lists and fp64 keys must equal the oracle's on
    sampled rows and, everywhere, those of the one-level scheme (float32 scan, `knn_filter = 2`) -- also far from the
    origin, where the margins are wide and the lists long."""
    import torch
    from vqvae_amd._device import device
    from vqvae_amd.geo.knn_graph_optimized import knn_search_device
    from vqvae_amd import _lib
    n, kq = 210000, 21
    z = (latents(n, d, 33) + np.float32(shift)).astype(np.float32)
    out = {}
    try:
        for filt in (1, 2):
            _lib.check(_lib.load().geo_set_option(b"knn_filter", filt), "geo_set_option")
            idx, d2 = knn_search_device(torch.from_numpy(z).to(device()), kq)
            out[filt] = (idx.cpu().numpy(), d2.cpu().numpy())
    finally:
        _lib.load().geo_set_option(b"knn_filter", 1)
    np.testing.assert_array_equal(out[1][0], out[2][0])
    np.testing.assert_array_equal(out[1][1], out[2][1])
    # a rank's share of the query rows (sharded kNN): the same rows of the full answer
    ia, da = knn_search_device(torch.from_numpy(z).to(device()), kq, 70001, 140777)
    np.testing.assert_array_equal(ia.cpu().numpy(), out[1][0][70001:140777])
    np.testing.assert_array_equal(da.cpu().numpy(), out[1][1][70001:140777])
    for r0, r1 in ((0, 24), (104000, 104024), (n - 24, n)):
        io, do = _oracle_rows(z, kq, 1 if d > 15 else 0, r0, r1)
        np.testing.assert_array_equal(out[1][0][r0:r1], io)
        np.testing.assert_array_equal(out[1][1][r0:r1], do)

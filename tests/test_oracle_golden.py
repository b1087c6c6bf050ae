"""The oracle (oracle/, CPU restatement) against the golden vectors generated from the reference import
(tests/golden/*.npz, oracle/gen_golden.py) and against the reference's own known-answer tests
(reference tests/test_geo_shortest_paths.py:37-46,56-71,82-90).  Runs on CPU."""
import numpy as np
import pytest
import torch
from scipy import sparse

from conftest import clustered_latents, csr_from_golden, latents
from oracle import kmedoids as ok
from oracle import knn as okn
from oracle import metric as om
from oracle import pipeline as op
from oracle import sssp as osp

KNN_CASES = {"g16": (2048, 16, 0), "g32": (512, 32, 1), "g64": (300, 64, 2), "g8": (400, 8, 3)}


def line_graph(N, w=1.0):
    rows, cols, data = [], [], []
    for i in range(N - 1):
        rows += [i, i + 1]
        cols += [i + 1, i]
        data += [w, w]
    return sparse.csr_matrix((data, (rows, cols)), shape=(N, N), dtype=np.float32)


@pytest.mark.parametrize("name", ["g32", "g64", "g8"])
def test_knn_graph_structure_equals_reference(golden, name):
    g = golden("knn")
    N, d, seed = KNN_CASES[name]
    z = latents(N, d, seed)
    for k in (1, 5, 20):
        for mode in ("connectivity", "distance"):
            for sym in ("union", "mutual"):
                tag = f"{name}/k{k}/{mode}/{sym}"
                W, info = okn.build_knn_graph_auto(z, k=k, mode=mode, sym=sym)
                W.sort_indices()
                np.testing.assert_array_equal(W.indptr, g[f"{tag}/indptr"], err_msg=tag)
                np.testing.assert_array_equal(W.indices, g[f"{tag}/indices"], err_msg=tag)
                if mode == "distance":
                    ref = g[f"{tag}/data"]
                    assert np.max(np.abs(W.data.astype(np.float64) - ref) / np.spacing(ref)) <= 1.0
        np.testing.assert_array_equal(info["indices"], g[f"{name}/k{k}/nbr_indices"])


def test_knn_c1_graph_equals_reference(golden):
    g = golden("knn")
    W, _ = okn.build_knn_graph_auto(latents(2048, 16, 0), k=20, mode="connectivity", sym="union")
    W.sort_indices()
    np.testing.assert_array_equal(W.indptr, g["g16/k20/connectivity/union/indptr"])
    np.testing.assert_array_equal(W.indices, g["g16/k20/connectivity/union/indices"])


def test_knn_duplicates_differ_only_at_ties(golden):
    g = golden("knn")
    z = clustered_latents(256, 16, 4)
    dup_nodes = {3, 5, 15, 16, 17, 128, 255}
    W, info = okn.build_knn_graph_auto(z, k=5, mode="connectivity", sym="union")
    Wr = csr_from_golden(g, "dup16/k5/connectivity/union", 256, with_data=False)
    diff = abs(W - Wr).tocoo()
    assert all((r in dup_nodes) or (c in dup_nodes) for r, c, v in zip(diff.row, diff.col, diff.data) if v != 0)


def test_sssp_known_answers_from_reference_tests():
    D = osp.dijkstra_multi_source(line_graph(5), [0, 2])
    np.testing.assert_array_equal(D[0], np.array([0, 1, 2, 3, 4], np.float32))
    np.testing.assert_array_equal(D[1], np.array([2, 1, 0, 1, 2], np.float32))
    W = sparse.csr_matrix((np.array([1, 1, 10, 10], np.float32), ([0, 1, 1, 2], [1, 0, 2, 1])), shape=(3, 3))
    assert osp.dijkstra_multi_source(W, [0])[0, 2] == 11.0
    assert osp.dijkstra_multi_source(W, [0], unweighted=True)[0, 2] == 2.0
    Wd = sparse.block_diag((line_graph(3), line_graph(4)), format="csr", dtype=np.float32)
    D = osp.dijkstra_multi_source(Wd, [0, 3])
    assert np.isinf(D[0, 3:]).all() and np.isinf(D[1, :3]).all()
    D, P = osp.dijkstra_multi_source(line_graph(4), [0], return_predecessors=True)
    assert D.dtype == np.float32 and P.dtype == np.int32 and list(P[0]) == [-9999, 0, 1, 2]
    with pytest.raises(ValueError):
        osp.dijkstra_multi_source(line_graph(3), [])
    with pytest.raises(TypeError):
        osp.ensure_valid_graph(np.zeros((2, 2)))


def test_sssp_golden_bit_exact(golden):
    gk, gs = golden("knn"), golden("sssp")
    W = csr_from_golden(gk, "g16/k20/distance/union", 2048)
    D, P = osp.dijkstra_multi_source(W, gs["g16/sources"], return_predecessors=True)
    np.testing.assert_array_equal(D, gs["g16/D"])
    np.testing.assert_array_equal(P, gs["g16/P"])
    np.testing.assert_array_equal(osp.dijkstra_multi_source(W, gs["g16/sources"][:3], unweighted=True),
                                  gs["g16/D_unweighted"])
    Wt = sparse.triu(W).tocsr()
    np.testing.assert_array_equal(osp.dijkstra_multi_source(Wt, [0, 5], directed=True), gs["g16/D_triu_directed"])
    np.testing.assert_array_equal(osp.dijkstra_multi_source(Wt, [0, 5]), gs["g16/D_triu_undirected"])
    Wm, _ = okn.build_knn_graph(latents(240, 12, 1), k=1, mode="distance", sym="mutual")
    np.testing.assert_array_equal(osp.dijkstra_multi_source(Wm, [0, 10]), gs["disc/D"])
    np.testing.assert_array_equal(okn.largest_connected_component(Wm), gs["disc/lcc"])


def test_kmedoids_golden_and_single_pass(golden):
    gk, gm = golden("knn"), golden("kmedoids")
    Wd = csr_from_golden(gk, "g16/k20/distance/union", 2048)
    Wm, _ = okn.build_knn_graph(latents(240, 12, 1), k=1, mode="distance", sym="mutual")
    for gname, W in (("g16", Wd), ("disc", Wm)):
        for K in (1, 8, 64):
            for init in ("kpp", "random"):
                tag = f"{gname}/K{K}/{init}/s42"
                med, assign, qe = ok.fit_kmedoids_optimized(W, K=K, init=init, seed=42)
                np.testing.assert_array_equal(med, gm[f"{tag}/medoids"])
                np.testing.assert_array_equal(assign, gm[f"{tag}/assign"])
                gq = float(gm[f"{tag}/qe"])
                assert qe == gq or (np.isinf(qe) and np.isinf(gq))
                if init == "kpp":
                    m2, a2, q2 = ok.fit_kmedoids_single_pass(W, K=K, seed=42)
                    np.testing.assert_array_equal(m2, med)
                    np.testing.assert_array_equal(a2, assign)
                    assert q2 == qe or (np.isinf(q2) and np.isinf(qe))
    with pytest.raises(ValueError):
        ok.fit_kmedoids_optimized(Wd, K=3, init="bogus")


@pytest.mark.parametrize("name,cfg", [("fm_batch", (16, 1, 28, "batch", 10)), ("fm_none", (16, 1, 28, "none", 11)),
                                      ("fm_group", (16, 1, 28, "group", 12)), ("cf_batch", (32, 3, 32, "batch", 13))])
def test_metric_closed_form_vs_reference(golden, name, cfg):
    d, cout, size, norm, seed = cfg
    g = golden("metric")
    sd = om.make_decoder_state(seed, d, cout, norm_type=norm)
    r = np.random.RandomState(100 + seed)
    zs = r.randn(2048, d).astype(np.float32)
    ze = (zs + 0.3 * r.randn(2048, d)).astype(np.float32)
    for training in (True, False):
        L = om.edge_lengths(sd, norm, size, zs[:512], ze[:512], batch_size=512, training=training).numpy()
        ref = g[f"{name}/train{int(training)}/bs512"][:512]
        rel = np.abs(L - ref) / np.abs(ref)
        assert np.mean(rel <= 1e-5) >= 0.998, rel.max()      # ReLU-boundary outliers allowed (SURVEY finding 9)


def test_pipeline_c1_equals_reference_cli(golden):
    g = golden("cli")
    d, cout, size, seed, n_img = [int(x) for x in g["c1_fm/meta"]]
    sd = om.make_decoder_state(seed, d, cout, norm_type="batch")
    z4 = np.random.RandomState(seed).randn(n_img * 16, d).astype(np.float32).reshape(n_img, 4, 4, d)
    z4 = np.ascontiguousarray(np.transpose(z4, (0, 3, 1, 2)))
    # the JVP stage is the slow part on CPU: reuse the reference's graph weights and check what follows
    W = sparse.csr_matrix((g["c1_fm/data"], g["c1_fm/indices"], g["c1_fm/indptr"]), shape=(2048, 2048))
    med, assign, qe = ok.fit_kmedoids_optimized(W, K=64, init="kpp", seed=42)
    np.testing.assert_array_equal(med.astype(np.int32), g["c1_fm/medoid_indices"])
    np.testing.assert_array_equal(assign.astype(np.int32).reshape(n_img, 4, 4), g["c1_fm/codes"])
    np.testing.assert_array_equal(op.flatten_latents(z4)[med], g["c1_fm/z_medoid"])
    We, _ = okn.build_knn_graph_auto(op.flatten_latents(z4), k=20, mode="connectivity", sym="union")
    We.sort_indices()
    np.testing.assert_array_equal(We.indptr, g["c1_fm/indptr"])
    np.testing.assert_array_equal(We.indices, g["c1_fm/indices"])
    edges = op.upper_edges(We)
    L = om.edge_lengths(sd, "batch", size, op.flatten_latents(z4)[edges[:1024, 0]], op.flatten_latents(z4)[edges[:1024, 1]],
                        batch_size=512, training=True).numpy()
    ref = np.asarray(W[edges[:1024, 0], edges[:1024, 1]]).ravel()
    assert np.mean(np.abs(L - ref) / ref <= 1e-5) >= 0.998


def test_zero_weight_fallback_chains_equal_reference(golden):
    """Repeated uniform fallbacks in one chain (kmeans_optimized.py:62-69): the oracle follows the reference's
    RandomState stream (tests/golden/kmedoids_zero.npz holds the reference's outputs)."""
    from oracle import synthetic as syn
    g = golden("kmedoids_zero")
    for i, (sizes, K, seed) in enumerate(syn.ZERO_CASES):
        W = syn.zero_clusters(sizes)
        for fit in (ok.fit_kmedoids_optimized, lambda W, K, seed: ok.fit_kmedoids_single_pass(W, K=K, seed=seed)):
            med, assign, qe = fit(W, K=K, seed=seed)
            np.testing.assert_array_equal(med, g[f"case{i}/medoids"])
            np.testing.assert_array_equal(assign, g[f"case{i}/assign"])
            assert qe == float(g[f"case{i}/qe"])


def test_cf64_edge_lengths_vs_reference(golden):
    """latent_dim 64, 32-px decoder (BASELINE config 3 as stated): closed-form f32 oracle vs the reference's lengths."""
    g = golden("metric_cf64")
    d, cout, size, seed, zseed, E = (int(v) for v in g["cf64_batch/meta"])
    sd = om.make_decoder_state(seed, d, cout, norm_type="batch")
    r = np.random.RandomState(zseed)
    zs = r.randn(E, d).astype(np.float32)
    ze = (zs + 0.3 * r.randn(E, d)).astype(np.float32)
    for training, bs in ((True, 512), (False, 100)):
        L = om.edge_lengths(sd, "batch", size, zs, ze, batch_size=bs, training=training).numpy()
        ref = g[f"cf64_batch/train{int(training)}/bs{bs}"]
        rel = np.abs(L - ref) / ref
        assert np.mean(rel <= 1e-5) >= 0.999 and np.quantile(rel, 0.99) < 2e-6, rel.max()


def test_c2_fixture_structure_and_first_draws(golden):
    """Full-size fixture (tests/golden/c2_formula.npz, reference outputs at N=60 000, K=512): the oracle's kNN structure
    hashes to the reference's and its first k-means++ draws on the formula weights are the reference's (the whole
    512-centre chain is compared on the GPU box, where the oracle's 512 solves take seconds)."""
    from oracle import synthetic as syn
    g = golden("c2_formula")
    n, d, k, K, seed, nnz = (int(v) for v in g["meta"])
    W, _ = okn.build_knn_graph_auto(syn.gauss_latents(n, d, 0), k=k, mode="connectivity", sym="union")
    W.sort_indices()
    assert W.nnz == nnz
    np.testing.assert_array_equal(syn.digest(W.indptr.astype(np.int32)), g["indptr_sha256"])
    np.testing.assert_array_equal(syn.digest(W.indices.astype(np.int32)), g["indices_sha256"])
    rows = np.repeat(np.arange(n), np.diff(W.indptr))
    Wf = sparse.csr_matrix((syn.formula_weights(rows, W.indices), W.indices, W.indptr), shape=(n, n))
    assert ok.kpp_initialization_graph(Wf, 6, seed=seed) == g["medoids"][:6].tolist()


def test_dense_closed_form_equals_layer_by_layer(golden):
    """oracle.metric.edge_lengths_dense (every transposed convolution on the 1x1 latent image as a matrix, stacks of
    chunks at once: the full-size fp64 checker) against the layer-by-layer closed form and the reference's lengths."""
    for name, (d, cout, size, seed) in {"fm_batch": (16, 1, 28, 10), "cf_batch": (32, 3, 32, 13)}.items():
        sd = om.make_decoder_state(seed, d, cout, norm_type="batch")
        r = np.random.RandomState(100 + seed)
        zs = r.randn(2048, d).astype(np.float32)
        ze = (zs + 0.3 * r.randn(2048, d)).astype(np.float32)
        for bs, E in ((512, 2048), (100, 2048), (512, 1300)):
            a = om.edge_lengths(sd, "batch", size, zs[:E], ze[:E], batch_size=bs, training=True, dtype=torch.float64).numpy()
            b = om.edge_lengths_dense(sd, size, zs[:E], ze[:E], batch_size=bs).numpy()
            np.testing.assert_allclose(a, b, rtol=2e-7)
        ref = golden("metric")[f"{name}/train1/bs512"]
        rel = np.abs(om.edge_lengths_dense(sd, size, zs, ze, batch_size=512).numpy() - ref) / ref
        assert np.quantile(rel, 0.99) < 2e-6


def test_cosine_knn_graph_equals_reference(golden):
    """metric="cosine" (sklearn cosine_distances behind knn_graph_optimized.py:40): structure identical, weights within
    float32 rounding of sklearn's float32 pipeline."""
    g = golden("knn_metrics")
    for name, (N, d, seed) in {"g16": (2048, 16, 0), "g32": (512, 32, 1)}.items():
        z = latents(N, d, seed)
        for mode, sym in (("distance", "union"), ("connectivity", "mutual")):
            tag = f"{name}/cosine/{mode}/{sym}"
            W, info = okn.build_knn_graph_auto(z, k=20, metric="cosine", mode=mode, sym=sym)
            W.sort_indices()
            np.testing.assert_array_equal(W.indptr, g[f"{tag}/indptr"], err_msg=tag)
            np.testing.assert_array_equal(W.indices, g[f"{tag}/indices"], err_msg=tag)
            if mode == "distance":
                assert np.abs(W.data - g[f"{tag}/data"]).max() <= 5e-7
        np.testing.assert_array_equal(info["indices"], g[f"{name}/cosine/nbr_indices"])


def test_voronoi_iteration_oracle_properties():
    """Extension without a reference: the numpy restatement against brute force on a small graph."""
    import math
    from oracle import kmedoids as okm
    from oracle import knn as okn
    z = np.random.RandomState(3).randn(300, 6).astype(np.float32)
    W, _ = okn.build_knn_graph(z, k=6, mode="distance", sym="union")
    mask = okn.largest_connected_component(W)
    W = W[mask][:, mask].tocsr()
    n = W.shape[0]
    D = okm.all_pairs(W)
    np.testing.assert_array_equal(D, D.T)                                   # undirected graph
    med0 = np.random.RandomState(0).choice(n, 9, replace=False)
    assign0 = np.argmin(D[med0], axis=0)
    new, cost = okm.medoid_update(D, assign0, med0, power=2)
    for c in range(9):
        members = np.flatnonzero(assign0 == c)
        exact = [math.fsum(float(D[i, j]) ** 2 for j in members) for i in members]
        np.testing.assert_allclose(cost[members], exact, rtol=1e-13)
        assert new[c] == members[int(np.argmin(cost[members]))]
    med, assign, qe, hist = okm.voronoi_iteration(W, med0, max_iter=20, D=D)
    assert all(b <= a for a, b in zip(hist, hist[1:])) and qe == hist[-1]
    again, _ = okm.medoid_update(D, assign, med, power=2)                   # converged: a further update is the identity
    np.testing.assert_array_equal(again, med)
    X = np.random.RandomState(1).rand(7, 333)
    np.testing.assert_allclose(okm._tree_sum_rows(X), X.sum(axis=1), rtol=1e-14)


def test_held_out_assignment_oracle_against_explicit_graph_extension():
    """assign_new_latents (extension) = shortest paths on the graph with the new node and its k attachment
    edges added explicitly (one new node at a time, so new nodes never see each other)."""
    import scipy.sparse as sp
    from oracle import kmedoids as okm
    from oracle import knn as okn
    from oracle import pipeline as opl
    from oracle import sssp as osp
    rs = np.random.RandomState(5)
    z = rs.randn(260, 5).astype(np.float32)
    W, _ = okn.build_knn_graph(z[:240], k=6, mode="distance", sym="union")
    mask = okn.largest_connected_component(W)
    W = W[mask][:, mask].tocsr()
    zg, znew = z[:240][mask], z[240:]
    n = W.shape[0]
    med, _, _ = okm.fit_kmedoids_optimized(W, K=7, init="kpp", seed=0)
    codes, dist, idx, lengths = opl.assign_new_latents(znew, zg, W, med, k=4)
    brute = np.sqrt(((znew[:, None, :].astype(np.float64) - zg[None, :, :].astype(np.float64)) ** 2).sum(-1))
    np.testing.assert_array_equal(np.sort(idx, axis=1), np.sort(np.argsort(brute, axis=1, kind="stable")[:, :4], axis=1))
    for v in range(znew.shape[0]):
        rows = np.concatenate([np.full(4, n), idx[v]])
        cols = np.concatenate([idx[v], np.full(4, n)])
        A = sp.csr_matrix((np.concatenate([lengths[v], lengths[v]]), (rows, cols)), shape=(n + 1, n + 1))
        Wx = (sp.block_diag([W, sp.csr_matrix((1, 1))]).tocsr() + A).tocsr().astype(np.float32)
        d = osp.dijkstra_multi_source(Wx, np.asarray(med))[:, n]
        np.testing.assert_allclose(dist[v], d.min(), rtol=1e-6)
        assert d[codes[v]] <= d.min() * (1 + 1e-6)


def test_pam_restatement_is_self_consistent():
    """oracle/kmedoids.py's PAM (extension without a reference): the brute-force swap pass agrees with the FastPAM1 identity
    the GPU kernel uses, a swap changes the total cost by exactly its delta, and PAM never raises the cost."""
    from oracle import kmedoids as ok
    r = np.random.RandomState(0)
    n, K = 90, 4
    P = r.rand(n, 3)
    D = np.sqrt(((P[:, None] - P[None]) ** 2).sum(-1)).astype(np.float32)
    med = r.choice(n, K, replace=False)
    for power in (1, 2):
        delta, i, x = ok.pam_swap_pass(D, med, power)
        Dp = D.astype(np.float64) ** power
        rows = Dp[med]
        near, c1 = rows.argmin(0), rows.min(0)
        r2 = rows.copy()
        r2[near, np.arange(n)] = np.inf
        c2 = r2.min(0)
        c = Dp[x]
        shared = np.minimum(c - c1, 0)
        own = (np.minimum(c, c2) - c1 - shared)[near == i].sum()
        assert abs(shared.sum() + own - delta) < 1e-9
        trial = med.copy()
        trial[i] = x
        assert abs(ok.total_cost(D, trial, power) - ok.total_cost(D, med, power) - delta) < 1e-9
    med2, assign, hist = ok.pam(D, med, power=2, max_swaps=30)
    assert all(b < a for a, b in zip(hist, hist[1:])) and len(set(med2.tolist())) == K
    assert (assign == np.argmin(D[med2], axis=0)).all()


def test_faiss_semantics_restatement_shapes_and_self_column():
    """oracle/knn.py: build_knn_graph_faiss_semantics (IndexFlatL2 / IndexFlatIP restated; unpinned): squared float32
    distances, self column dropped when it leads every row, symmetric graph without diagonal."""
    from oracle import knn as okn
    z = latents(120, 8, 4)
    W, info = okn.build_knn_graph_faiss_semantics(z, k=5, metric="euclidean", mode="distance", sym="union")
    assert info["indices"].shape == (120, 5) and info["distances"].dtype == np.float32
    d2 = ((z[:, None, :].astype(np.float64) - z[None]) ** 2).sum(-1)
    np.testing.assert_allclose(info["distances"], np.sort(d2, axis=1)[:, 1:6], rtol=1e-6)
    assert (W - W.T).nnz == 0 and W.diagonal().sum() == 0
    Wc, ic = okn.build_knn_graph_faiss_semantics(z, k=5, metric="cosine", mode="connectivity", sym="mutual")
    assert set(np.unique(Wc.data)) <= {1.0} and ic["distances"].min() > -1e-6
    with pytest.raises(ValueError):
        okn.build_knn_graph_faiss_semantics(z, k=5, metric="manhattan")

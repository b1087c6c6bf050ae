"""Parity at BASELINE.json's full sizes on the MI355X.

C2 (60 000 latents, d=16, k=20, K=512): the whole k-means++ chain / assignment / QE against the oracle's
single-pass chain on the GPU's own graph (all 512 medoids, all 60 000 assignments), every BatchNorm chunk of
the JVP against the fp64 closed form, the reference's own C2 results (tests/golden/c2_formula.npz: reference
fit_kmedoids_optimized on formula weights; tests/golden/c2_cli.npz: the reference CLI end to end).
C3 as BASELINE states it (50 000 latents, d=64, 32-px decoder, K=512) and C4 on one GPU (1 M latents, K=1024).
"""
import ctypes
import json
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

N, D, KNN, KMED = 60000, 16, 20, 512
TOL = 1e-5                       # edge lengths within 1e-5 relative ...
GATE = 0.999                     # ... for >= 99.9 % of the edges
# Which chunks may miss the 1e-5 gate is decided by a CRITERION, not by index ("... or flagged ReLU-boundary
# (|pre-activation| < 1e-6)").  A pre-activation within float32 rounding reach of zero makes that sample's ReLU mask
# implementation-dependent (any two float32 evaluations, the reference's included, may disagree with fp64 there), and in
# train mode the flipped tangent enters the next layer's batch means, so it can move EVERY edge of its chunk.  Measured on
# the CPU with the float32 closed form at C2 (1 848 chunks): all 1 323 edges beyond 1e-5 of fp64 sit in chunks whose
# smallest |pre-activation| is below 1e-6; the chunks above 3e-6 have max error 1.2e-6.  |batch mean| / batch std, the
# suspect named in round 2, is at most 0.95 in every chunk: BatchNorm conditioning is NOT the cause (recorded as R).
NEAR0 = 1e-6
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _record(name, payload):
    """Measured (not asserted) quantities, kept for DESIGN.md: gpurun_out/parity_<name>.json."""
    out = os.path.join(ROOT, "gpurun_out")
    os.makedirs(out, exist_ok=True)
    with open(os.path.join(out, f"parity_{name}.json"), "w") as f:
        json.dump(payload, f, indent=1)
    print(name, json.dumps(payload))


def _decoder(sd, d, cout, size):
    from vqvae_amd.spatial_decoder import SpatialDecoder
    dec = SpatialDecoder(cout, (256, 128, 64), d, size, "batch")
    dec.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()})
    return dec


def _pipeline(n, d, cout, size, K):
    from oracle import metric as om
    from oracle import synthetic as syn
    from vqvae_amd._device import device
    from vqvae_amd.scripts.build_codebook import build_codebook_device
    dev = device()
    z_h = syn.gauss_latents(n, d, 0)
    sd = om.make_decoder_state(0, d, cout, norm_type="batch")
    z = torch.from_numpy(z_h).to(dev)
    res = build_codebook_device(z, _decoder(sd, d, cout, size).to(dev).train(), k=KNN, sym="union", K=K, init="kpp",
                                seed=42, batch_size=512)
    return {"z": z, "z_h": z_h, "sd": sd, "res": res, "dev": dev, "size": size}


@pytest.fixture(scope="module")
def c2():
    return _pipeline(N, D, 1, 28, KMED)


def _columns_strictly_ascending(W):
    d = np.diff(W.indices.astype(np.int64))
    inside = np.ones(len(d), dtype=bool)
    ends = W.indptr[1:-1]
    inside[ends[(ends > 0) & (ends < W.nnz)] - 1] = False       # differences that straddle two rows
    return bool((d[inside] > 0).all())


def _knn_rows_vs_oracle(ctx, n, d, spans):
    from oracle import _clib
    from vqvae_amd.geo.knn_graph_optimized import knn_search_device
    idx, d2 = knn_search_device(ctx["z"], KNN + 1)
    idx_h, d2_h = idx.cpu().numpy(), d2.cpu().numpy()
    for r0, r1 in spans:
        io = np.empty((r1 - r0, KNN + 1), np.int64)
        do = np.empty((r1 - r0, KNN + 1), np.float64)
        _clib.lib().oracle_knn(ctypes.c_void_p(ctx["z_h"].ctypes.data), n, d, KNN + 1, 1, r0, r1,
                               ctypes.c_void_p(io.ctypes.data), ctypes.c_void_p(do.ctypes.data))
        np.testing.assert_array_equal(idx_h[r0:r1], io)
        np.testing.assert_array_equal(d2_h[r0:r1], do)          # same fma chains -> bit-equal fp64 keys
    assert (idx_h[:, 0] == np.arange(n)).all() and (d2_h[:, 0] == 0).all()
    assert (np.diff(d2_h, axis=1) >= 0).all()                   # sorted


def _graph_invariants(ctx, n):
    W = ctx["res"]["W_lcc"].to_scipy()
    assert W.shape == (n, n) and (W - W.T).nnz == 0 and W.diagonal().sum() == 0
    assert _columns_strictly_ascending(W)                        # canonical CSR
    assert np.diff(W.indptr).min() >= KNN and ctx["res"]["n_edges"] * 2 == W.nnz
    return W


def _all_chunks_vs_fp64(ctx, name):
    """EVERY BatchNorm chunk against the fp64 closed form (oracle/metric.py's dense-matrix statement of it, run in fp64
    torch on the GPU and tied to the layer-by-layer CPU form on three chunks); returns the per-edge relative errors."""
    from oracle import metric as om
    if "_vs_fp64" in ctx:
        return ctx["_vs_fp64"]
    src, dst = (t.cpu().numpy() for t in ctx["res"]["edges"])
    L = ctx["res"]["edge_lengths"].cpu().numpy()
    assert L.shape == src.shape and np.isfinite(L).all() and (L > 0).all()
    assert (src < dst).all() and (np.diff(src) >= 0).all()       # row-major upper triangle
    zs, ze = ctx["z_h"][src], ctx["z_h"][dst]
    ref64, cond = om.edge_lengths_dense(ctx["sd"], ctx["size"], zs, ze, batch_size=512, dtype=torch.float64,
                                        device="cuda", with_conditioning=True)
    ref64, R, near0 = ref64.numpy(), cond.numpy()[:, 0], cond.numpy()[:, 1]
    n_chunks = (len(src) + 511) // 512
    assert len(R) == n_chunks
    for c in (0, min(917, n_chunks - 2), n_chunks - 1):          # the checker itself: dense GPU fp64 == conv CPU fp64
        sl = slice(c * 512, min((c + 1) * 512, len(src)))
        cpu64 = om.edge_lengths(ctx["sd"], "batch", ctx["size"], zs[sl], ze[sl], batch_size=512, training=True,
                                dtype=torch.float64).numpy()
        np.testing.assert_allclose(ref64[sl], cpu64, rtol=2e-6)  # both are f32-rounded fp64 results
    rel = np.abs(L - ref64) / ref64
    over_per_chunk = np.array([np.mean(rel[c * 512:(c + 1) * 512] > TOL) for c in range(n_chunks)])
    flagged = near0 < NEAR0                                          # a ReLU-boundary sample somewhere in the chunk
    clear_edges = ~np.repeat(flagged, 512)[:len(rel)]
    _record(f"{name}_jvp_vs_fp64", {"edges": int(len(L)), "chunks": int(n_chunks), "edges_over_1e-5": int((rel > TOL).sum()),
                                    "frac_within": float(np.mean(rel <= TOL)), "p99": float(np.quantile(rel, 0.99)),
                                    "max_rel": float(rel.max()), "worst_chunk_frac_over": float(over_per_chunk.max()),
                                    "near0_threshold": NEAR0, "boundary_flagged_chunks": int(flagged.sum()),
                                    "chunks_with_an_edge_over": int((over_per_chunk > 0).sum()),
                                    "chunks_with_an_edge_over_not_flagged": int(((over_per_chunk > 0) & ~flagged).sum()),
                                    "edges_over_1e-5_in_unflagged_chunks": int((rel[clear_edges] > TOL).sum()),
                                    "max_rel_in_unflagged_chunks": float(rel[clear_edges].max()),
                                    "frac_within_flagged_chunks": float(np.mean(rel[~clear_edges] <= TOL)),
                                    "R_max_mean_over_std": float(R.max()), "R_median": float(np.median(R))})
    assert np.mean(rel <= TOL) >= GATE, (int((rel > TOL).sum()), rel.max())
    assert rel[clear_edges].max() <= TOL                            # chunks without a boundary sample: EVERY edge
    assert (~flagged).sum() >= 64
    assert np.quantile(rel, 0.99) < 2e-6
    ctx["_vs_fp64"] = (rel, ref64)
    return rel, ref64


def _full_chain_vs_oracle(ctx, K):
    """The oracle's single-pass chain (K heap-Dijkstra solves + numpy draws) on the GPU's own graph: every medoid,
    every assignment and QE must be equal."""
    from oracle import kmedoids as ok
    res = ctx["res"]
    W = res["W_lcc"].to_scipy()
    med_o, assign_o, qe_o = ok.fit_kmedoids_single_pass(W, K=K, seed=42)
    np.testing.assert_array_equal(res["medoids"], med_o)
    np.testing.assert_array_equal(res["assign_flat"][res["mask_lcc"]], assign_o)
    assert res["qe"] == qe_o
    return W


# ------------------------------------------------------------------------------------------------- C2
def test_c2_knn_rows_vs_oracle_and_graph_invariants(c2, golden):
    from oracle import synthetic as syn
    _knn_rows_vs_oracle(c2, N, D, ((0, 64), (29968, 30032), (N - 64, N)))
    W = _graph_invariants(c2, N)
    g = golden("c2_formula")                                     # the reference's (sklearn) structure at C2
    assert W.nnz == int(g["meta"][5])
    np.testing.assert_array_equal(syn.digest(W.indptr.astype(np.int32)), g["indptr_sha256"])
    np.testing.assert_array_equal(syn.digest(W.indices.astype(np.int32)), g["indices_sha256"])


def test_c2_every_bn_chunk_vs_fp64_and_reference_cli_sample(c2, golden):
    """All 1 848 chunks against fp64, then the reference CLI's own f32 edge lengths at C2 (tests/golden/c2_cli.npz:
    every 61st chunk, the ill-conditioned chunks 0 and 917, and the tail).  In a chunk whose batch has |mean| >> std
    (chunk 0: 17 distinct start points) float32 BatchNorm statistics are themselves only good to ~1e-3, the
    reference's included: the gate against the reference is taken over the edges on which the reference itself is
    within 1e-5 of fp64 (the others are counted and recorded), the gate against fp64 over every edge."""
    _, ref64 = _all_chunks_vs_fp64(c2, "c2")
    g = golden("c2_cli")
    L = c2["res"]["edge_lengths"].cpu().numpy()
    assert len(L) == int(g["meta"][6])
    pick = np.concatenate([np.arange(c * 512, min((c + 1) * 512, len(L))) for c in g["sample_chunks"]])
    got, ref, truth = L[pick], g["sample_lengths"], ref64[pick]
    rel = np.abs(got - ref) / ref
    ref_ok = np.abs(ref - truth) / truth <= TOL
    _record("c2_jvp_vs_reference_sample", {"edges": int(len(ref)), "frac_within": float(np.mean(rel <= TOL)),
                                           "reference_itself_over_1e-5_of_fp64": int((~ref_ok).sum()),
                                           "frac_within_where_reference_accurate": float(np.mean(rel[ref_ok] <= TOL)),
                                           "p99": float(np.quantile(rel, 0.99)), "max_rel": float(rel.max()),
                                           "bit_equal_frac": float(np.mean(got == ref))})
    assert np.mean(ref_ok) > 0.99
    assert np.mean(rel[ref_ok] <= TOL) >= GATE, rel[ref_ok].max()
    assert abs(float(L.astype(np.float64).sum()) / float(g["lengths_sum"]) - 1.0) < 1e-6


def test_c2_eval_mode_per_node_primal_at_full_size(c2, request):
    """The C2 graph's 946 059 edges with the decoder in EVAL mode (fixed BatchNorm statistics): the per-node primal path
    (one primal pass over the 60 000 latents, tangent-only edge slots; default) against the per-edge-end path bit for bit,
    and against the fp64 closed form (torch fp64 on the GPU) at the usual gate."""
    from oracle import metric as om
    from vqvae_amd import _lib
    from vqvae_amd.geo.riemannian_metric import edge_lengths_graph_device
    from vqvae_amd.spatial_decoder import DecoderExport
    src, dst = c2["res"]["edges"]
    dec = _decoder(c2["sd"], D, 1, c2["size"]).to(c2["dev"]).eval()
    ex = DecoderExport(dec, c2["dev"])
    def restore():
        _lib.load().geo_set_option(b"jvp_per_node", 1)
        _lib.load().geo_set_option(b"jvp_node_jacobian", 1)
    request.addfinalizer(restore)
    _lib.check(_lib.load().geo_set_option(b"jvp_node_jacobian", 0), "geo_set_option")
    out = {}
    for mode in (1, 0):
        _lib.check(_lib.load().geo_set_option(b"jvp_per_node", mode), "geo_set_option")
        out[mode] = edge_lengths_graph_device(ex, c2["z"], src, dst, 512).cpu().numpy()
    np.testing.assert_array_equal(out[1], out[0])                 # all 946 059 edges, bit for bit
    # default route at this size (15.8 edges per latent, d = 16): one decoder Jacobian per latent, edge ends from its columns
    _lib.check(_lib.load().geo_set_option(b"jvp_per_node", 1), "geo_set_option")
    _lib.check(_lib.load().geo_set_option(b"jvp_node_jacobian", 1), "geo_set_option")
    jac = edge_lengths_graph_device(ex, c2["z"], src, dst, 512).cpu().numpy()
    assert not np.array_equal(jac, out[1])
    rel_routes = np.abs(jac - out[1]) / out[1]
    assert np.quantile(rel_routes, 0.99) < 2e-6 and rel_routes.max() < 1e-4, (np.quantile(rel_routes, 0.99), rel_routes.max())
    # fp64 closed form (layer-by-layer torch fp64 on the GPU: 170 s for all edges) on every 8th block of 512 edges -- eval-mode
    # lengths do not depend on their batch, a spread sample of 118 000 edges carries the gate
    s_h, d_h = src.cpu().numpy(), dst.cpu().numpy()
    pick = np.concatenate([np.arange(c * 512, min((c + 1) * 512, len(s_h))) for c in range(0, (len(s_h) + 511) // 512, 8)])
    ref64 = om.edge_lengths(c2["sd"], "batch", c2["size"], c2["z_h"][s_h[pick]], c2["z_h"][d_h[pick]], batch_size=512,
                            training=False, dtype=torch.float64, device="cuda").numpy()
    rel = np.abs(out[1][pick] - ref64) / ref64
    rel_jac = np.abs(jac[pick] - ref64) / ref64
    _record("c2_eval_mode_per_node_vs_fp64", {"edges_bit_identical": int(len(s_h)), "edges_vs_fp64": int(len(rel)),
                                              "frac_within": float(np.mean(rel <= TOL)),
                                              "p99": float(np.quantile(rel, 0.99)), "max_rel": float(rel.max()),
                                              "per_latent_jacobian": {"frac_within": float(np.mean(rel_jac <= TOL)),
                                                                      "p99": float(np.quantile(rel_jac, 0.99)),
                                                                      "max_rel": float(rel_jac.max()),
                                                                      "p99_vs_per_edge_end": float(np.quantile(rel_routes, 0.99)),
                                                                      "max_vs_per_edge_end": float(rel_routes.max())}})
    assert np.mean(rel <= TOL) >= GATE, (int((rel > TOL).sum()), rel.max())
    assert np.quantile(rel, 0.99) < 2e-6
    assert np.mean(rel_jac <= TOL) >= GATE, (int((rel_jac > TOL).sum()), rel_jac.max())
    assert np.quantile(rel_jac, 0.99) < 2e-6


def test_c2_full_chain_vs_oracle_and_dense_matrix(c2):
    from oracle import kmedoids as ok
    from oracle import sssp as osp
    from vqvae_amd.geo.geo_shortest_paths import sssp_multi_device
    res = c2["res"]
    W = _full_chain_vs_oracle(c2, KMED)
    med, assign = res["medoids"], res["assign_flat"]
    assert len(set(med.tolist())) == KMED and (assign[med] == np.arange(KMED)).all()
    Dg, _, dmin, arg, _ = sssp_multi_device(res["W_lcc"], torch.from_numpy(med.astype(np.int32)).to(c2["dev"]),
                                            want_D=True, want_min=True)
    Dg = Dg.cpu().numpy()
    sample = list(range(0, KMED, 16)) + [1, 255, 511]            # 35 of the 512 rows against heap Dijkstra
    np.testing.assert_array_equal(Dg[sample], osp.dijkstra_multi_source(W, med[sample]))
    # the fused chain's assignment/QE equal the dense matrix's argmin/min (reference stages 2 and 3)
    np.testing.assert_array_equal(Dg.argmin(axis=0), assign)
    np.testing.assert_array_equal(arg.cpu().numpy(), assign)
    np.testing.assert_array_equal(dmin.cpu().numpy(), Dg.min(axis=0))
    assert res["qe"] == ok.quantization_error_from(Dg.min(axis=0))


def test_c2_kmedoids_vs_reference_on_formula_weights(c2, golden):
    """Same W as the reference (its own kNN structure, bit-reproducible weights) -> the reference's medoids, all 60 000
    assignments and QE, bit for bit (tests/golden/c2_formula.npz, generated by oracle/gen_golden_c2.py)."""
    from oracle import synthetic as syn
    from vqvae_amd._device import DeviceCSR
    from vqvae_amd.geo.kmeans_optimized import fit_kmedoids_optimized
    g = golden("c2_formula")
    G = c2["res"]["W_lcc"]
    rows = np.repeat(np.arange(N), np.diff(G.indptr.cpu().numpy()))
    w = syn.formula_weights(rows, G.indices.cpu().numpy())
    Gf = DeviceCSR(G.n, G.indptr, G.indices, torch.from_numpy(w).to(c2["dev"]))
    med, assign, qe = fit_kmedoids_optimized(Gf, K=KMED, init="kpp", seed=42)
    np.testing.assert_array_equal(med, g["medoids"])
    np.testing.assert_array_equal(assign, g["assign"].astype(int))
    assert qe == float(g["qe"])


def _graph_with_reference_lengths(c2, golden):
    """The reference CLI's OWN graph at C2: its structure (= ours, sha256-checked) carrying ITS float32 edge lengths
    (tests/golden/c2_cli_lengths.npz: all 946 059, row-major upper triangle = the JVP chunk order)."""
    from oracle import synthetic as syn
    from vqvae_amd.geo.knn_graph_optimized import reweight_device, upper_edges_device
    g, gl = golden("c2_cli"), golden("c2_cli_lengths")
    G = c2["res"]["W_lcc"]
    assert G.n == N                                              # one component: LCC-local = global indices
    W = G.to_scipy()
    np.testing.assert_array_equal(syn.digest(W.indptr.astype(np.int32)), g["indptr_sha256"])
    np.testing.assert_array_equal(syn.digest(W.indices.astype(np.int32)), g["indices_sha256"])
    L = gl["lengths"]
    assert L.dtype == np.float32 and len(L) == int(g["meta"][6])
    np.testing.assert_array_equal(syn.digest(L), gl["lengths_sha256"])
    _, _, entry_edge = upper_edges_device(G)
    return reweight_device(G, entry_edge, torch.from_numpy(L).to(c2["dev"])), L


def test_c2_kmedoids_on_reference_jvp_weights_gives_reference_codes(c2, golden):
    """The stage-wise check at the headline size on the reference's own numbers: reference structure + reference JVP
    edge lengths -> fit_kmedoids_optimized(K=512, seed=42) on the MI355X must return the reference CLI's
    medoid_indices and EVERY entry of its codes.npy (src/scripts/build_codebook.py:70-103)."""
    from vqvae_amd.geo.kmeans_optimized import fit_kmedoids_optimized
    g = golden("c2_cli")
    Gref, _ = _graph_with_reference_lengths(c2, golden)
    med, assign, qe = fit_kmedoids_optimized(Gref, K=KMED, init="kpp", seed=42)
    np.testing.assert_array_equal(med, g["medoid_indices"])
    np.testing.assert_array_equal(assign, g["codes"].astype(np.int64).reshape(-1))
    assert np.isfinite(qe) and qe > 0


def test_c2_all_reference_edge_lengths_vs_gpu(c2, golden):
    """Every one of the reference CLI's 946 059 float32 edge lengths against the HIP JVP (the round-2 fixture held
    every 61st chunk).  Gate on the edges where the reference itself is within 1e-5 of the fp64 closed form."""
    _, ref64 = _all_chunks_vs_fp64(c2, "c2")
    _, Lref = _graph_with_reference_lengths(c2, golden)
    L = c2["res"]["edge_lengths"].cpu().numpy()
    rel = np.abs(L - Lref) / Lref
    ref_ok = np.abs(Lref - ref64) / ref64 <= TOL
    _record("c2_jvp_vs_reference_all", {"edges": int(len(L)), "frac_within": float(np.mean(rel <= TOL)),
                                        "reference_itself_over_1e-5_of_fp64": int((~ref_ok).sum()),
                                        "gpu_over_1e-5_of_fp64": int((np.abs(L - ref64) / ref64 > TOL).sum()),
                                        "frac_within_where_reference_accurate": float(np.mean(rel[ref_ok] <= TOL)),
                                        "p50": float(np.median(rel)), "p99": float(np.quantile(rel, 0.99)),
                                        "max_rel": float(rel.max()), "bit_equal_frac": float(np.mean(L == Lref))})
    assert np.mean(ref_ok) > 0.998                                  # (the reference's own float32 misses fp64 on 0.14 %)
    assert np.mean(rel[ref_ok] <= TOL) >= GATE


def _draw_state(G, centres, u, dev):
    """numpy's RandomState.choice restated on the d_min of `centres` (kmeans_optimized.py:47-61): returns the picked
    index, the distance of u to the nearest cdf step (the draw's flip margin) and the normalised weights."""
    from vqvae_amd.geo.geo_shortest_paths import sssp_multi_device
    _, _, dmin, _, _ = sssp_multi_device(G, torch.from_numpy(np.asarray(centres, np.int32)).to(dev), want_D=False,
                                         want_min=True)
    d = dmin.cpu().numpy()
    p = d ** 2
    p[list(centres)] = 0.0
    p /= p.sum()
    cdf = p.astype(np.float64).cumsum()
    cdf /= cdf[-1]
    i = int(cdf.searchsorted(u, side="right"))
    below = cdf[i - 1] if i > 0 else 0.0
    return i, float(min(u - below, cdf[i] - u)), p, cdf, d


def test_c2_draw_flip_margins_measured(c2, golden):
    """WHY the end-to-end chains part ways (round-2 review, item 1b), from data: for every k-means++ draw up to the first
    one that differs between the GPU's and the reference CLI's own edge lengths, the distance of the uniform deviate to
    the nearest cdf step under both weight sets, and how far the two cdfs are apart at the drawn index.  A draw flips
    when the cdf displacement (the accumulated effect of ALL edge-length differences on all d_min values before the
    drawn node) exceeds that draw's margin.  Recorded in gpurun_out/parity_c2_flip_margins.json; DESIGN.md section 2
    quotes it.  Asserted: the draws before the flip agree, and the flip is explained by displacement > margin."""
    from scipy.sparse import csgraph
    g = golden("c2_cli")
    Gref, Lref = _graph_with_reference_lengths(c2, golden)
    res, dev = c2["res"], c2["dev"]
    Ggpu, Lgpu = res["W_lcc"], res["edge_lengths"].cpu().numpy()
    med_ref, med_gpu = g["medoid_indices"].astype(np.int64), res["medoids"].astype(np.int64)
    same = med_ref == med_gpu
    if same.all():
        _record("c2_flip_margins", {"chains_identical": True})
        return
    j = int(np.argmin(same))                                     # first centre that differs (centre j = draw j)
    rng = np.random.RandomState(42)
    assert int(rng.randint(0, N)) == med_ref[0] == med_gpu[0]
    us = [float(rng.random_sample()) for _ in range(j)]          # one uniform per draw (legacy RandomState.choice)
    margins_ref, margins_gpu, displacement = [], [], []
    for t in range(1, j + 1):
        centres = med_ref[:t].tolist()
        i_ref, m_ref, p_ref, cdf_ref, d_ref = _draw_state(Gref, centres, us[t - 1], dev)
        i_gpu, m_gpu, p_gpu, cdf_gpu, d_gpu = _draw_state(Ggpu, centres, us[t - 1], dev)
        assert i_ref == med_ref[t]                               # the restated draw IS the reference's draw
        assert i_gpu == med_gpu[t]
        margins_ref.append(m_ref)
        margins_gpu.append(m_gpu)
        lo = max(i_ref - 1, 0)
        displacement.append(float(max(abs(cdf_gpu[lo] - cdf_ref[lo]), abs(cdf_gpu[i_ref] - cdf_ref[i_ref]))))
        if t < j:
            assert i_ref == i_gpu
    # the flipping draw: which nodes move the cdf, and which edges move those nodes
    contrib = p_gpu - p_ref
    upto = contrib[:med_ref[j]]
    order = np.argsort(-np.abs(upto))[:8]
    src, dst = (x.cpu().numpy() for x in res["edges"])
    edge_rel = np.abs(Lgpu - Lref) / Lref
    Wref = Gref.to_scipy()
    nodes = []
    for v in order.tolist():
        dist, pred = csgraph.dijkstra(Wref, directed=False, indices=[int(v)], return_predecessors=True)
        centre = min(med_ref[:j].tolist(), key=lambda c: dist[0, c])
        path, node = [], centre
        while node != v and node >= 0:                           # walk centre -> v along the tree rooted at v
            nxt = int(pred[0, node])
            a, b = (node, nxt) if node < nxt else (nxt, node)
            e = int(np.flatnonzero((src == a) & (dst == b))[0])
            path.append({"edge": e, "chunk": e // 512, "rel_diff_gpu_vs_ref": float(edge_rel[e])})
            node = nxt
        nodes.append({"node": int(v), "centre": int(centre), "d_min_ref": float(d_ref[v]),
                      "d_min_rel_diff": float(abs(d_gpu[v] - d_ref[v]) / d_ref[v]), "cdf_contribution": float(contrib[v]),
                      "path": path})
    ulp_free = float(np.mean(Lgpu == Lref))
    _record("c2_flip_margins", {
        "chains_identical": False, "first_differing_centre": j, "draws_checked": j,
        "margin_ref_min_before_flip": float(min(margins_ref[:-1])) if j > 1 else None,
        "margin_ref_median": float(np.median(margins_ref)), "margin_ref_at_flip": margins_ref[-1],
        "margin_gpu_at_flip": margins_gpu[-1], "cdf_displacement_at_flip": displacement[-1],
        "cdf_displacement_median_before_flip": float(np.median(displacement[:-1])) if j > 1 else None,
        "cdf_displacement_max_before_flip": float(max(displacement[:-1])) if j > 1 else None,
        "sum_abs_contribution_before_drawn_node": float(np.abs(upto).sum()),
        "edge_lengths_bit_equal_frac": ulp_free, "edge_rel_diff_p50": float(np.median(edge_rel)),
        "edge_rel_diff_p99": float(np.quantile(edge_rel, 0.99)), "edge_rel_diff_max": float(edge_rel.max()),
        "edges_over_1e-5": int((edge_rel > TOL).sum()),
        "top_nodes_moving_the_cdf": nodes, "margins_ref": margins_ref, "displacements": displacement})
    assert displacement[-1] >= margins_ref[-1] * 0.999           # the flip is a displacement beyond the draw's margin


def test_c2_end_to_end_vs_reference_cli_measured(c2, golden):
    """North-star check at the headline size: codes / medoids of the GPU pipeline against the reference CLI's.
    The graph structure must be identical.  The 511 D^2 draws see edge weights that agree with the reference's only to
    float32 rounding of a different (equally valid) operation order, so identity of the drawn centres is MEASURED and
    recorded here, not assumed (DESIGN.md section 2 states the result); the first centre (randint) must agree."""
    from oracle import synthetic as syn
    g = golden("c2_cli")
    res = c2["res"]
    W = res["W_lcc"].to_scipy()
    np.testing.assert_array_equal(syn.digest(W.indptr.astype(np.int32)), g["indptr_sha256"])
    np.testing.assert_array_equal(syn.digest(W.indices.astype(np.int32)), g["indices_sha256"])
    med_ref, codes_ref = g["medoid_indices"], g["codes"].astype(np.int32).reshape(-1)
    same = res["medoids"] == med_ref
    lead = int(np.argmin(same)) if not same.all() else len(same)
    _record("c2_end_to_end_vs_reference", {"medoids_identical": bool(same.all()), "leading_identical_draws": lead,
                                           "medoids_in_common": int(len(set(res["medoids"].tolist()) & set(med_ref.tolist()))),
                                           "codes_identical_frac": float(np.mean(res["assign_flat"] == codes_ref))})
    assert lead >= 1


# ------------------------------------------------------------------------------------------------- C3
@pytest.fixture(scope="module")
def c3():
    """BASELINE config 3 as stated: 50 000 latents, d=64, 32-px (CIFAR-shaped, 3-channel) decoder, K=512."""
    return _pipeline(50000, 64, 3, 32, KMED)


def test_c3_d64_knn_jvp_chain_vs_oracle(c3):
    n = 50000
    _knn_rows_vs_oracle(c3, n, 64, ((0, 48), (25000, 25048), (n - 48, n)))
    _graph_invariants(c3, n)
    _all_chunks_vs_fp64(c3, "c3")
    _full_chain_vs_oracle(c3, KMED)


def _nearest_equals_k_source(G, K, seeds, dev):
    """Random medoid sets on a resident graph: the one-solve assignment (float32 tie rule of kmeans_optimized.py:100) against
    the K-source solve's column argmin / minimum.  Returns what each attempt reported."""
    from vqvae_amd.geo.geo_shortest_paths import nearest_source_device, sssp_multi_device
    log = []
    for seed in seeds:
        src = torch.from_numpy(np.random.RandomState(seed).choice(G.n, K, replace=False).astype(np.int32)).to(dev)
        info = {}
        d1, a1, _ = nearest_source_device(G, src, info=info)
        _, _, dk, ak, _ = sssp_multi_device(G, src, want_D=False, want_min=True)
        assert torch.equal(a1, ak) and torch.equal(d1, dk), (seed, info, int((a1 != ak).sum()))
        log.append(info)
    return log


def test_c2_one_solve_assignment_equals_k_source_argmin_with_float32_ties(c2):
    """Round-3 review, next-round item 1(b) at C2: on the JVP-weighted 60 000-node graph, 512 and 4 096 random medoids per
    seed; `nearest == K-source argmin` for every node, number of suspect nodes (two medoids inside one float32 rounding
    window) recorded."""
    G = c2["res"]["W_lcc"]
    log512 = _nearest_equals_k_source(G, KMED, range(6), c2["dev"])
    log4k = _nearest_equals_k_source(G, 4096, range(2), c2["dev"])
    _record("c2_one_solve_assignment_float32_ties", {"K512": log512, "K4096": log4k})
    assert not any(i["declined"] for i in log512 + log4k)


def test_real_shape_960k_one_solve_assignment_equals_k_source_argmin():
    """The same at the pipeline's real node count (SURVEY finding 7: 960 000 nodes of d=16, distance-weighted k=20 union
    graph): about one float32 collision per call is expected here, so the suspect resolution runs on real data."""
    from vqvae_amd._device import device, release_workspace
    from vqvae_amd.geo.knn_graph_optimized import knn_graph_device
    from oracle import synthetic as syn
    dev = device()
    z = torch.from_numpy(syn.gauss_latents(960_000, D, 3)).to(dev)
    out = knn_graph_device(z, KNN, mode="distance", sym="union")
    G = out[0] if isinstance(out, tuple) else out
    log = _nearest_equals_k_source(G, KMED, range(4), dev)
    _record("real_one_solve_assignment_float32_ties", {"n": int(G.n), "nnz": int(G.nnz), "K512": log})
    assert not any(i["declined"] for i in log)
    del G, z
    release_workspace()
    torch.cuda.empty_cache()


# ------------------------------------------------------------------------------------------------- C4 on one GPU
def test_c4_one_gpu_1m_latents_k1024():
    """BASELINE config 4's workload on ONE GPU (the 8-GPU sharding is covered by the multi-rank tests): oracle rows
    and draws on samples, invariants on the rest (the oracle chain at this size would take ~1 h)."""
    from oracle import kmedoids as ok
    from oracle import metric as om
    from oracle import sssp as osp
    from vqvae_amd._device import release_workspace
    from vqvae_amd.geo.geo_shortest_paths import sssp_multi_device
    n, K = 1_000_000, 1024
    ctx = _pipeline(n, D, 1, 28, K)
    res = ctx["res"]
    _knn_rows_vs_oracle(ctx, n, D, ((0, 32), (n - 32, n)))
    W = res["W_lcc"].to_scipy()
    assert W.shape == (n, n) and _columns_strictly_ascending(W) and np.diff(W.indptr).min() >= KNN
    assert res["n_edges"] * 2 == W.nnz
    src, dst = (t.cpu().numpy() for t in res["edges"])
    L = res["edge_lengths"].cpu().numpy()
    assert np.isfinite(L).all() and (L > 0).all()
    n_chunks = (len(src) + 511) // 512
    # 256 chunks spread over the run (+ chunk 0, the one round 2 exempted by index) against the fp64 closed form.  Which of
    # them may miss the gate is decided by the CRITERION near0 < NEAR0 (a ReLU-boundary sample in the chunk), not by index:
    # those are recorded (and still held to a median error below 1e-6), every other sampled chunk is gated.
    sample = np.unique(np.concatenate([[0], np.linspace(0, n_chunks - 2, 256).astype(np.int64)]))
    pick = np.concatenate([np.arange(c * 512, (c + 1) * 512) for c in sample])
    ref, cond = om.edge_lengths_dense(ctx["sd"], 28, ctx["z_h"][src[pick]], ctx["z_h"][dst[pick]], batch_size=512,
                                      dtype=torch.float64, device="cuda", with_conditioning=True)
    c_chk = int(sample[len(sample) // 2])                           # the checker itself on one chunk: dense GPU == conv CPU
    cpu64 = om.edge_lengths(ctx["sd"], "batch", 28, ctx["z_h"][src[c_chk * 512:(c_chk + 1) * 512]],
                            ctx["z_h"][dst[c_chk * 512:(c_chk + 1) * 512]], batch_size=512, training=True, dtype=torch.float64).numpy()
    np.testing.assert_allclose(ref.numpy()[(len(sample) // 2) * 512:(len(sample) // 2 + 1) * 512], cpu64, rtol=2e-6)
    rel = (np.abs(L[pick] - ref.numpy()) / ref.numpy()).reshape(len(sample), 512)
    R, near0 = cond.numpy()[:, 0], cond.numpy()[:, 1]
    flagged = near0 < NEAR0
    frac_over = (rel > TOL).mean(axis=1)
    _record("c4_jvp_sampled_chunks_vs_fp64", {
        "sampled_chunks": int(len(sample)), "near0_threshold": NEAR0, "flagged_chunks": int(flagged.sum()),
        "flagged_with_an_edge_over": [{"chunk": int(c), "near0": float(m), "frac_over": float(f)}
                                      for c, m, f in zip(sample[flagged], near0[flagged], frac_over[flagged]) if f > 0],
        "flagged_frac_within": float((rel[flagged] <= TOL).mean()), "flagged_median": float(np.median(rel[flagged])),
        "gated_chunks": int((~flagged).sum()), "gated_frac_within": float((rel[~flagged] <= TOL).mean()),
        "gated_worst_chunk_frac_over": float(frac_over[~flagged].max()), "gated_p99": float(np.quantile(rel[~flagged], 0.99)),
        "gated_max_rel": float(rel[~flagged].max()), "R_max_mean_over_std": float(R.max())})
    assert (~flagged).sum() >= 64
    assert (rel[~flagged] <= TOL).mean() >= GATE and np.quantile(rel[~flagged], 0.99) < 2e-6
    assert rel[~flagged].max() <= TOL                                # chunks without a boundary sample: EVERY edge
    assert np.median(rel[flagged]) < 1e-6
    med, assign = res["medoids"], res["assign_flat"]
    assert len(set(med.tolist())) == K and (assign >= 0).all() and (assign < K).all()
    assert (assign[med] == np.arange(K)).all()
    centers = ok.kpp_initialization_graph(W, 4, seed=42)         # the first draws on the oracle's own solves
    assert centers == med[:4].tolist()
    rows = [0, 3, 511, 1023]
    Dg, _, dmin, arg, _ = sssp_multi_device(res["W_lcc"], torch.from_numpy(med[rows].astype(np.int32)).to(ctx["dev"]),
                                            want_D=True, want_min=True)
    np.testing.assert_array_equal(Dg.cpu().numpy(), osp.dijkstra_multi_source(W, med[rows]))
    _, _, dmin, arg, _ = sssp_multi_device(res["W_lcc"], torch.from_numpy(med.astype(np.int32)).to(ctx["dev"]),
                                           want_D=False, want_min=True)
    np.testing.assert_array_equal(arg.cpu().numpy(), assign)     # the reference's assignment stage == fused chain
    assert res["qe"] == ok.quantization_error_from(dmin.cpu().numpy())
    del ctx, res
    release_workspace()
    torch.cuda.empty_cache()

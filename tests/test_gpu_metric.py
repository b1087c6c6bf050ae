"""GPU parity of csrc/jvp.hip through vqvae_amd.geo.riemannian_metric against the reference's edge
lengths (golden) and the fp64 closed-form oracle; plus the reference's own property tests
(reference tests/test_riemannian_metric.py) on the generic-decoder path."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

DEC_CASES = {"fm_batch": (16, 1, 28, "batch", 10), "fm_none": (16, 1, 28, "none", 11),
             "cf_batch": (32, 3, 32, "batch", 13),
             "cf64_batch": (64, 3, 32, "batch", 14)}      # BASELINE config 3 as stated: latent_dim 64, 32-px decoder
E = 2048
# SURVEY 8(a) gate: >= 99.9 % of edges within 1e-5 relative of the reference f32 output; every edge within
# 1e-5 of the fp64 closed form unless a ReLU pre-activation sits on its rounding boundary (rare outliers).
TOL = 1e-5


def _decoder(name, training):
    from oracle import metric as om
    from vqvae_amd.spatial_decoder import SpatialDecoder
    d, cout, size, norm, seed = DEC_CASES[name]
    sd = om.make_decoder_state(seed, d, cout, norm_type=norm)
    dec = SpatialDecoder(cout, (256, 128, 64), d, size, norm)
    dec.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()})
    dec.train(training)
    r = np.random.RandomState(100 + seed)
    zs = r.randn(E, d).astype(np.float32)
    ze = (zs + 0.3 * r.randn(E, d)).astype(np.float32)
    return dec, sd, zs, ze, (d, cout, size, norm)


@pytest.mark.parametrize("name", list(DEC_CASES))
@pytest.mark.parametrize("training", [True, False])
@pytest.mark.parametrize("bs", [512, 100])
def test_edge_lengths_vs_reference_and_fp64(golden, name, training, bs):
    from oracle import metric as om
    from vqvae_amd.geo.riemannian_metric import edge_lengths_riemannian
    dec, sd, zs, ze, (d, cout, size, norm) = _decoder(name, training)
    dec = dec.cuda()
    L = edge_lengths_riemannian(dec, torch.from_numpy(zs), torch.from_numpy(ze), batch_size=bs)
    assert L.is_cuda and L.dtype == torch.float32 and L.shape == (E,)
    L = L.cpu().numpy()
    ref = golden("metric_cf64" if name == "cf64_batch" else "metric")[f"{name}/train{int(training)}/bs{bs}"]
    rel = np.abs(L - ref) / np.abs(ref)
    assert np.mean(rel <= TOL) >= 0.999, (rel.max(), np.quantile(rel, 0.999))
    L64 = om.edge_lengths(sd, norm, size, zs, ze, batch_size=bs, training=training, dtype=torch.float64).numpy()
    rel64 = np.abs(L - L64) / np.abs(L64)
    assert np.mean(rel64 <= TOL) >= 0.999, (rel64.max(), np.quantile(rel64, 0.999))
    assert np.quantile(rel64, 0.99) < 2e-6


def test_decoder_on_cpu_result_returns_to_cpu():
    from vqvae_amd.geo.riemannian_metric import edge_lengths_riemannian
    dec, sd, zs, ze, _ = _decoder("fm_batch", False)
    L = edge_lengths_riemannian(dec, torch.from_numpy(zs[:100]), torch.from_numpy(ze[:100]), batch_size=32)
    assert not L.is_cuda and L.shape == (100,)
    dec2 = dec.cuda()
    L2 = edge_lengths_riemannian(dec2, torch.from_numpy(zs[:100]).cuda(), torch.from_numpy(ze[:100]).cuda(), batch_size=32)
    np.testing.assert_array_equal(L.numpy(), L2.cpu().numpy())


def test_eval_mode_is_batch_size_invariant_and_train_mode_is_not():
    from vqvae_amd.geo.riemannian_metric import edge_lengths_riemannian
    dec, sd, zs, ze, _ = _decoder("fm_batch", False)
    dec = dec.cuda()
    a = edge_lengths_riemannian(dec, torch.from_numpy(zs[:300]), torch.from_numpy(ze[:300]), batch_size=512)
    b = edge_lengths_riemannian(dec, torch.from_numpy(zs[:300]), torch.from_numpy(ze[:300]), batch_size=37)
    assert torch.allclose(a, b, rtol=1e-5, atol=1e-7)
    dec.train(True)
    c = edge_lengths_riemannian(dec, torch.from_numpy(zs[:300]), torch.from_numpy(ze[:300]), batch_size=512)
    e = edge_lengths_riemannian(dec, torch.from_numpy(zs[:300]), torch.from_numpy(ze[:300]), batch_size=37)
    assert not torch.allclose(c, e, rtol=1e-3)            # batch statistics couple the samples (SURVEY finding 4)


def test_graph_edge_entry_point_matches_pairs():
    from vqvae_amd._device import device
    from vqvae_amd.geo.riemannian_metric import edge_lengths_device, edge_lengths_graph_device
    from vqvae_amd.spatial_decoder import DecoderExport
    dec, sd, zs, ze, _ = _decoder("fm_batch", True)
    dev = device()
    z = torch.from_numpy(zs[:400]).to(dev)
    r = np.random.RandomState(3)
    src = torch.from_numpy(r.randint(0, 400, 1000).astype(np.int32)).to(dev)
    dst = torch.from_numpy(r.randint(0, 400, 1000).astype(np.int32)).to(dev)
    ex = DecoderExport(dec.to(dev), dev)
    a = edge_lengths_graph_device(ex, z, src, dst, 512)
    b = edge_lengths_device(ex, z[src.long()].contiguous(), z[dst.long()].contiguous(), 512)
    np.testing.assert_array_equal(a.cpu().numpy(), b.cpu().numpy())


class DummyDec(torch.nn.Module):
    """Linear test decoder of the reference's tests/test_riemannian_metric.py:6-14."""

    def __init__(self, d=16, hw=28 * 28):
        super().__init__()
        self.lin = torch.nn.Linear(d, hw)

    def forward(self, z):
        return self.lin(z).view(z.size(0), 1, 28, 28)


def _pairs(M=64, D=16, eps=0.1, seed=42, device="cpu"):
    g = torch.Generator(device=device).manual_seed(seed)
    zi = torch.randn(M, D, generator=g, device=device)
    return zi, zi + eps * torch.randn(M, D, generator=g, device=device)


def test_generic_decoder_properties_on_gpu():
    from vqvae_amd.geo.riemannian_metric import edge_lengths_riemannian
    dec = DummyDec().eval().cuda()
    zi, zj = _pairs(device="cuda")
    L = edge_lengths_riemannian(dec, zi, zj, batch_size=32)
    assert L.is_cuda and L.dtype == torch.float32 and L.shape == (64,) and torch.all(L >= 0)
    assert torch.allclose(L, edge_lengths_riemannian(dec, zj, zi, batch_size=64), rtol=1e-4, atol=1e-6)
    zi, zj = _pairs(M=127, device="cuda")
    assert torch.allclose(edge_lengths_riemannian(dec, zi, zj, batch_size=16),
                          edge_lengths_riemannian(dec, zi, zj, batch_size=1024), rtol=1e-5, atol=1e-7)


def _group_decoder(channels, seed=12):
    from oracle import metric as om
    from vqvae_amd.spatial_decoder import SpatialDecoder
    sd = om.make_decoder_state(seed, 16, 1, channels=channels, norm_type="group")
    dec = SpatialDecoder(1, channels, 16, 28, "group")
    dec.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()})
    r = np.random.RandomState(112)
    zs = r.randn(E, 16).astype(np.float32)
    ze = (zs + 0.3 * r.randn(E, 16)).astype(np.float32)
    return dec, sd, zs, ze


@pytest.mark.parametrize("bs", [512, 100])
def test_groupnorm_decoder_on_the_hip_path(golden, bs):
    """GroupNorm (32 groups per layer, the reference's default 256-128-64 decoder) runs in csrc/jvp.hip: per-sample
    group statistics are independent of the chunking and of train/eval mode."""
    from oracle import metric as om
    from vqvae_amd.geo.riemannian_metric import edge_lengths_riemannian
    from vqvae_amd.spatial_decoder import hip_kernels_cover
    dec, sd, zs, ze = _group_decoder((256, 128, 64))
    assert hip_kernels_cover(dec)
    L = edge_lengths_riemannian(dec.cuda().eval(), torch.from_numpy(zs), torch.from_numpy(ze), batch_size=bs)
    assert L.is_cuda and L.shape == (E,)
    L = L.cpu().numpy()
    ref = golden("metric")["fm_group/train0/bs512"]
    rel = np.abs(L[:len(ref)] - ref) / ref
    assert np.mean(rel <= TOL) >= 0.999, (rel.max(), np.quantile(rel, 0.999))
    L64 = om.edge_lengths(sd, "group", 28, zs, ze, batch_size=512, training=False, dtype=torch.float64).numpy()
    rel64 = np.abs(L - L64) / np.abs(L64)
    assert np.mean(rel64 <= TOL) >= 0.999, (rel64.max(), np.quantile(rel64, 0.999))
    assert np.quantile(rel64, 0.99) < 2e-6
    Lt = edge_lengths_riemannian(dec.train(), torch.from_numpy(zs), torch.from_numpy(ze), batch_size=bs).cpu().numpy()
    np.testing.assert_array_equal(L, Lt)


def test_groupnorm_decoder_outside_the_kernels_takes_the_autograd_path():
    """A GroupNorm decoder whose layout the kernels do not cover (dec_channels[2] != 64): the drop-in still answers
    (autograd on the GPU, the reference's own method) and agrees with the fp64 closed form."""
    from oracle import metric as om
    from vqvae_amd.geo.riemannian_metric import edge_lengths_riemannian
    from vqvae_amd.spatial_decoder import hip_kernels_cover
    dec, sd, zs, ze = _group_decoder((64, 32, 16))
    assert not hip_kernels_cover(dec)
    L = edge_lengths_riemannian(dec.cuda().eval(), torch.from_numpy(zs[:512]), torch.from_numpy(ze[:512]), batch_size=512)
    L64 = om.edge_lengths(sd, "group", 28, zs[:512], ze[:512], batch_size=512, training=False, dtype=torch.float64).numpy()
    rel = np.abs(L.cpu().numpy() - L64) / L64
    assert L.is_cuda and np.mean(rel <= 1e-4) >= 0.99


@pytest.mark.parametrize("bs", [512, 100])
def test_train_mode_jvp_updates_batchnorm_running_statistics(golden, bs):
    """torch folds every train-mode batch into running_mean / running_var (momentum 0.1, unbiased variance) and counts
    it; the reference's edge_lengths_riemannian therefore mutates the decoder (riemannian_metric.py:57-58).  The HIP
    path does the same, in call order (start side, end side of each chunk): values from the reference run."""
    from vqvae_amd.geo.riemannian_metric import edge_lengths_riemannian
    g = golden("bn_running_stats")
    for where in ("cuda", "cpu"):                       # resident decoder (updated in place) and a CPU one (copied back)
        dec, sd, zs, ze, _ = _decoder("fm_batch", True)
        dec = dec.to(where)
        edge_lengths_riemannian(dec, torch.from_numpy(zs), torch.from_numpy(ze), batch_size=bs)
        st = dec.state_dict()
        for idx in (1, 4):
            for key in ("running_mean", "running_var"):
                ref = g[f"bs{bs}/deconv_layers.{idx}.{key}"]
                np.testing.assert_allclose(st[f"deconv_layers.{idx}.{key}"].cpu().numpy(), ref, rtol=2e-5, atol=1e-6)
            assert int(st[f"deconv_layers.{idx}.num_batches_tracked"]) == int(g[f"bs{bs}/deconv_layers.{idx}.num_batches_tracked"])
    dec, sd, zs, ze, _ = _decoder("fm_batch", False)    # eval mode: untouched
    dec = dec.cuda()
    before = {k: v.clone() for k, v in dec.state_dict().items()}
    edge_lengths_riemannian(dec, torch.from_numpy(zs), torch.from_numpy(ze), batch_size=bs)
    assert all(torch.equal(v, dec.state_dict()[k]) for k, v in before.items())


@pytest.mark.parametrize("d,norm", [(64, "none"), (24, "none"), (40, "batch"), (64, "group")])
def test_wide_latents_matrix_core_front_equals_the_vector_kernel(d, norm, request):
    """d > 16: the first layer runs on the float32 matrix cores, whose accumulation is the same k-ordered fmaf chain as
    the vector kernel's -- identical lengths without a norm layer (also for a padded latent width), within rounding of
    the fp64 batch statistics' summation order with one; both within the usual gate of the fp64 closed form."""
    from oracle import metric as om
    from vqvae_amd import _lib
    from vqvae_amd._device import device
    from vqvae_amd.geo.riemannian_metric import edge_lengths_riemannian
    from vqvae_amd.spatial_decoder import SpatialDecoder
    sd = om.make_decoder_state(11, d, 3, norm_type=norm)
    dec = SpatialDecoder(3, (256, 128, 64), d, 32, norm)
    dec.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()})
    dec = dec.to(device()).train()
    r = np.random.RandomState(d)
    zs = r.randn(1500, d).astype(np.float32)
    ze = (zs + 0.3 * r.randn(1500, d)).astype(np.float32)
    request.addfinalizer(lambda: _lib.load().geo_set_option(b"jvp_front_valu", 0))
    out = {}
    for valu in (0, 1):
        _lib.check(_lib.load().geo_set_option(b"jvp_front_valu", valu), "geo_set_option")
        out[valu] = edge_lengths_riemannian(dec, torch.from_numpy(zs), torch.from_numpy(ze), 512).cpu().numpy()
    if norm == "none":
        np.testing.assert_array_equal(out[0], out[1])
    else:
        np.testing.assert_allclose(out[0], out[1], rtol=2e-6)
    ref64 = om.edge_lengths(sd, norm, 32, zs, ze, 512, True, dtype=torch.float64).numpy()
    rel = np.abs(out[0] - ref64) / np.abs(ref64)
    assert (rel <= TOL).mean() >= 0.999, (rel <= TOL).mean()


@pytest.mark.parametrize("norm,training,d,cout,size,n_nodes,n_edges,bs", [
    ("batch", False, 16, 1, 28, 3000, 20000, 512),      # BatchNorm in eval mode, more edges than one pass of chunks has tiles
    ("none", True, 16, 1, 28, 777, 5001, 100),          # no norm layer, ragged sizes
    ("batch", False, 64, 3, 32, 1500, 6000, 512),       # wide latents (matrix-core first layer), 192-output head
    ("group", True, 16, 1, 28, 2000, 9000, 512),        # GroupNorm: primal statistics per latent, the tangent's per slot
    ("group", False, 32, 3, 32, 900, 3000, 200),
    ("batch", False, 16, 1, 28, 5000, 100, 512),        # more latents than slots: the call falls back to the per-edge-end path
])
def test_per_node_primal_is_bit_identical_to_the_per_edge_end_path(norm, training, d, cout, size, n_nodes, n_edges, bs, request):
    """Decoders with fixed statistics: the primal pass runs once per LATENT (`jvp_per_node`, default on; SURVEY 2.1 K4'), the edge
    slots carry the tangent alone and take ReLU masks / sigmoid' from their node's rows (GroupNorm: the primal's group statistics
    per latent, the tangent's own per slot).  Same products in the same order: the
    lengths of the edge-list entry point must equal the per-edge-end path's (option off, and the pairs entry point) bit for bit,
    and pass the usual gate of the fp64 closed form."""
    from oracle import metric as om
    from vqvae_amd import _lib
    from vqvae_amd._device import device
    from vqvae_amd.geo.riemannian_metric import edge_lengths_device, edge_lengths_graph_device
    from vqvae_amd.spatial_decoder import DecoderExport, SpatialDecoder
    dev = device()
    sd = om.make_decoder_state(9, d, cout, norm_type=norm)
    dec = SpatialDecoder(cout, (256, 128, 64), d, size, norm)
    dec.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()})
    dec = dec.to(dev)
    dec = dec.train() if training else dec.eval()
    r = np.random.RandomState(n_edges)
    z_h = r.randn(n_nodes, d).astype(np.float32)
    src_h = r.randint(0, n_nodes, n_edges).astype(np.int32)
    dst_h = (src_h + 1 + r.randint(0, n_nodes - 1, n_edges)).astype(np.int32) % n_nodes
    z, src, dst = (torch.from_numpy(a).to(dev) for a in (z_h, src_h, dst_h))
    ex = DecoderExport(dec, dev)
    def restore():
        _lib.load().geo_set_option(b"jvp_per_node", 1)
        _lib.load().geo_set_option(b"jvp_node_jacobian", 1)
    request.addfinalizer(restore)
    _lib.check(_lib.load().geo_set_option(b"jvp_node_jacobian", 0), "geo_set_option")   # (its own test below: not bit-identical)
    out = {}
    for mode in (1, 0):
        _lib.check(_lib.load().geo_set_option(b"jvp_per_node", mode), "geo_set_option")
        out[mode] = edge_lengths_graph_device(ex, z, src, dst, bs).cpu().numpy()
    pairs = edge_lengths_device(ex, z[src.long()].contiguous(), z[dst.long()].contiguous(), bs).cpu().numpy()
    np.testing.assert_array_equal(out[1], out[0])
    np.testing.assert_array_equal(out[1], pairs)
    ref64 = om.edge_lengths(sd, norm, size, z_h[src_h], z_h[dst_h], bs, training, dtype=torch.float64).numpy()
    rel = np.abs(out[1] - ref64) / np.abs(ref64)
    assert (rel <= TOL).mean() >= 0.999, (rel <= TOL).mean()


@pytest.mark.gpu
@pytest.mark.parametrize("norm,training,d,cout,size,n_nodes,n_edges,bs,forced", [
    ("batch", False, 16, 1, 28, 3000, 40000, 512, False),     # 13 edges per latent: the route is taken on its own
    ("batch", False, 16, 1, 28, 3001, 20000, 512, True),      # odd latent count: the last latent is its own partner
    ("none", True, 16, 1, 28, 777, 5001, 100, True),          # no norm layer, ragged sizes
    ("none", False, 5, 1, 28, 1000, 9000, 64, False),         # narrow latent, small chunks
    ("batch", False, 16, 3, 32, 1500, 6000, 512, True),       # 192-output head
    ("group", True, 16, 1, 28, 2000, 9000, 512, True),        # GroupNorm: statistics per sample, the Jacobian still per latent
])
def test_per_latent_jacobian_route_matches_the_per_edge_end_path(norm, training, d, cout, size, n_nodes, n_edges, bs, forced,
                                                                 request):
    """Decoders with fixed statistics and d <= 16 (`jvp_node_jacobian`; SURVEY 2.1 K4'): the d unit tangents of every latent go
    through the tangent-only pass once, and each edge end is |J(z_node) dz| from the stored columns.  A different summation
    order from the per-edge-end path (J's columns are rounded to float32 before they are combined), so not bit-identical:
    the route must pass the same fp64 gate as every other path, and sit within 2e-6 of the per-edge-end lengths at the 99th
    percentile.  `forced`: option 2 (the automatic rule wants >= 0.75 d edges per latent)."""
    from oracle import metric as om
    from vqvae_amd import _lib
    from vqvae_amd._device import device
    from vqvae_amd.geo.riemannian_metric import edge_lengths_graph_device
    from vqvae_amd.spatial_decoder import DecoderExport, SpatialDecoder
    dev = device()
    sd = om.make_decoder_state(11, d, cout, norm_type=norm)
    dec = SpatialDecoder(cout, (256, 128, 64), d, size, norm)
    dec.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()})
    dec = dec.to(dev)
    dec = dec.train() if training else dec.eval()
    r = np.random.RandomState(n_edges + 1)
    z_h = r.randn(n_nodes, d).astype(np.float32)
    src_h = r.randint(0, n_nodes, n_edges).astype(np.int32)
    dst_h = (src_h + 1 + r.randint(0, n_nodes - 1, n_edges)).astype(np.int32) % n_nodes
    src_h[:3], dst_h[:3] = n_nodes - 1, [0, 1, n_nodes - 2]     # the last latent is on edges, on both sides
    dst_h[3], src_h[3] = n_nodes - 1, 0
    z, src, dst = (torch.from_numpy(a).to(dev) for a in (z_h, src_h, dst_h))
    ex = DecoderExport(dec, dev)
    lib = _lib.load()
    request.addfinalizer(lambda: lib.geo_set_option(b"jvp_node_jacobian", 1))
    _lib.check(lib.geo_set_option(b"jvp_node_jacobian", 0), "geo_set_option")
    per_edge_end = edge_lengths_graph_device(ex, z, src, dst, bs).cpu().numpy()
    _lib.check(lib.geo_set_option(b"jvp_node_jacobian", 2 if forced else 1), "geo_set_option")
    got = edge_lengths_graph_device(ex, z, src, dst, bs).cpu().numpy()
    assert not np.array_equal(got, per_edge_end)                 # (it did run: another summation order)
    ref64 = om.edge_lengths(sd, norm, size, z_h[src_h], z_h[dst_h], bs, training, dtype=torch.float64).numpy()
    rel = np.abs(got - ref64) / np.abs(ref64)
    assert (rel <= TOL).mean() >= 0.999, (rel <= TOL).mean()
    assert np.quantile(np.abs(got - per_edge_end) / per_edge_end, 0.99) < 2e-6
    # the route's error against fp64 is no worse than the per-edge-end path's
    assert np.median(rel) <= 2 * np.median(np.abs(per_edge_end - ref64) / np.abs(ref64)) + 1e-8


@pytest.mark.gpu
def test_per_latent_jacobian_route_over_several_passes(request):
    """70 001 latents x 16 unit tangents = 1.12 M tangent slots: more than one pass of the activation workspace (2^20 slots), the
    columns of a latent pair may straddle two passes.  Against the per-edge-end path on 600 000 edges."""
    from oracle import metric as om
    from vqvae_amd import _lib
    from vqvae_amd._device import device
    from vqvae_amd.geo.riemannian_metric import edge_lengths_graph_device
    from vqvae_amd.spatial_decoder import DecoderExport, SpatialDecoder
    dev = device()
    n_nodes, n_edges, d = 70001, 600000, 16
    sd = om.make_decoder_state(13, d, 1, norm_type="batch")
    dec = SpatialDecoder(1, (256, 128, 64), d, 28, "batch")
    dec.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()})
    dec = dec.to(dev).eval()
    r = np.random.RandomState(5)
    z = torch.from_numpy(r.randn(n_nodes, d).astype(np.float32)).to(dev)
    src_h = r.randint(0, n_nodes, n_edges).astype(np.int32)
    dst_h = (src_h + 1 + r.randint(0, n_nodes - 1, n_edges)).astype(np.int32) % n_nodes
    src, dst = torch.from_numpy(src_h).to(dev), torch.from_numpy(dst_h).to(dev)
    ex = DecoderExport(dec, dev)
    lib = _lib.load()
    request.addfinalizer(lambda: lib.geo_set_option(b"jvp_node_jacobian", 1))
    _lib.check(lib.geo_set_option(b"jvp_node_jacobian", 0), "geo_set_option")
    per_edge_end = edge_lengths_graph_device(ex, z, src, dst, 512).cpu().numpy()
    _lib.check(lib.geo_set_option(b"jvp_node_jacobian", 2), "geo_set_option")
    got = edge_lengths_graph_device(ex, z, src, dst, 512).cpu().numpy()
    assert not np.array_equal(got, per_edge_end)
    rel = np.abs(got - per_edge_end) / per_edge_end
    assert np.quantile(rel, 0.99) < 2e-6 and rel.max() < 1e-4, (np.quantile(rel, 0.99), rel.max())


@pytest.mark.gpu
def test_per_latent_jacobian_route_is_not_taken_where_it_does_not_apply(request):
    """Train-mode BatchNorm (the statistics are the batch's), d > 16, and graphs with few edges per latent keep the per-edge-end
    path: same bits with the option on (1, 2) and off."""
    from oracle import metric as om
    from vqvae_amd import _lib
    from vqvae_amd._device import device
    from vqvae_amd.geo.riemannian_metric import edge_lengths_graph_device
    from vqvae_amd.spatial_decoder import DecoderExport, SpatialDecoder
    dev = device()
    lib = _lib.load()
    request.addfinalizer(lambda: lib.geo_set_option(b"jvp_node_jacobian", 1))
    for norm, training, d, n_nodes, n_edges, modes in [("batch", True, 16, 1200, 30000, (1, 2)), ("batch", False, 32, 600, 30000, (1, 2)),
                                                       ("batch", False, 16, 3000, 9000, (1,))]:
        sd = om.make_decoder_state(3, d, 1, norm_type=norm)
        dec = SpatialDecoder(1, (256, 128, 64), d, 28, norm)
        dec.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()})
        dec = dec.to(dev)
        dec = dec.train() if training else dec.eval()
        r = np.random.RandomState(d + n_edges)
        z = torch.from_numpy(r.randn(n_nodes, d).astype(np.float32)).to(dev)
        src_h = r.randint(0, n_nodes, n_edges).astype(np.int32)
        dst_h = (src_h + 1 + r.randint(0, n_nodes - 1, n_edges)).astype(np.int32) % n_nodes
        src, dst = torch.from_numpy(src_h).to(dev), torch.from_numpy(dst_h).to(dev)
        ex = DecoderExport(dec, dev)
        _lib.check(lib.geo_set_option(b"jvp_node_jacobian", 0), "geo_set_option")
        off = edge_lengths_graph_device(ex, z, src, dst, 512).cpu().numpy()
        for mode in modes:
            _lib.check(lib.geo_set_option(b"jvp_node_jacobian", mode), "geo_set_option")
            np.testing.assert_array_equal(edge_lengths_graph_device(ex, z, src, dst, 512).cpu().numpy(), off)

import sys, numpy as np, torch
sys.path.insert(0, '.')
from oracle import metric as om
from vqvae_amd._device import device
from vqvae_amd.spatial_decoder import SpatialDecoder, DecoderExport
from vqvae_amd.geo.knn_graph_optimized import knn_graph_device, upper_edges_device
from vqvae_amd.geo.riemannian_metric import edge_lengths_graph_device
N, D = 60000, 16
dev = device()
z_h = np.random.RandomState(0).randn(N, D).astype(np.float32)
sd = om.make_decoder_state(0, D, 1, norm_type="batch")
dec = SpatialDecoder(1, (256, 128, 64), D, 28, "batch")
dec.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()})
z = torch.from_numpy(z_h).to(dev)
G, _, _ = knn_graph_device(z, 20, mode="connectivity", sym="union")
src, dst, _ = upper_edges_device(G)
ex = DecoderExport(dec.to(dev).train(), dev)
worst = []
for c in (917, 100, 500, 1500):
    sl = slice(c*512, (c+1)*512)
    L = edge_lengths_graph_device(ex, z, src[sl].contiguous(), dst[sl].contiguous(), 512).cpu().numpy()
    s, d_ = src[sl].cpu().numpy(), dst[sl].cpu().numpy()
    r32 = om.edge_lengths(sd, "batch", 28, z_h[s], z_h[d_], 512, True).numpy()
    r64 = om.edge_lengths(sd, "batch", 28, z_h[s], z_h[d_], 512, True, dtype=torch.float64).numpy()
    for name, a, b in (("gpu vs f64", L, r64), ("o32 vs f64", r32, r64), ("gpu vs o32", L, r32)):
        rel = np.abs(a-b)/np.abs(b)
        print(c, name, "frac<=1e-5", round(float(np.mean(rel<=1e-5)),4), "max %.2e"%rel.max(), "p99 %.2e"%np.quantile(rel,.99), "median %.2e"%np.median(rel))

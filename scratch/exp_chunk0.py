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
L = edge_lengths_graph_device(ex, z, src[:1024], dst[:1024], 512).cpu().numpy()
s, d_ = src[:1024].cpu().numpy(), dst[:1024].cpu().numpy()
for c in (0, 1):
    sl = slice(c*512, (c+1)*512)
    r32 = om.edge_lengths(sd, "batch", 28, z_h[s[sl]], z_h[d_[sl]], 512, True).numpy()
    r64 = om.edge_lengths(sd, "batch", 28, z_h[s[sl]], z_h[d_[sl]], 512, True, dtype=torch.float64).numpy()
    for name, a, b in (("gpu vs f32", L[sl], r32), ("gpu vs f64", L[sl], r64), ("f32 vs f64", r32, r64)):
        rel = np.abs(a-b)/np.abs(b)
        print(c, name, "frac<=1e-5", np.mean(rel<=1e-5), "max", rel.max(), "median", np.median(rel))
    print("distinct sources in chunk", len(set(s[sl].tolist())))

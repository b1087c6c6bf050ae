"""Eval-mode (fixed BatchNorm statistics) edge lengths at the C2 size, the three routes on one box:
per-edge-end (primal + tangent per slot), per-node primal (tangent-only slots), per-latent Jacobian (d unit tangents per latent)."""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vqvae_amd._lib as _lib
from oracle import metric as om
from vqvae_amd._device import device
from vqvae_amd.spatial_decoder import SpatialDecoder, DecoderExport
from vqvae_amd.geo.riemannian_metric import edge_lengths_graph_device
N, D, E = 60000, 16, 946059
dev = device()
rs = np.random.RandomState(0)
z = torch.from_numpy(rs.randn(N, D).astype(np.float32)).to(dev)
src = torch.from_numpy(rs.randint(0, N, E).astype(np.int32)).to(dev)
dst = torch.from_numpy(rs.randint(0, N, E).astype(np.int32)).to(dev)
sd = om.make_decoder_state(0, D, 1, norm_type="batch")
dec = SpatialDecoder(1, (256, 128, 64), D, 28, "batch")
dec.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()})
ex = DecoderExport(dec.to(dev).eval(), dev)
lib = _lib.load()
out = {}
for name, per_node, jac in (("per-edge-end", 0, 0), ("per-node primal", 1, 0), ("per-latent jacobian", 1, 1)):
    lib.geo_set_option(b"jvp_per_node", per_node)
    lib.geo_set_option(b"jvp_node_jacobian", jac)
    for _ in range(2):
        out[name] = edge_lengths_graph_device(ex, z, src, dst, 512)
    torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); edge_lengths_graph_device(ex, z, src, dst, 512); b.record(); torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
    print("%-20s ms: min %.3f median %.3f" % (name, min(ts), sorted(ts)[2]), flush=True)
a, b = out["per-latent jacobian"].cpu().numpy(), out["per-node primal"].cpu().numpy()
ok = b > 0                               # (random endpoints: a few self loops of length 0)
rel = np.abs(a[ok] - b[ok]) / b[ok]
assert (a[~ok] == 0).all()
print("jacobian vs per-node: p50 %.2e p99 %.2e max %.2e" % (np.median(rel), np.quantile(rel, 0.99), rel.max()), flush=True)

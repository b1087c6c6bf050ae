"""JVP stage of the 60 000-latent configuration with the decoder in EVAL mode (fixed BatchNorm statistics): per-node primal
(default) against the per-edge-end path (jvp_per_node = 0)."""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import metric as om
from vqvae_amd import _lib
from vqvae_amd._device import device
from vqvae_amd.spatial_decoder import SpatialDecoder, DecoderExport
from vqvae_amd.geo.riemannian_metric import edge_lengths_graph_device
N, D, E = 60000, 16, 946059
dev = device()
rs = np.random.RandomState(0)
z = torch.from_numpy(rs.randn(N, D).astype(np.float32)).to(dev)
src = torch.from_numpy(rs.randint(0, N, E).astype(np.int32)).to(dev)
dst = torch.from_numpy(rs.randint(0, N, E).astype(np.int32)).to(dev)
for norm, train in (("batch", False), ("none", True), ("batch", True)):
    sd = om.make_decoder_state(0, D, 1, norm_type=norm)
    dec = SpatialDecoder(1, (256, 128, 64), D, 28, norm)
    dec.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()})
    dec = dec.to(dev)
    ex = DecoderExport(dec.train() if train else dec.eval(), dev)
    res = {}
    for mode in (1, 0):
        _lib.load().geo_set_option(b"jvp_per_node", mode)
        for _ in range(2):
            out = edge_lengths_graph_device(ex, z, src, dst, 512)
        torch.cuda.synchronize()
        ts = []
        for _ in range(5):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(); out = edge_lengths_graph_device(ex, z, src, dst, 512); b.record(); torch.cuda.synchronize()
            ts.append(a.elapsed_time(b))
        res[mode] = (sorted(ts)[2], out.cpu().numpy())
    print(f"norm={norm} training={train}: per-node {res[1][0]:.2f} ms, per-edge-end {res[0][0]:.2f} ms, identical {np.array_equal(res[1][1], res[0][1])}", flush=True)
_lib.load().geo_set_option(b"jvp_per_node", 1)

"""Timing of the JVP stage alone (946 059 random edges over 60 000 latents, batch 512, train-mode BatchNorm).
usage: exp_jvp_ablate.py [prod | path/to/another/libgeo_hip.so]  -- the second form times a library variant (A/B on one box)."""
import os, sys, time, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vqvae_amd._lib as _lib
if len(sys.argv) > 1 and sys.argv[1] != "prod":
    _lib.LIB_PATH = sys.argv[1]
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
ex = DecoderExport(dec.to(dev).train(), dev)
for _ in range(2):
    edge_lengths_graph_device(ex, z, src, dst, 512)
torch.cuda.synchronize()
ts = []
for _ in range(5):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); edge_lengths_graph_device(ex, z, src, dst, 512); b.record(); torch.cuda.synchronize()
    ts.append(a.elapsed_time(b))
print(sys.argv[1] if len(sys.argv) > 1 else "prod", "jvp ms: min %.3f median %.3f" % (min(ts), sorted(ts)[2]), flush=True)

import ctypes
lib = _lib.load()
try:
    f = lib.geo_debug_mid_prof
except AttributeError:
    f = None
if f is not None:                       # a library built with -DGEO_MID_PROF: s_memtime stamps per phase of mid_all_kernel
    buf = (ctypes.c_ulonglong * 16)()
    f(buf, 1)
    edge_lengths_graph_device(ex, z, src, dst, 512); torch.cuda.synchronize()
    f(buf, 0)
    names = ["prologue+S0", "interval0", "interval1", "interval2", "interval3 (P3)", "in barriers", "epilogue", "tiles"]
    for g in (0, 1):
        n = max(1, buf[g * 8 + 7])
        print("wave group", g, {names[k]: round(buf[g * 8 + k] / n) for k in range(7)}, "cycles per tile; tiles", buf[g * 8 + 7], flush=True)

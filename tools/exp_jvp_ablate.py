"""Timing of the JVP stage with a library variant (scratch/abl/libgeo_abl<mask>.so, built with -DGEO_MID_ABLATE=<mask>):
which part of mid_all_kernel the time goes to.  Results of ablated variants are wrong by construction."""
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
lib = _lib.load() if hasattr(_lib, "load") else None
try:
    f = lib.geo_debug_mid_prof
except AttributeError:
    f = None
if f is not None:
    buf = (ctypes.c_ulonglong * 16)()
    f(buf, 1)
    edge_lengths_graph_device(ex, z, src, dst, 512); torch.cuda.synchronize()
    f(buf, 0)
    names = ["prologue", "stage", "prefetch+barrier", "products", "epilogue", "-", "-", "waves"]
    for g in (0, 1):
        n = max(1, buf[g * 8 + 7])
        print("wave group", g, {names[k]: round(buf[g * 8 + k] / n) for k in range(5)}, "per tile (s_memtime ticks), waves", buf[g * 8 + 7])

try:
    fs = lib.geo_debug_stat_prof
except AttributeError:
    fs = None
if fs is not None:
    buf = (ctypes.c_ulonglong * (256 * 8 * 8))()
    edge_lengths_graph_device(ex, z, src, dst, 512); torch.cuda.synchronize()
    fs(buf)
    a = np.frombuffer(buf, dtype=np.uint64).reshape(256, 8, 8).astype(np.float64)
    for kind in (0, 1, 2):
        for grp, name in ((slice(0, 2), "waves 0-1"), (slice(2, 4), "waves 2-3")):
            sel = a[:, grp, :][a[:, grp, 6] == kind]
            if len(sel) == 0:
                continue
            n = sel[:, 4].sum()
            print("kind", kind, name, "per phase: - %.0f products+stage %.0f barrier %.0f epilogue %.0f  (phases per wave %.0f; loop %.0f ticks = %.0f us of the 100 MHz clock -> %.2f GHz)" % (
                sel[:, 0].sum() / n, sel[:, 1].sum() / n, sel[:, 2].sum() / n, sel[:, 3].sum() / n, n / len(sel),
                sel[:, 5].mean(), sel[:, 7].mean() / 100.0, sel[:, 5].mean() / (sel[:, 7].mean() * 10.0)))

"""The K-source assignment solve alone on a bench workload's OWN graph (JVP-weighted, the chain's medoids as sources):
what bench.py times as `assign_sweep`.  GEO_SSSP_TRACE=1 prints the sampled improvement counts of every sweep."""
import os, sys, time, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from vqvae_amd import _lib
from vqvae_amd.geo.geo_shortest_paths import sssp_multi_device
from vqvae_amd.scripts.build_codebook import build_codebook_device
name = sys.argv[1] if len(sys.argv) > 1 else "swiss"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
dev = torch.device("cuda", 0)
z, dec, cfg = bench.make_inputs(name, dev)
import contextlib
with contextlib.redirect_stdout(sys.stderr):
    res = build_codebook_device(z, dec, k=cfg["k"], sym="union", K=cfg["K"], init="kpp", seed=42, batch_size=512)
G = res["W_lcc"]
src = torch.from_numpy(res["medoids"].astype(np.int32)).to(dev)
w = G.data
print(f"{name}: n={G.n} nnz={G.nnz} weights min={float(w.min()):.4g} max={float(w.max()):.4g}", flush=True)
lib = _lib.load()
for rep in range(reps):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    _, _, dmin, amin, sweeps = sssp_multi_device(G, src, want_D=False, want_min=True)
    torch.cuda.synchronize(); wall = (time.perf_counter() - t0) * 1e3
    ms, l = np.zeros(1), np.zeros(1, np.int32)
    layout = lib.geo_sssp_last_profile(ms.ctypes.data, l.ctypes.data)
    print(f"rep {rep}: wall {wall:.2f} ms, sweeps {int(l[0])} in {float(ms[0]):.2f} ms, layout {layout}, max dmin {float(dmin.max()):.4g}, "
          f"checksum {float(dmin.double().sum()):.9e} {int(amin.long().sum())}", flush=True)
# the same question answered by ONE label-carrying solve (geo_sssp_nearest_source)
from vqvae_amd.geo.geo_shortest_paths import nearest_source_device
for rep in range(reps):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    d1, a1, sw = nearest_source_device(G, src)
    torch.cuda.synchronize(); wall = (time.perf_counter() - t0) * 1e3
    print(f"nearest-source rep {rep}: wall {wall:.2f} ms, sweeps {sw}, equal to the K-source solve: {bool(torch.equal(d1, dmin))} {bool(torch.equal(a1, amin))}", flush=True)

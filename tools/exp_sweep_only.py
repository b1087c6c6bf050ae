"""Multi-source sweep alone on the bench c2 graph shape (distance-weighted kNN graph, 512 random sources).
Run under rocprofv3 --kernel-trace to get per-launch durations (tools/sweep_trace.py prints them)."""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vqvae_amd import _lib
from vqvae_amd.geo.knn_graph_optimized import knn_graph_device
from vqvae_amd.geo.geo_shortest_paths import sssp_multi_device
dev = torch.device('cuda', 0)
lib = _lib.load()
N = int(sys.argv[1]) if len(sys.argv) > 1 else 60000
kind = sys.argv[2] if len(sys.argv) > 2 else 'gauss'
import bench
z = torch.from_numpy(bench.swiss_roll(N, 16, 0) if kind == 'swiss' else np.random.RandomState(0).randn(N, 16).astype(np.float32)).to(dev)
G, _, _ = knn_graph_device(z, 20, mode='distance', sym='union')
src = torch.from_numpy(np.random.RandomState(1).choice(N, 512, replace=False).astype(np.int32)).to(dev)
for rep in range(2):
    _, _, dmin, amin, sweeps = sssp_multi_device(G, src, want_D=False, want_min=True)
ms, l = np.zeros(1), np.zeros(1, np.int32)
lib.geo_sssp_last_profile(ms.ctypes.data, l.ctypes.data)
print(f"N={N} nnz={G.nnz} sweeps={int(l[0])} ms={float(ms[0]):.3f} checksum={float(dmin.double().sum()):.9e} {int(amin.long().sum())}")

import sys, numpy as np, torch
sys.path.insert(0, '.')
from vqvae_amd import _lib
from vqvae_amd.geo.knn_graph_optimized import knn_graph_device
from vqvae_amd.geo.geo_shortest_paths import sssp_multi_device
dev = torch.device('cuda', 0)
lib = _lib.load()
def prof():
    ms, l = np.zeros(1), np.zeros(1, np.int32)
    lib.geo_sssp_last_profile(ms.ctypes.data, l.ctypes.data)
    return float(ms[0]), int(l[0])
for N in (2048, 4096, 6000, 8192, 16384, 60000):
    z = torch.from_numpy(np.random.RandomState(0).randn(N, 16).astype(np.float32)).to(dev)
    G, _, _ = knn_graph_device(z, 20, mode='distance', sym='union')
    src = torch.from_numpy(np.random.RandomState(1).choice(N, 512, replace=False).astype(np.int32)).to(dev)
    for rep in range(2):
        sssp_multi_device(G, src, want_D=False, want_min=True)
        m, l = prof()
    gather = G.nnz * 512.0 * 8 * l      # bytes gathered over all sweeps
    print(f"N={N} nnz={G.nnz} D_batch={N*512/1e6:.1f}MB sweeps={l} ms={m:.3f} per-sweep us={1e3*m/l:.1f} gather TB/s={gather/m/1e9:.2f}")

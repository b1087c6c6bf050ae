import sys, numpy as np, torch
sys.path.insert(0, '.')
from vqvae_amd import _lib
from vqvae_amd.geo.knn_graph_optimized import knn_graph_device
from vqvae_amd.geo.geo_shortest_paths import sssp_multi_device
dev = torch.device('cuda', 0)
z = torch.from_numpy(np.random.RandomState(0).randn(60000, 16).astype(np.float32)).to(dev)
G, _, _ = knn_graph_device(z, 20, mode='distance', sym='union')
src = torch.from_numpy(np.random.RandomState(1).choice(60000, 512, replace=False).astype(np.int32)).to(dev)
lib = _lib.load()
def prof():
    ms, l = np.zeros(1), np.zeros(1, np.int32)
    lib.geo_sssp_last_profile(ms.ctypes.data, l.ctypes.data)
    return float(ms[0]), int(l[0])
for per in (512, 256, 128, 64):
    for rep in range(2):
        tot, launches = 0.0, 0
        for s0 in range(0, 512, per):
            sssp_multi_device(G, src[s0:s0 + per].contiguous(), want_D=False, want_min=True)
            m, l = prof(); tot += m; launches += l
    B = 512 * (16.0 * G.nnz + 16.0 * G.n)
    print(f"sources per call {per}: sweep ms {tot:.3f} launches {launches} -> {B / tot / 1e6:.0f} GB/s algorithmic")

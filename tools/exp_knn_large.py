"""kNN stage alone at a large size (default 1 000 000 x 16): wall time of knn_graph_device + upper_edges_device, for a kernel trace."""
import os, sys, time, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vqvae_amd._lib as _lib
if len(sys.argv) > 3:
    _lib.LIB_PATH = sys.argv[3]
from vqvae_amd.geo.knn_graph_optimized import knn_graph_device, upper_edges_device
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
d = int(sys.argv[2]) if len(sys.argv) > 2 else 16
dev = torch.device("cuda", 0)
z = torch.from_numpy(np.random.RandomState(0).randn(n, d).astype(np.float32)).to(dev)
for rep in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    G, _, _ = knn_graph_device(z, 20, mode="connectivity", sym="union", need_dist=False)
    src, dst, ee = upper_edges_device(G)
    torch.cuda.synchronize()
    print(f"rep {rep}: {1e3 * (time.perf_counter() - t0):.1f} ms, nnz {G.nnz}", flush=True)

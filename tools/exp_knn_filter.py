"""kNN with and without the matrix-core filter: equality of the lists, time."""
import os, sys, time, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vqvae_amd.geo.knn_graph_optimized import knn_search_device
dev = torch.device('cuda', 0)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 40000
d = int(sys.argv[2]) if len(sys.argv) > 2 else 16
z = torch.from_numpy(np.random.RandomState(0).randn(N, d).astype(np.float32)).to(dev)
def run(flag):
    os.environ["GEO_KNN_FILTER"] = flag
    for _ in range(2):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        idx, d2 = knn_search_device(z, 21)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
    return idx.cpu().numpy(), d2.cpu().numpy(), dt
i0, d0, t0 = run("0")
i1, d1, t1 = run("1")
bad = (i0 != i1).any(axis=1)
print(f"N={N} d={d} exact {t0*1e3:.2f} ms  filter {t1*1e3:.2f} ms  rows differing {bad.sum()}  d2 equal {np.array_equal(d0, d1)}")
if bad.any():
    r = np.nonzero(bad)[0][0]
    print("row", r, "\n exact ", i0[r], "\n filter", i1[r], "\n", d0[r], "\n", d1[r])

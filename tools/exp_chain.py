"""k-means++ chain: log of the geo_kpp_chain calls (segment, budget, sweeps used, aborts) and total time.
usage: exp_chain.py [gauss|swiss] [N] [K] [path/to/another/libgeo_hip.so]"""
import sys, os, time, ctypes, contextlib, io, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from vqvae_amd.geo.knn_graph_optimized import knn_graph_device
import vqvae_amd.geo.kmeans_optimized as km
from vqvae_amd import _lib
if len(sys.argv) > 4:
    _lib.LIB_PATH = sys.argv[4]
import bench
dev = torch.device('cuda', 0)
kind = sys.argv[1] if len(sys.argv) > 1 else 'gauss'
N = int(sys.argv[2]) if len(sys.argv) > 2 else 60000
K = int(sys.argv[3]) if len(sys.argv) > 3 else 512
z = bench.swiss_roll(N, 16, 0) if kind == 'swiss' else np.random.RandomState(0).randn(N, 16).astype(np.float32)
G, _, _ = knn_graph_device(torch.from_numpy(z).to(dev), 20, mode='distance', sym='union')
lib = _lib.load()
orig = lib.geo_kpp_chain
calls = []
def spy(*a):
    t0 = time.perf_counter(); rc = orig(*a); dt = time.perf_counter() - t0
    st = np.ctypeslib.as_array((ctypes.c_int32 * 4).from_address(a[-2]))
    calls.append((a[9], a[10], a[12], int(st[0]), int(st[1]), int(st[3]), round(dt * 1e3, 2)))
    return rc
class L:
    def __getattr__(self, k): return spy if k == 'geo_kpp_chain' else getattr(lib, k)
km._lib.load = lambda: L()
for rep in range(2):
    calls.clear()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    with contextlib.redirect_stdout(io.StringIO()):
        med, assign, qe = km.fit_kmedoids_optimized(G, K=K, init="kpp", seed=42)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
print(kind, "N", N, "K", K, "total ms", round(dt * 1e3, 1), "calls", len(calls), "aborts", sum(1 for c in calls if c[3] >= 0), "qe", qe)
print("(it0, it1, budget, abort_iter, reason, used, ms):")
for c in calls: print("  ", c)

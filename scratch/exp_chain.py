import sys, os, time, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vqvae_amd.geo.knn_graph_optimized import knn_graph_device
import vqvae_amd.geo.kmeans_optimized as km
from vqvae_amd import _lib
dev = torch.device('cuda', 0)
z = torch.from_numpy(np.random.RandomState(0).randn(60000, 16).astype(np.float32)).to(dev)
G, _, _ = knn_graph_device(z, 20, mode='distance', sym='union')
lib = _lib.load()
orig = lib.geo_kpp_chain
calls = []
def spy(*a):
    t0 = time.perf_counter(); rc = orig(*a); dt = time.perf_counter() - t0
    st = np.ctypeslib.as_array((__import__('ctypes').c_int32 * 4).from_address(a[-2]))
    calls.append((a[9], a[10], a[13], a[14], int(st[0]), int(st[1]), int(st[2]), round(dt * 1e3, 2)))
    return rc
class L:  # proxy
    def __getattr__(self, k): return spy if k == 'geo_kpp_chain' else getattr(lib, k)
km._lib.load = lambda: L()
import contextlib, io
for micro, sw in (("0","10"),("0","8"),("0","7"),("0","6")):
    os.environ["GEO_KPP_MICRO"] = micro; os.environ["GEO_KPP_SWEEPS"] = sw
    for rep in range(2):
        calls.clear()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        with contextlib.redirect_stdout(io.StringIO()):
            med, assign, qe = km.fit_kmedoids_optimized(G, K=512, init="kpp", seed=42)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print("sweeps", sw, "micro", micro, "total ms", round(dt * 1e3, 1), "calls", len(calls), "aborts", sum(1 for c in calls if c[4] >= 0), "qe", qe)
    print("   first calls (it0,it1,micro,finite,abort_iter,reason,n_inf,ms):", calls[:6])

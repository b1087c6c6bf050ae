"""All-pairs matrix + Voronoi iterations at the 60 000-latent configuration: times of the pieces (extension, f4)."""
import os, sys, time, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vqvae_amd._device import device
from vqvae_amd.geo.knn_graph_optimized import knn_graph_device
from vqvae_amd.geo.geo_shortest_paths import all_pairs_geodesic_device
from vqvae_amd.geo.kmeans_optimized import assign_from_rows_device, medoid_update_device, fit_kmedoids_optimized
N, d, K = int(sys.argv[1]) if len(sys.argv) > 1 else 60000, 16, 512
dev = device()
z = torch.from_numpy(np.random.RandomState(0).randn(N, d).astype(np.float32)).to(dev)
G, _, _ = knn_graph_device(z, 20, mode="distance", sym="union")
def timed(f):
    torch.cuda.synchronize(); t = time.perf_counter(); r = f(); torch.cuda.synchronize(); return r, (time.perf_counter() - t) * 1e3
med, assign, qe = fit_kmedoids_optimized(G, K=K, seed=42)
D, ms = timed(lambda: all_pairs_geodesic_device(G))
print(f"all-pairs {N} x {N}: {ms:.1f} ms ({N * N * 4 / 2**30:.1f} GiB)", flush=True)
m = torch.from_numpy(np.asarray(med, np.int32)).to(dev); a = torch.from_numpy(np.asarray(assign, np.int32)).to(dev)
hist = [qe]
medoid_update_device(D, a, m, 2); assign_from_rows_device(D, m)       # (first-use overheads of the torch ops)
for it in range(8):
    (new, cost), t1 = timed(lambda: medoid_update_device(D, a, m, 2))
    if bool((new == m).all()): break
    m = new
    (dmin, a), t2 = timed(lambda: assign_from_rows_device(D, m))
    hist.append(float((dmin.double() ** 2).sum()))
    print(f"iteration {it + 1}: update {t1:.2f} ms, re-assignment {t2:.2f} ms, qe {hist[-1]:.1f}", flush=True)
print("qe history", hist)

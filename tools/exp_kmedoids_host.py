"""Where the host time of fit_kmedoids_optimized goes at a large size (cProfile, cumulative): usage exp_kmedoids_host.py [N] [K]"""
import sys, os, io, contextlib, cProfile, pstats, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vqvae_amd.geo.knn_graph_optimized import knn_graph_device
import vqvae_amd.geo.kmeans_optimized as km
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
K = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
dev = torch.device("cuda", 0)
z = np.random.RandomState(0).randn(N, 16).astype(np.float32)
G, _, _ = knn_graph_device(torch.from_numpy(z).to(dev), 20, mode="distance", sym="union")
with contextlib.redirect_stdout(io.StringIO()):
    km.fit_kmedoids_optimized(G, K=K, init="kpp", seed=42)
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
with contextlib.redirect_stdout(io.StringIO()):
    km.fit_kmedoids_optimized(G, K=K, init="kpp", seed=42)
torch.cuda.synchronize()
pr.disable()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(28)
print(s.getvalue())

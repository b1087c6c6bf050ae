import sys, time, ctypes, numpy as np, torch
sys.path.insert(0, '.')
from oracle import _clib, sssp as osp
from vqvae_amd._device import device
from vqvae_amd.scripts.build_codebook import build_codebook_device
from vqvae_amd.spatial_decoder import SpatialDecoder
from vqvae_amd.geo.geo_shortest_paths import sssp_multi_device
dev = device()
N, D, K = int(sys.argv[1]), 16, int(sys.argv[2])
z_h = np.random.RandomState(0).randn(N, D).astype(np.float32)
z = torch.from_numpy(z_h).to(dev)
torch.manual_seed(0)
dec = SpatialDecoder(1, (256, 128, 64), D, 28, "batch").to(dev).train()
timers = {}
t0 = time.perf_counter()
res = build_codebook_device(z, dec, k=20, sym="union", K=K, init="kpp", seed=42, batch_size=512, timers=timers)
torch.cuda.synchronize()
print("N", N, "K", K, "total s", round(time.perf_counter() - t0, 3), {k: round(v, 3) for k, v in timers.items()},
      "nnz", res["W_lcc"].nnz, "edges", res["n_edges"], "qe", res["qe"])
G = res["W_lcc"]
W = G.to_scipy()
med = res["medoids"]
Dg, _, dmin, arg, sw = sssp_multi_device(G, torch.from_numpy(med[:8].astype(np.int32)).to(dev), want_D=True, want_min=True)
Do = osp.dijkstra_multi_source(W, med[:3])
print("sssp rows bit-equal:", np.array_equal(Dg.cpu().numpy()[:3], Do), "sweeps", sw)
assign = res["assign_flat"]
print("medoids self-assigned:", bool((assign[med] == np.arange(len(med))).all()), "unique", len(set(med.tolist())) == K)
rows = 32
io = np.empty((rows, 21), np.int64); do = np.empty((rows, 21), np.float64)
_clib.lib().oracle_knn(ctypes.c_void_p(z_h.ctypes.data), N, D, 21, 1, N - rows, N, ctypes.c_void_p(io.ctypes.data), ctypes.c_void_p(do.ctypes.data))
from vqvae_amd.geo.knn_graph_optimized import knn_search_device
idx, d2 = knn_search_device(z, 21, N - rows, N)
print("knn tail rows equal:", np.array_equal(idx.cpu().numpy(), io), np.array_equal(d2.cpu().numpy(), do))

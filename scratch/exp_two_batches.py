"""Two Morton-adjacent batches (32 sources) on the swiss graph: per-sweep improvement trace (GEO_SSSP_TRACE=1)."""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vqvae_amd import _lib
from vqvae_amd.geo.knn_graph_optimized import knn_graph_device
from vqvae_amd.geo.geo_shortest_paths import sssp_multi_device
import bench
dev = torch.device('cuda', 0)
N = 60000
z = torch.from_numpy(bench.swiss_roll(N, 16, 0)).to(dev)
G, _, _ = knn_graph_device(z, 20, mode='distance', sym='union')
src = np.random.RandomState(1).choice(N, 512, replace=False).astype(np.int32)
def single(s):
    D, _, _, _, _ = sssp_multi_device(G, torch.tensor([s], dtype=torch.int32, device=dev))
    return D[0].cpu().numpy()
dA = single(int(src[0])); far = src[np.argmax(dA[src])]; dB = single(int(far))
qa = (65535 * dA[src] / dA[src].max()).astype(np.uint64); qb = (65535 * dB[src] / dB[src].max()).astype(np.uint64)
m = np.zeros(512, np.uint64)
for bit in range(15, -1, -1):
    m = (m << np.uint64(2)) | (((qa >> np.uint64(bit)) & np.uint64(1)) << np.uint64(1)) | ((qb >> np.uint64(bit)) & np.uint64(1))
order = np.argsort(m, kind='stable')
b0 = int(sys.argv[1]) if len(sys.argv) > 1 else 0
nbat = int(sys.argv[2]) if len(sys.argv) > 2 else 2
pick = src[order[b0 * 16:(b0 + nbat) * 16]]
print("spread xyz", np.ptp(z[pick.astype(np.int64)].cpu().numpy()[:, :3], axis=0), file=sys.stderr)
os.environ["GEO_SSSP_GROUP"] = "0"
_, _, dmin, amin, sweeps = sssp_multi_device(G, torch.from_numpy(pick).to(dev), want_D=False, want_min=True)
print("sweeps", sweeps, file=sys.stderr)

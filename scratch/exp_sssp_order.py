"""Does the node numbering matter for the multi-source sweep?  Same graph (bench c2 shape), same 512 sources,
nodes renumbered (a) as given, (b) by Voronoi cell of the sources, (c) reverse Cuthill-McKee."""
import sys, numpy as np, torch, scipy.sparse as sp
from scipy.sparse.csgraph import reverse_cuthill_mckee
sys.path.insert(0, '.')
from vqvae_amd import _lib
from vqvae_amd._device import DeviceCSR
from vqvae_amd.geo.knn_graph_optimized import knn_graph_device
from vqvae_amd.geo.geo_shortest_paths import sssp_multi_device
import bench
dev = torch.device('cuda', 0)
lib = _lib.load()
def prof():
    ms, l = np.zeros(1), np.zeros(1, np.int32)
    lib.geo_sssp_last_profile(ms.ctypes.data, l.ctypes.data)
    return float(ms[0]), int(l[0])
z, dec, cfg = bench.make_inputs('c2', dev)
N = z.shape[0]
G, _, _ = knn_graph_device(z, 20, mode='distance', sym='union')
W = G.to_scipy().tocsr()
src = np.random.RandomState(1).choice(N, 512, replace=False).astype(np.int32)
def run(Wp, srcp, tag):
    Gp = DeviceCSR.from_scipy(Wp, dev)
    s = torch.from_numpy(srcp.astype(np.int32)).to(dev)
    for rep in range(3):
        _, _, dmin, amin, _ = sssp_multi_device(Gp, s, want_D=False, want_min=True)
        m, l = prof()
    print(f"{tag:10s} sweeps={l} ms={m:.3f} per-sweep us={1e3*m/l:.1f}")
    return dmin.cpu().numpy(), amin.cpu().numpy()
d0, a0 = run(W, src, 'given')
def permuted(order):                       # order[new] = old
    inv = np.empty(N, np.int64); inv[order] = np.arange(N)
    Wp = W[order][:, order].tocsr(); Wp.sort_indices()
    return Wp, inv[src], inv
order = np.argsort(a0, kind='stable')
Wp, sp_, inv = permuted(order)
d1, a1 = run(Wp, sp_, 'by cell')
print('same result:', np.array_equal(d1[inv], d0), np.array_equal(a1[inv], a0))
order = np.asarray(reverse_cuthill_mckee(W, symmetric_mode=True))
Wp, sp_, inv = permuted(order)
d2, a2 = run(Wp, sp_, 'rcm')
print('same result:', np.array_equal(d2[inv], d0), np.array_equal(a2[inv], a0))
# cells ordered by a 1-D embedding of the medoids (first latent coordinate of the source)
zc = z.cpu().numpy()
key = zc[src[a0], 0]
order = np.lexsort((np.arange(N), a0, key))
Wp, sp_, inv = permuted(order)
d3, a3 = run(Wp, sp_, 'cell+coord')
print('same result:', np.array_equal(d3[inv], d0), np.array_equal(a3[inv], a0))

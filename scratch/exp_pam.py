"""PAM swap pass at the 60 000-latent size: time of the kernel alone (HIP events) and of the host-side preparation."""
import os, sys, time, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vqvae_amd import _lib
from vqvae_amd._device import device, ptr, stream_ptr
from vqvae_amd.geo.geo_shortest_paths import all_pairs_geodesic_device
from vqvae_amd.geo.kmeans_optimized import assign_from_rows_device, fit_kmedoids_optimized, pam_swap_pass_device
from vqvae_amd.geo.knn_graph_optimized import knn_graph_device
dev = device()
n, K = 60000, 512
z = torch.from_numpy(np.random.RandomState(0).randn(n, 16).astype(np.float32)).to(dev)
G, _, _ = knn_graph_device(z, 20, mode="distance", sym="union")
med, _, _ = fit_kmedoids_optimized(G, K=K, init="kpp", seed=42)
D = all_pairs_geodesic_device(G)
m = torch.from_numpy(med.astype(np.int32)).to(dev)
for rep in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    delta, i, x, total = pam_swap_pass_device(D, m, 2)
    torch.cuda.synchronize(); print(f"whole pass {1e3 * (time.perf_counter() - t0):.2f} ms  delta {delta:.3f}")
# kernel alone
lib = _lib.load()
dmin, near = assign_from_rows_device(D, m); near = near.long()
rows = D[m.long()].clone(); rows.scatter_(0, near[None, :], float('inf')); d2 = rows.min(dim=0).values.contiguous(); d1 = dmin.contiguous()
is_med = torch.zeros(n, dtype=torch.uint8, device=dev); is_med[m.long()] = 1
gain = d2.double() ** 2 - dmin.double() ** 2
base = torch.segment_reduce(gain[torch.argsort(near, stable=True)], 'sum', lengths=torch.bincount(near, minlength=K)).contiguous()
n32 = near.to(torch.int32).contiguous()
best = torch.empty(n, dtype=torch.float64, device=dev); which = torch.empty(n, dtype=torch.int32, device=dev)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for rep in range(3):
    e0.record()
    _lib.check(lib.geo_pam_swap_deltas(ptr(D), D.stride(0), ptr(n32), ptr(d1), ptr(d2), ptr(base), ptr(is_med), n, K, 2, ptr(best), ptr(which), stream_ptr()), "pam")
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1)
    print(f"kernel alone {ms:.2f} ms = {n * n * 4 / ms / 1e6:.0f} GB/s of matrix bytes")

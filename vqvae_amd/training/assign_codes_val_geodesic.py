"""Geodesic code assignment of latents that are NOT nodes of the training graph (validation / test sets).

The reference's result notes name `src/training/assign_codes_val_geodesic.py` ("robust geodesic assignment with
connectivity safeguards", docs/results/cifar10_quantization_analysis.md:147: it replaced a Euclidean nearest-medoid
assignment of the validation set) but the file is not in its repository.  This is a definition of that step in terms of
the pieces the training path already has, not a restatement (no oracle in the reference):

  1. every new latent v is attached to its k nearest graph nodes (exact fp64 squared Euclidean distances, ties by index);
     only nodes of the graph handed in take part -- with the LCC graph of build_codebook_device that is the
     connectivity safeguard: a neighbour outside the connected component can never carry the path;
  2. an attachment edge is as long as the training edges are: the decoder pull-back length
     (geo/riemannian_metric.py, same kernels, chunks of `batch_size` edges) when a decoder is given, else Euclidean;
  3. dist(v, medoid m) = min over the k attachments u of  len(v, u) + D[m][u]  with D the medoids' geodesic rows on the
     training graph (geo_sssp_multi); the code of v is the first medoid with the smallest distance
     (geo_attach_argmin).  A latent that coincides with a graph node and is attached along graph edges only (k not larger
     than the graph's own k) gets exactly that node's assignment and distance.

numpy restatement: oracle/pipeline.py (assign_new_latents); tests: tests/test_gpu_voronoi.py.
"""
from typing import Dict, Optional

import numpy as np
import torch

from .. import _lib
from .._device import DeviceCSR, ptr, stream_ptr
from ..geo.geo_shortest_paths import sssp_multi_device


def attach_neighbors_device(z_new: torch.Tensor, z_graph: torch.Tensor, k: int, rows_per_block: int = 0):
    """(idx int32 [V, k'], d2 f64 [V, k']) with k' = min(k, n): the k' nearest graph nodes of every new latent, ranked by
    the fp64 squared distance accumulated dimension by dimension (sum_k (a_k - b_k)^2, ascending k), ties by node index."""
    V, n, d = int(z_new.shape[0]), int(z_graph.shape[0]), int(z_graph.shape[1])
    kk = min(int(k), n)
    dev = z_graph.device
    idx = torch.empty((V, kk), dtype=torch.int32, device=dev)
    d2 = torch.empty((V, kk), dtype=torch.float64, device=dev)
    if V == 0 or kk == 0:
        return idx, d2
    b64 = z_graph.to(torch.float64)
    step = rows_per_block or max(1, (1 << 26) // max(1, n))            # <= 512 MiB of fp64 distances per block
    for r0 in range(0, V, step):
        a64 = z_new[r0:r0 + step].to(device=dev, dtype=torch.float64)
        acc = torch.zeros((a64.shape[0], n), dtype=torch.float64, device=dev)
        for c in range(d):                                              # one rounding per dimension, the oracle's order
            diff = a64[:, c:c + 1] - b64[:, c].unsqueeze(0)
            acc += diff * diff
        vals, order = torch.sort(acc, dim=1, stable=True)              # stable: equal distances keep ascending index
        idx[r0:r0 + step] = order[:, :kk].to(torch.int32)
        d2[r0:r0 + step] = vals[:, :kk]
    return idx, d2


def assign_codes_geodesic(z_new: torch.Tensor, z_graph: torch.Tensor, G: DeviceCSR, medoids, *, k: int = 20,
                          decoder=None, batch_size: int = 512, D_rows: Optional[torch.Tensor] = None) -> Dict:
    """z_new f32 [V, d] (any device), z_graph f32 [n, d] = the latents of G's nodes on the GPU, G = the geodesic training
    graph (e.g. W_lcc of build_codebook_device), medoids = node indices of G.  Returns a dict: codes int64 [V] (position
    in `medoids`), dist f32 [V], neighbors int32 [V, k], lengths f32 [V, k]."""
    lib = _lib.load()
    dev = z_graph.device
    z_new = z_new.to(device=dev, dtype=torch.float32).contiguous()
    V, n = int(z_new.shape[0]), G.n
    assert int(z_graph.shape[0]) == n, "z_graph must hold one latent per node of G"
    med = torch.from_numpy(np.asarray(medoids, dtype=np.int32)).to(dev)
    K = int(med.numel())
    idx, d2 = attach_neighbors_device(z_new, z_graph, k)
    kk = int(idx.shape[1])
    if decoder is not None and V > 0:
        from ..geo.riemannian_metric import edge_lengths_graph_device, edge_lengths_riemannian
        from ..spatial_decoder import DecoderExport, hip_kernels_cover
        z_cat = torch.cat([z_graph, z_new], dim=0).contiguous()
        src = (n + torch.arange(V, dtype=torch.int32, device=dev)).repeat_interleave(kk)
        dst = idx.reshape(-1).contiguous()
        if hip_kernels_cover(decoder):
            lengths = edge_lengths_graph_device(DecoderExport(decoder, dev), z_cat, src, dst, batch_size)
        else:
            lengths = edge_lengths_riemannian(decoder, z_cat[src.long()], z_cat[dst.long()], batch_size)
        lengths = lengths.reshape(V, kk).contiguous()
    else:
        lengths = torch.sqrt(d2).to(torch.float32).contiguous()
    if D_rows is None:
        D_rows, _, _, _, _ = sssp_multi_device(G, med, want_D=True)      # [K, n]
    Dt = D_rows.t().contiguous()                                         # [n, K]: a node's K distances are contiguous
    dist = torch.empty(V, dtype=torch.float32, device=dev)
    arg = torch.empty(V, dtype=torch.int32, device=dev)
    if V > 0:
        with torch.cuda.device(dev):
            _lib.check(lib.geo_attach_argmin(ptr(Dt), Dt.stride(0), K, ptr(idx), ptr(lengths), kk, V, ptr(dist), ptr(arg),
                                             stream_ptr()), "geo_attach_argmin")
    return {"codes": arg.to(torch.int64), "dist": dist, "neighbors": idx, "lengths": lengths}

"""world_size-2 gloo tests of vqvae_amd/parallel.py (the N>1 path of SURVEY 8e) on CPU: the sharding,
all-gather and (min, lowest-index) merge logic is the code the GPU ranks run; the local compute is the
CPU oracle here and the HIP kernels there.  Every sharded result must equal the single-process one."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _inputs():
    from oracle import knn as okn
    from oracle import metric as om
    z = np.random.RandomState(0).randn(301, 16).astype(np.float32)
    W, _ = okn.build_knn_graph(z, k=6, mode="distance", sym="union")
    sd = om.make_decoder_state(2, 16, 1, norm_type="batch")
    r = np.random.RandomState(1)
    src, dst = r.randint(0, 301, 333), r.randint(0, 301, 333)
    medoids = np.random.RandomState(2).choice(301, size=13, replace=False)
    return z, W, sd, src, dst, medoids


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import ctypes
        from oracle import _clib, metric as om, sssp as osp
        from vqvae_amd import parallel as par
        z, W, sd, src, dst, medoids = _inputs()
        zt = torch.from_numpy(z)

        def search(zz, nn, r0, r1):
            idx = np.empty((r1 - r0, nn), np.int64)
            d2 = np.empty((r1 - r0, nn), np.float64)
            zz = np.ascontiguousarray(zz.numpy())
            _clib.lib().oracle_knn(ctypes.c_void_p(zz.ctypes.data), zz.shape[0], zz.shape[1], nn, 1, r0, r1,
                                   ctypes.c_void_p(idx.ctypes.data), ctypes.c_void_p(d2.ctypes.data))
            return torch.from_numpy(idx), torch.from_numpy(d2)

        lo, hi = par.block_range(len(z), rank, world)
        z_full = par.gather_latents(zt[lo:hi], len(z))
        idx, d2 = par.sharded_knn(z_full, 7, search)
        idx_c, d2_later = par.sharded_knn(z_full, 7, search, gather_d2=False)    # connectivity graphs: no fp64 gather ...
        assert torch.equal(idx_c, idx) and callable(d2_later)
        assert torch.equal(d2_later(), d2)                                        # ... unless asked for (collective)

        def lengths(e0, e1):
            return om.edge_lengths(sd, "batch", 28, z[src[e0:e1]], z[dst[e0:e1]], batch_size=64, training=True)

        L = par.sharded_edge_lengths(len(src), 64, lengths)

        def solve(s0, s1):
            if s1 <= s0:
                return torch.full((W.shape[0],), float("inf")), torch.zeros(W.shape[0], dtype=torch.int32)
            D = osp.dijkstra_multi_source(W, medoids[s0:s1])
            return torch.from_numpy(D.min(axis=0)), torch.from_numpy(D.argmin(axis=0).astype(np.int32))

        dmin, arg = par.sharded_assign(len(medoids), solve)
        np.savez(os.path.join(out_dir, f"rank{rank}.npz"), z_full=z_full.numpy(), idx=idx.numpy(), d2=d2.numpy(),
                 L=L.numpy(), dmin=dmin.numpy(), arg=arg.numpy())
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_path_equals_single_process(tmp_path, world):
    from oracle import knn as okn
    from oracle import metric as om
    from oracle import sssp as osp
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    z, W, sd, src, dst, medoids = _inputs()
    dist_ref, idx_ref = okn.knn_search(z.astype(np.float32), 7) if z.shape[1] > 15 else (None, None)
    L_ref = om.edge_lengths(sd, "batch", 28, z[src], z[dst], batch_size=64, training=True).numpy()
    D = osp.dijkstra_multi_source(W, medoids)
    for rank in range(world):
        got = np.load(os.path.join(str(tmp_path), f"rank{rank}.npz"))
        np.testing.assert_array_equal(got["z_full"], z)
        np.testing.assert_array_equal(got["idx"], idx_ref)
        np.testing.assert_array_equal(np.sqrt(got["d2"]), dist_ref)
        np.testing.assert_array_equal(got["L"], L_ref)                   # chunk-aligned shards: same BN batches
        np.testing.assert_array_equal(got["dmin"], D.min(axis=0))
        np.testing.assert_array_equal(got["arg"], D.argmin(axis=0))      # first index on ties across ranks


def test_partitions_and_merge_rules():
    from vqvae_amd import parallel as par
    for n in (0, 1, 7, 8, 100):
        for world in (1, 2, 3, 8):
            spans = [par.block_range(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            assert max(b - a for a, b in spans) - min(b - a for a, b in spans) <= 1
    for E, B in ((0, 512), (100, 512), (1025, 512), (5000, 100)):
        for world in (1, 2, 8):
            spans = [par.chunk_range(E, B, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == E
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            assert all(a % B == 0 for a, _ in spans if a < E)                # shards start on chunk boundaries
    d, a = torch.tensor([1.0, float("inf")]), torch.tensor([3, 0], dtype=torch.int32)
    assert par.merge_min_argmin(d, a, [0])[1].tolist() == [3, 0]          # world 1: identity

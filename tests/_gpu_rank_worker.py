"""One rank of the multi-rank GPU test (tests/test_gpu_multirank.py starts N fresh copies of this script, all on the
box's GPU(s), rendezvous over gloo on 127.0.0.1): the sharded hot path on HIP kernels -- gather_latents, sharded kNN,
sharded edge lengths (whole BatchNorm chunks per rank), the replicated k-means++ chain, sharded assignment solve --
on C1 (2 048 latents, d=16, k=20, K=64).  Writes this rank's results to <out_dir>/rank<r>.npz."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main(out_dir):
    from oracle import metric as om
    from oracle import synthetic as syn
    from vqvae_amd import parallel as par
    from vqvae_amd.geo.geo_shortest_paths import sssp_multi_device
    from vqvae_amd.scripts.build_codebook import build_codebook_device
    from vqvae_amd.spatial_decoder import SpatialDecoder
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dev = torch.device("cuda", int(os.environ.get("LOCAL_RANK", "0")) % torch.cuda.device_count())
    torch.cuda.set_device(dev)
    backend = os.environ.get("GEO_TEST_BACKEND", "gloo")
    if backend == "nccl":
        dist.init_process_group("nccl", device_id=dev)
    else:
        dist.init_process_group(backend)
    try:
        n, d, K = (int(v) for v in os.environ.get("GEO_TEST_SHAPE", "2048,16,64").split(","))
        z_h = syn.gauss_latents(n, d, 0)
        sd = om.make_decoder_state(0, d, 1, norm_type="batch")
        dec = SpatialDecoder(1, (256, 128, 64), d, 28, "batch")
        dec.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()})
        dec = dec.to(dev).train()
        if os.environ.get("GEO_TEST_MODE") == "in_flight":
            # five latent sets of different sizes, three sharded builds in flight on every rank (ticket-ordered collectives)
            from vqvae_amd.scripts.build_codebook import build_codebooks_pipelined
            sizes = [n, n - 301, n // 2, n - 77, n // 2 + 13]
            sets = [torch.from_numpy(syn.gauss_latents(m, d, 10 + i)).to(dev) for i, m in enumerate(sizes)]
            out = build_codebooks_pipelined(sets, dec, depth=3, k=20, sym="union", K=K, init="kpp", seed=42, batch_size=512)
            np.savez(os.path.join(out_dir, f"rank{rank}.npz"),
                     **{f"{i}/{key}": np.asarray(r[key]) for i, r in enumerate(out) for key in ("medoids", "assign_flat", "qe")},
                     **{f"{i}/lengths": r["edge_lengths"].cpu().numpy() for i, r in enumerate(out)},
                     sharded=np.array([int(all(r["sharded"]["knn"] and r["sharded"]["jvp"] for r in out))]))
            return
        lo, hi = par.block_range(n, rank, world)
        z = par.gather_latents(torch.from_numpy(z_h[lo:hi]).to(dev), n)        # latents arrive row-sharded
        res = build_codebook_device(z, dec, k=20, sym="union", K=K, init="kpp", seed=42, batch_size=512)
        G = res["W_lcc"]
        src = torch.from_numpy(res["medoids"].astype(np.int32)).to(dev)

        def solve(s0, s1):
            if s1 <= s0:
                return (torch.full((G.n,), float("inf"), device=dev), torch.zeros(G.n, dtype=torch.int32, device=dev))
            _, _, dmin, arg, _ = sssp_multi_device(G, src[s0:s1].contiguous(), want_D=False, want_min=True)
            return dmin, arg

        dmin, arg = par.sharded_assign(K, solve)
        np.savez(os.path.join(out_dir, f"rank{rank}.npz"), z=z.cpu().numpy(), indptr=G.indptr.cpu().numpy(),
                 indices=G.indices.cpu().numpy(), data=G.data.cpu().numpy(), lengths=res["edge_lengths"].cpu().numpy(),
                 medoids=res["medoids"], assign=res["assign_flat"], qe=np.float64(res["qe"]),
                 dmin=dmin.cpu().numpy(), arg=arg.cpu().numpy(),
                 sharded=np.array([int(res["sharded"]["knn"]), int(res["sharded"]["jvp"])]),
                 **{"bn/" + k: v.cpu().numpy() for k, v in dec.state_dict().items() if "running" in k or "tracked" in k})
    finally:
        dist.destroy_process_group()


if __name__ == "__main__":
    main(sys.argv[1])

"""Generate tests/golden/*.npz by IMPORTING THE REFERENCE (build container only) and check the
oracle restatement against it on the same inputs.  TEST INFRASTRUCTURE ONLY.

    python -m oracle.gen_golden            # from the repo root; needs /root/reference

The reference never travels: only the input seeds / small inputs and the reference's outputs are
stored.  Every fixture records how its inputs are regenerated (numpy RandomState seeds).
"""
import argparse
import contextlib
import io
import os
import sys
import tempfile
import types

import numpy as np
import torch
from scipy import sparse

REF = os.environ.get("GEO_REFERENCE_ROOT", "/root/reference")
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
GOLD = os.path.join(ROOT, "tests", "golden")


def latents(N, d, seed, scale=1.0):
    return (np.random.RandomState(seed).randn(N, d) * scale).astype(np.float32)


def clustered_latents(N, d, seed):
    """A few tight clusters plus exact duplicate points (kNN tie / zero-distance edge cases)."""
    r = np.random.RandomState(seed)
    centres = r.randn(6, d).astype(np.float32) * 3
    z = (centres[r.randint(0, 6, size=N)] + 0.05 * r.randn(N, d)).astype(np.float32)
    z[5] = z[3]
    z[N - 1] = z[N // 2]
    z[17] = z[16] = z[15]
    return z


def quiet(fn, *a, **kw):
    with contextlib.redirect_stdout(io.StringIO()):
        return fn(*a, **kw)


def csr_parts(W):
    W = sparse.csr_matrix(W)
    W.sort_indices()
    return W.indptr.astype(np.int32), W.indices.astype(np.int32), W.data.astype(np.float32)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=GOLD)
    args = ap.parse_args()
    os.makedirs(args.out, exist_ok=True)
    sys.path.insert(0, REF)
    sys.dont_write_bytecode = True
    from src.geo import knn_graph_optimized as rk
    from src.geo import geo_shortest_paths as rs
    from src.geo import kmeans_optimized as rm
    from src.geo import riemannian_metric as rr
    from src.models.spatial_vae import SpatialDecoder, SpatialVAE

    sys.path.insert(0, ROOT)
    from oracle import kmedoids as ok
    from oracle import knn as okn
    from oracle import metric as om
    from oracle import pipeline as op
    from oracle import sssp as osp

    report = []

    def note(msg):
        print(msg)
        report.append(msg)

    # ------------------------------------------------------------------ G2: kNN graphs
    knn_cases = {"g16": (2048, 16, 0, "gauss"), "g32": (512, 32, 1, "gauss"), "g64": (300, 64, 2, "gauss"),
                 "g8": (400, 8, 3, "gauss"), "dup16": (256, 16, 4, "clustered")}
    out = {}
    for name, (N, d, seed, kind) in knn_cases.items():
        z = latents(N, d, seed) if kind == "gauss" else clustered_latents(N, d, seed)
        out[f"{name}/meta"] = np.array([N, d, seed, 0 if kind == "gauss" else 1])
        for k in (1, 5, 20):
            for mode in ("connectivity", "distance"):
                for sym in ("union", "mutual"):
                    Wr, info = quiet(rk.build_knn_graph_auto, z, k=k, mode=mode, sym=sym)
                    Wo, info_o = okn.build_knn_graph_auto(z, k=k, mode=mode, sym=sym)
                    ip, ix, dat = csr_parts(Wr)
                    ipo, ixo, dato = csr_parts(Wo)
                    same_struct = np.array_equal(ip, ipo) and np.array_equal(ix, ixo)
                    same_sets = all(set(a) == set(b) for a, b in zip(info["indices"], info_o["indices"]))
                    ulp = 0.0
                    if same_struct and mode == "distance" and len(dat):
                        ulp = float(np.max(np.abs(dat.astype(np.float64) - dato) / np.spacing(np.abs(dat))))
                    note(f"knn {name} k={k} {mode}/{sym}: nnz={Wr.nnz} struct_equal={same_struct} "
                         f"neighbour_sets_equal={same_sets} max_ulp={ulp:.2f}")
                    tag = f"{name}/k{k}/{mode}/{sym}"
                    out[f"{tag}/indptr"], out[f"{tag}/indices"] = ip, ix
                    if mode == "distance":
                        out[f"{tag}/data"] = dat
            out[f"{name}/k{k}/nbr_indices"] = info["indices"].astype(np.int32)
            out[f"{name}/k{k}/nbr_distances"] = info["distances"].astype(np.float32)
    np.savez_compressed(os.path.join(args.out, "knn.npz"), **out)

    # ------------------------------------------------------------------ G5: shortest paths
    out = {}
    z = latents(2048, 16, 0)
    Wd, _ = quiet(rk.build_knn_graph_auto, z, k=20, mode="distance", sym="union")
    srcs = np.array([0, 7, 100, 1023, 2047, 512, 3, 1999])
    Dr, Pr = rs.dijkstra_multi_source(Wd, srcs, return_predecessors=True)
    Do, Po = osp.dijkstra_multi_source(Wd, srcs, return_predecessors=True)
    note(f"sssp g16 union: D bit-equal={np.array_equal(Dr, Do)} pred_equal_frac={np.mean(Pr == Po):.4f}")
    out["g16/sources"], out["g16/D"], out["g16/P"] = srcs, Dr, Pr
    Du = rs.dijkstra_multi_source(Wd, srcs[:3], unweighted=True)
    note(f"sssp g16 unweighted: bit-equal={np.array_equal(Du, osp.dijkstra_multi_source(Wd, srcs[:3], unweighted=True))}")
    out["g16/D_unweighted"] = Du
    zs = latents(240, 12, 1)
    Wm, _ = quiet(rk.build_knn_graph_auto, zs, k=1, mode="distance", sym="mutual")
    Dm = rs.dijkstra_multi_source(Wm, [0, 10])
    note(f"sssp disconnected (N=240,k=1,mutual): bit-equal={np.array_equal(Dm, osp.dijkstra_multi_source(Wm, [0, 10]))} "
         f"has_inf={bool(np.isinf(Dm).any())}")
    out["disc/D"] = Dm
    lcc_r = rk.largest_connected_component(Wm)
    note(f"lcc disconnected: mask equal={np.array_equal(lcc_r, okn.largest_connected_component(Wm))} size={lcc_r.sum()}")
    out["disc/lcc"] = lcc_r
    # directed solve on an asymmetric matrix (upper triangle only)
    Wt = sparse.triu(Wd).tocsr()
    Dd = rs.dijkstra_multi_source(Wt, [0, 5], directed=True)
    Dnd = rs.dijkstra_multi_source(Wt, [0, 5], directed=False)
    note(f"sssp triu directed: bit-equal={np.array_equal(Dd, osp.dijkstra_multi_source(Wt, [0, 5], directed=True))} "
         f"undirected-on-triu: bit-equal={np.array_equal(Dnd, osp.dijkstra_multi_source(Wt, [0, 5], directed=False))}")
    out["g16/D_triu_directed"], out["g16/D_triu_undirected"] = Dd, Dnd
    np.savez_compressed(os.path.join(args.out, "sssp.npz"), **out)

    # ------------------------------------------------------------------ G6: k-medoids
    out = {}
    for gname, W in (("g16", Wd), ("disc", Wm)):
        for K in (1, 8, 64):
            for init in ("kpp", "random"):
                for seed in (0, 42):
                    mr, ar, qr = quiet(rm.fit_kmedoids_optimized, W, K=K, init=init, seed=seed)
                    mo, ao, qo = ok.fit_kmedoids_optimized(W, K=K, init=init, seed=seed)
                    line = (f"kmedoids {gname} K={K} {init} seed={seed}: medoids={np.array_equal(mr, mo)} "
                            f"assign={np.array_equal(ar, ao)} qe_equal={qr == qo or (np.isinf(qr) and np.isinf(qo))}")
                    if init == "kpp":
                        ms, as_, qs = ok.fit_kmedoids_single_pass(W, K=K, seed=seed)
                        line += (f" | single-pass medoids={np.array_equal(mr, ms)} assign={np.array_equal(ar, as_)} "
                                 f"qe_equal={qr == qs or (np.isinf(qr) and np.isinf(qs))}")
                    note(line)
                    tag = f"{gname}/K{K}/{init}/s{seed}"
                    out[f"{tag}/medoids"], out[f"{tag}/assign"] = mr.astype(np.int32), ar.astype(np.int32)
                    out[f"{tag}/qe"] = np.float64(qr)
    np.savez_compressed(os.path.join(args.out, "kmedoids.npz"), **out)

    # ------------------------------------------------------------------ G3/G4: decoder JVP edge lengths
    out = {}
    dec_cases = {"fm_batch": (16, 1, 28, "batch", 10), "fm_none": (16, 1, 28, "none", 11),
                 "fm_group": (16, 1, 28, "group", 12), "cf_batch": (32, 3, 32, "batch", 13)}
    E = 2048
    for name, (d, cout, size, norm, seed) in dec_cases.items():
        sd = om.make_decoder_state(seed, d, cout, norm_type=norm)
        dec = SpatialDecoder(cout, (256, 128, 64), d, size, norm)
        dec.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()})
        r = np.random.RandomState(100 + seed)
        zs_ = r.randn(E, d).astype(np.float32)
        ze_ = (zs_ + 0.3 * r.randn(E, d)).astype(np.float32)
        out[f"{name}/meta"] = np.array([d, cout, size, seed, 100 + seed, E])
        for training in (True, False):
            for bs in (512, 100):
                # train-mode JVPs mutate the BatchNorm running statistics: start every run from the fixture
                dec.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()})
                dec.train(training)
                Lr = rr.edge_lengths_riemannian(dec, torch.from_numpy(zs_), torch.from_numpy(ze_), batch_size=bs).numpy()
                Lo = om.edge_lengths(sd, norm, size, zs_, ze_, batch_size=bs, training=training).numpy()
                L64 = om.edge_lengths(sd, norm, size, zs_, ze_, batch_size=bs, training=training,
                                      dtype=torch.float64).numpy()
                rel = np.abs(Lo - Lr) / np.abs(Lr)
                rel64 = np.abs(L64 - Lr) / np.abs(Lr)
                note(f"jvp {name} train={training} bs={bs}: oracle-f32 vs ref max_rel={rel.max():.2e} "
                     f"p99={np.quantile(rel, 0.99):.2e}; oracle-f64 vs ref max_rel={rel64.max():.2e} "
                     f"p99={np.quantile(rel64, 0.99):.2e}")
                out[f"{name}/train{int(training)}/bs{bs}"] = Lr
    np.savez_compressed(os.path.join(args.out, "metric.npz"), **out)

    # ------------------------------------------------------------------ G7: CLI end to end (config C1)
    out = {}
    for name, (d, cout, size, norm, seed, n_img) in {"c1_fm": (16, 1, 28, "batch", 0, 128)}.items():
        sd = om.make_decoder_state(seed, d, cout, norm_type=norm)
        z4 = np.random.RandomState(seed).randn(n_img * 16, d).astype(np.float32).reshape(n_img, 4, 4, d)
        z4 = np.ascontiguousarray(np.transpose(z4, (0, 3, 1, 2)))          # (N,C,H,W), row (n,h,w) = randn row
        with tempfile.TemporaryDirectory() as tmp:
            torch.manual_seed(0)
            vae = SpatialVAE(in_channels=cout, enc_channels=[64, 128, 256], dec_channels=[256, 128, 64],
                             latent_dim=d, recon_loss="mse", output_image_size=size, norm_type=norm,
                             mse_use_sigmoid=True)
            vae.decoder.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()})
            torch.save({"model_state_dict": vae.state_dict(), "epoch": 0}, os.path.join(tmp, "best.pt"))
            torch.save(torch.from_numpy(z4), os.path.join(tmp, "z.pt"))
            from src.scripts import build_codebook as rb
            a = types.SimpleNamespace(latents_path=os.path.join(tmp, "z.pt"), out_dir=os.path.join(tmp, "out"),
                                      vae_ckpt_path=os.path.join(tmp, "best.pt"), in_channels=cout,
                                      output_image_size=size, latent_dim=d, enc_channels=[64, 128, 256],
                                      dec_channels=[256, 128, 64], recon_loss="mse", norm_type=norm,
                                      mse_use_sigmoid=True, k=20, sym="union", K=64, init="kpp", seed=42,
                                      batch_size=512)
            quiet(rb.main, a)
            codes = np.load(os.path.join(tmp, "out", "codes.npy"))
            cb = torch.load(os.path.join(tmp, "out", "codebook.pt"), weights_only=False)
            Wl = sparse.load_npz(os.path.join(tmp, "out", "knn_graph_geodesic.npz"))
        res = op.build_codebook(z4, sd, norm, size, k=20, sym="union", K=64, init="kpp", seed=42, batch_size=512,
                                training=True)
        ip, ix, dat = csr_parts(Wl)
        ipo, ixo, dato = csr_parts(res["W_lcc"])
        struct = np.array_equal(ip, ipo) and np.array_equal(ix, ixo)
        rel = np.abs(dat - dato) / np.abs(dat) if struct else np.array([np.inf])
        note(f"cli {name}: codes equal={np.array_equal(codes, res['codes'])} medoids equal="
             f"{np.array_equal(cb['medoid_indices'], res['medoid_indices'])} z_medoid equal="
             f"{np.array_equal(cb['z_medoid'].numpy(), res['z_medoid'])} graph struct equal={struct} "
             f"weights max_rel={rel.max():.2e} codes dtype={codes.dtype} shape={codes.shape} "
             f"medoid dtype={cb['medoid_indices'].dtype}")
        out[f"{name}/meta"] = np.array([d, cout, size, seed, n_img])
        out[f"{name}/codes"], out[f"{name}/medoid_indices"] = codes, cb["medoid_indices"]
        out[f"{name}/z_medoid"] = cb["z_medoid"].numpy()
        out[f"{name}/indptr"], out[f"{name}/indices"], out[f"{name}/data"] = ip, ix, dat
        out[f"{name}/config_keys"] = np.array(sorted(cb["config"].keys()))
    np.savez_compressed(os.path.join(args.out, "cli.npz"), **out)

    with open(os.path.join(args.out, "REPORT.txt"), "w") as f:
        f.write("oracle vs reference import (oracle/gen_golden.py), build container\n")
        f.write("\n".join(report) + "\n")


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""bench.py -- latents/s through the geodesic-codebook hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload c2|c1|c3|c3d32|c4|swiss|real|c5cb|c5prior|c2pam]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

`--gpus N` with N > 1 and no WORLD_SIZE in the environment starts the N ranks itself (one child process per GPU
under torch.distributed.run, RCCL backend) BEFORE this process touches the GPU, and relays rank 0's JSON line.

One step = one pass of the hot path over one batch of synthetic latents already resident in HBM
(BASELINE.json configs[1]: 60 000 latents, d=16, k=20, K=512, FashionMNIST-shaped decoder with
train-mode BatchNorm as the reference CLI leaves it):
    kNN graph -> upper edge list -> decoder pull-back edge lengths -> LCC -> k-means++ chain
    (K single-source solves with fused running argmin) -> the reference's assignment stage
    (K-source batched solve + column argmin = the "APSP + K-medoids sweep" of BASELINE.json).
Rank 0 prints ONE JSON line (metric/value/roofline/cpu_baseline ...); everything else goes to stderr.
"""
import argparse
import contextlib
import json
import os
import sys
import time


DEFAULT_CPU_CAP = 16


def host_cores() -> int:
    """CPU threads this process may really use (the GPU box gives a 1-GPU job a share of the host)."""
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    # The pool's 1-GPU boxes give a job a CPU share of 16 threads while the affinity mask lists every core of the host: with the
    # mask's 256 threads the OpenMP / torch-CPU stages of the baseline oversubscribe that share and crawl (measured: the default
    # run no longer finished within five minutes).  Default cap 16, stated in the line as "capped_at"; GEO_BENCH_CPU_THREADS overrides.
    return max(1, min(avail, int(os.environ.get("GEO_BENCH_CPU_THREADS", str(DEFAULT_CPU_CAP)))))


os.environ.setdefault("OMP_NUM_THREADS", str(host_cores()))      # before numpy/torch/OpenMP start their pools

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

WORKLOADS = {
    # name: (n_latents, d, out_channels, image_size, k, K)
    "c1": (2048, 16, 1, 28, 20, 64),
    "c2": (60000, 16, 1, 28, 20, 512),
    "c3": (50000, 64, 3, 32, 20, 512),          # BASELINE config 3 as stated: d=64, 32-px decoder
    "c3d32": (50000, 32, 3, 32, 20, 512),       # the reference's shipped CIFAR config (latent_dim 32)
    "c4": (1000000, 16, 1, 28, 20, 1024),       # BASELINE config 4's workload (any number of GPUs)
    # second latent distribution -- noisy 2-D swiss roll embedded in d dims (long geodesics,
    # many relaxation sweeps); same decoder / k / K as c2
    "swiss": (60000, 16, 1, 28, 20, 512),
    # the shapes the reference's pipeline really feeds build_codebook.py (SURVEY finding 7: N*H*W nodes, build_codebook.py:35):
    "real": (960000, 16, 1, 28, 20, 512),       # FashionMNIST: 60 000 images x 4x4 cells, d=16, K=512
    "c5cb": (800000, 32, 3, 32, 20, 512),       # CIFAR-10 (BASELINE config 5's codebook stage): 50 000 x 4x4 cells, d=32, 32-px decoder
    "c5prior": None,                            # BASELINE config 5's second stage: prior training tokens/s (prior_bench)
    "c2pam": None,                              # extension: dense all-pairs matrix + PAM swap sweeps over it (pam_bench)
}
HBM_PEAK_GBS = 8000.0       # MI355X_MICROARCH.md: HBM3E 8 TB/s
FP64_PEAK_TFLOPS = 78.6     # fp64 vector peak = half the 157.3 TFLOP/s f32 vector peak of MI355X_MICROARCH.md
F32_MFMA_PEAK_TFLOPS = 157.3
BF16_MFMA_PEAK_TFLOPS = 2500.0   # dense bf16 (the JVP's ConvT2 / ConvT3 products run as 6 bf16 MFMAs per f32 product)


def jvp_flop_per_edge(d, cout, size, channels=(256, 128, 64)):
    """f32 flop of one edge of the pull-back metric from the decoder's shape: 2 endpoints x (primal + tangent) x 2 flop per
    MAC of conv1x1(d->c0), ConvT(c0->c1) 1x1->2x2 (4 taps), ConvT(c1->c2) 2x2->4x4 (36 valid input/output pixel pairs),
    ConvT(c2->cout) 4x4->4x4 (28-px head, padding 3: 64 pairs) or 4x4->8x8 (32-px head, padding 1: 196 pairs).
    d=16, 28-px, 1 channel: 434 176 MAC -> 3.47 MFLOP."""
    c0, c1, c2 = channels
    pairs3 = 64 if int(size) == 28 else 196
    mac = d * c0 + 4 * c0 * c1 + 36 * c1 * c2 + pairs3 * c2 * cout
    return 8.0 * mac


def cpu_model():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.lower().startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def swiss_roll(n, d, seed):
    """Noisy 2-D swiss roll in the first three coordinates, small isotropic noise in all d (unit-ish scale)."""
    r = np.random.RandomState(seed)
    t = 1.5 * np.pi * (1.0 + 2.0 * r.rand(n))
    h = 21.0 * r.rand(n)
    x = np.zeros((n, d), dtype=np.float64)
    x[:, 0], x[:, 1], x[:, 2] = t * np.cos(t), h, t * np.sin(t)
    x = x / 7.0 + 0.02 * r.randn(n, d)
    return x.astype(np.float32)


def make_inputs(name, dev):
    from vqvae_amd.spatial_decoder import SpatialDecoder
    n, d, cout, size, k, K = WORKLOADS[name]
    z = torch.from_numpy(swiss_roll(n, d, 0) if name == "swiss"
                         else np.random.RandomState(0).randn(n, d).astype(np.float32)).to(dev)
    torch.manual_seed(0)
    dec = SpatialDecoder(cout, (256, 128, 64), d, size, "batch").to(dev).train()    # random init, BN in train mode
    return z, dec, dict(n=n, d=d, k=k, K=K, size=size, cout=cout)


def hot_path_step(z, dec, cfg, timers, rank, world, group=None):
    """One full pass; returns (result dict, sweep profile)."""
    from vqvae_amd import _lib
    from vqvae_amd.geo.geo_shortest_paths import sssp_multi_device
    from vqvae_amd.scripts.build_codebook import build_codebook_device
    from vqvae_amd.parallel import block_range, gather_latents, sharded_assign
    if world > 1:       # latents arrive row-sharded; the kNN corpus is replicated by one all-gather
        lo, hi = block_range(z.shape[0], rank, world)
        z = gather_latents(z[lo:hi], z.shape[0], group)
    res = build_codebook_device(z, dec, k=cfg["k"], sym="union", K=cfg["K"], init="kpp", seed=42, batch_size=512,
                                timers=timers, group=group)
    t0 = time.perf_counter()
    G = res["W_lcc"]
    src = torch.from_numpy(res["medoids"].astype(np.int32)).to(z.device)
    prof = {}

    def solve(s0, s1):
        if s1 <= s0:
            return (torch.full((G.n,), float("inf"), device=z.device), torch.zeros(G.n, dtype=torch.int32, device=z.device))
        _, _, dmin_, arg_, _ = sssp_multi_device(G, src[s0:s1].contiguous(), want_D=False, want_min=True)
        ms_, launches_ = np.zeros(1, np.float64), np.zeros(1, np.int32)
        layout = _lib.load().geo_sssp_last_profile(ms_.ctypes.data, launches_.ctypes.data)
        prof["ms"], prof["launches"], prof["sources"] = float(ms_[0]), int(launches_[0]), s1 - s0
        prof["kernel"] = ("push_sweep_kernel" if layout >= 4000 else "sweep_chunk32u_kernel" if layout >= 2000
                          else "sweep_chunk16_kernel" if layout >= 1000
                          else f"sweep_multi_kernel<{layout}>")
        return dmin_, arg_

    dmin, arg = sharded_assign(len(res["medoids"]), solve, group)
    ms, launches = np.array([prof.get("ms", 0.0)]), np.array([prof.get("launches", 0)])
    res["sharded"]["assign"] = world > 1
    res["sources_this_rank"] = prof.get("sources", 0)
    res["sweep_kernel"] = prof.get("kernel", "sweep_multi_kernel")
    if timers is not None:
        torch.cuda.synchronize(z.device)
        timers["assign_sweep"] = timers.get("assign_sweep", 0.0) + time.perf_counter() - t0
    res["assign_batched"] = arg
    return res, (float(ms[0]), int(launches[0]))


def cpu_baseline(res, z, dec, cfg, mode):
    """The oracle (CPU restatement, validated against the reference in the build container) timed on this host.
    mode "bounded" (default for c1 / c2; about 20-30 s of CPU work): kNN over all rows, the JVP over the first 1/8 of the
        BatchNorm chunks (x 8), and the k-medoids stage as the SINGLE-PASS chain -- K heap-Dijkstra solves with the running
        (min, first argmin), bit-equal to the reference's driver (tests/test_oracle_golden.py) -- whose solve time is scaled to
        the reference's 3K-1 solves; the port's medoids / QE are still compared with the GPU's;
    mode "full" (GEO_BENCH_CPU_FULL=1): every stage in full with the reference's solve count (3K-1 solves, ~75 s at c2);
    mode "sample" (large workloads): a bounded sample of every stage, scaled, and labelled as such."""
    from oracle import _clib, kmedoids as ok, metric as om, sssp as osp
    import ctypes
    n, d, K = cfg["n"], cfg["d"], cfg["K"]
    cores = host_cores()
    torch.set_num_threads(cores)
    zh = np.ascontiguousarray(z.cpu().numpy())
    lib = _clib.lib()
    rows = n if mode != "sample" else min(n, 6000)
    idx = np.empty((rows, cfg["k"] + 1), np.int64)
    d2 = np.empty((rows, cfg["k"] + 1), np.float64)
    t0 = time.perf_counter()
    lib.oracle_knn(ctypes.c_void_p(zh.ctypes.data), n, d, cfg["k"] + 1, 1 if d > 15 else 0, 0, rows,
                   ctypes.c_void_p(idx.ctypes.data), ctypes.c_void_p(d2.ctypes.data))
    t_knn = (time.perf_counter() - t0) * n / rows
    src, dst = (t.cpu().numpy() for t in res["edges"])
    E = len(src)
    e_s = E if mode == "full" else (min(E, 8192) if mode == "sample" else min(E, 512 * max(1, (E // 512) // 8)))
    sd = {k_: v.detach().cpu() for k_, v in dec.state_dict().items()}
    t0 = time.perf_counter()
    om.edge_lengths(sd, "batch", cfg["size"], zh[src[:e_s]], zh[dst[:e_s]], batch_size=512, training=True)
    t_jvp = (time.perf_counter() - t0) * E / e_s
    W = res["W_lcc"].to_scipy()
    n_solves = 3 * K - 1
    if mode == "full":
        t0 = time.perf_counter()
        med, assign, qe = ok.fit_kmedoids_optimized(W, K=K, init="kpp", seed=42)      # the reference's three stages
        t_kmed = time.perf_counter() - t0
        agree = bool(np.array_equal(med, res["medoids"]) and qe == res["qe"])
        sample = (f"full: oracle port on host, kNN {n} rows (OpenMP, {cores} threads), JVP {E} edges (torch CPU, "
                  f"{cores} threads), k-medoids {n_solves} heap-Dijkstra solves (1 thread, as scipy in the reference); "
                  f"port medoids/QE equal the GPU's: {agree}")
    elif mode == "bounded":
        t0 = time.perf_counter()
        med, assign, qe = ok.fit_kmedoids_single_pass(W, K=K, seed=42)                # K solves, same medoids / codes / QE
        t_one = time.perf_counter() - t0
        t_kmed = t_one * n_solves / K
        agree = bool(np.array_equal(med, res["medoids"]) and qe == res["qe"])
        sample = (f"bounded: oracle port on host, kNN all {n} rows (OpenMP, {cores} threads); JVP {e_s}/{E} edges = the first "
                  f"{e_s // 512} BatchNorm chunks x{E / e_s:.1f} (torch CPU, {cores} threads); k-medoids as the single-pass chain, "
                  f"{K} heap-Dijkstra solves + draws in {t_one:.1f} s (1 thread, as scipy in the reference) x{n_solves / K:.2f} for "
                  f"the reference's 3K-1 = {n_solves} solves; port medoids/QE equal the GPU's: {agree}; "
                  f"GEO_BENCH_CPU_FULL=1 runs all {n_solves} solves and every edge")
    else:
        n_src = 24 if n <= 200000 else 6            # (one heap-Dijkstra solve over 1 M nodes takes ~2.5 s)
        t0 = time.perf_counter()
        osp.dijkstra_multi_source(W, res["medoids"][:n_src])
        t_kmed = (time.perf_counter() - t0) / min(n_src, len(res["medoids"])) * n_solves
        sample = (f"EXTRAPOLATED from a sample: kNN {rows}/{n} query rows x{n / rows:.0f}, JVP {e_s}/{E} edges "
                  f"x{E / e_s:.0f} (torch CPU, {cores} threads), Dijkstra {n_src} sources x{n_solves / n_src:.0f} "
                  f"(reference runs 3K-1 = {n_solves} single-thread solves)")
    total = t_knn + t_jvp + t_kmed
    return {"value": n / total, "unit": "latents/s", "cores": cores, "kind": "port", "cpu_model": cpu_model(),
            "host_logical_cpus": os.cpu_count(), "affinity_mask_cpus": len(os.sched_getaffinity(0)),
            "capped_at": int(os.environ.get("GEO_BENCH_CPU_THREADS", str(DEFAULT_CPU_CAP))),
            "cores_note": "kNN (OpenMP) and JVP (torch CPU) use `cores` threads: the GPU box's CPU share for a 1-GPU job is 16 threads "
                          "although its affinity mask lists the whole host (more threads oversubscribe the share); the Dijkstra solves "
                          "are single-threaded as scipy's are in the reference", "sample": sample,
            "stages_s": {"knn": round(t_knn, 3), "jvp": round(t_jvp, 3), "kmedoids": round(t_kmed, 3)}}


def prior_bench(args, rank, world, dev, json_fd):
    """Secondary line (--workload c5prior): training throughput of the code prior at BASELINE config 5's shape -- the
    reference's configs/cifar10/spatial/geodesic/transformer.yaml model (4 layers, 256 dims, 4 heads, 512 tokens, dropout 0.1),
    batch 256 of 15-token sequences, AdamW -- on synthetic codes.  One step = forward + backward + (all-reduce) + optimiser.
    Timed twice: the MI355X-native step (HIP-graph replay, fused attention kernels, one-launch AdamW) and the same step on
    stock torch ops, both in this process."""
    import torch.nn.functional as F
    from vqvae_amd.parallel import block_range
    from vqvae_amd.prior.native import ArenaAdamW, GraphedStep
    from vqvae_amd.prior.transformer import Transformer
    K, B, T = 512, 256, 16
    torch.manual_seed(0)
    cfg = dict(num_classes=10, num_tokens=K, embed_dim=256, n_layers=4, n_head=4, max_seq_len=T, dropout=0.1)
    tokens = torch.randint(0, K, (50000, T), device=dev)
    labels_all = torch.randint(0, 10, (50000,), device=dev)
    lo, hi = block_range(B, rank, world)

    def run(native):
        torch.manual_seed(1)
        model = Transformer(**cfg).to(dev).train()
        model.fused_attention = native
        model.arena.grad = torch.zeros_like(model.arena)
        opt = ArenaAdamW(model.arena, 3e-4, 0.01) if native else torch.optim.AdamW(model.parameters(), lr=3e-4, weight_decay=0.01)

        def loss_fn(x, y, lab):
            logits = model(x, y=lab)
            return F.cross_entropy(logits.reshape(-1, K), y.reshape(-1), reduction="sum") / (B * (T - 1))

        graphed = None
        g = torch.Generator(device="cpu").manual_seed(2)

        def step():
            nonlocal graphed
            pick = torch.randint(0, 50000, (B,), generator=g)[lo:hi].to(dev)
            rows = tokens[pick]
            x, y, lab = rows[:, :-1], rows[:, 1:], labels_all[pick]
            if native:
                if graphed is None:
                    graphed = GraphedStep(model, loss_fn, x.contiguous(), y.contiguous(), lab)
                loss = graphed.run(x, y, lab)
            else:
                model.arena.grad.zero_()
                loss = loss_fn(x, y, lab)
                loss.backward()
            if world > 1:
                import torch.distributed as dist
                dist.all_reduce(model.arena.grad)
            opt.step()
            return loss

        for _ in range(max(args.warmup, 3)):
            step()
        torch.cuda.synchronize(dev)
        if world > 1:
            import torch.distributed as dist
            dist.barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            last = step()
        torch.cuda.synchronize(dev)
        if world > 1:
            import torch.distributed as dist
            dist.barrier()
        dt = time.perf_counter() - t0
        if world > 1:
            import torch.distributed as dist
            t = torch.tensor([dt], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        return dt, float(last)

    with contextlib.redirect_stdout(sys.stderr):
        dt_native, loss_native = run(True)
        dt_eager, loss_eager = run(False)
    tok = B * (T - 1) * args.steps
    out = {"metric": "prior training tokens/sec (BASELINE config 5's second stage; secondary to the codebook metric)",
           "value": tok / dt_native, "unit": "tokens/s", "n_gpus": world, "steps": args.steps, "warmup": max(args.warmup, 3),
           "ms_per_step": dt_native / args.steps * 1e3, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
           "dtype": "f32", "data": "synthetic",
           "config": {"workload": "c5prior: Transformer 4 layers x 256 dims x 4 heads, 512 tokens, 15-token sequences, global batch 256, "
                                  "dropout 0.1, AdamW lr 3e-4 wd 0.01", "parallelism": f"{world} rank(s), batch split, one flat all-reduce of 13 MB"},
           "native": {"what": "HIP-graph forward+backward, fused attention (csrc/prior.hip), one-launch AdamW", "ms_per_step": dt_native / args.steps * 1e3,
                      "last_loss": loss_native},
           "eager_torch": {"what": "same model and step on stock torch ops (the reference's method)", "ms_per_step": dt_eager / args.steps * 1e3,
                           "tokens_per_s": tok / dt_eager, "last_loss": loss_eager},
           "speedup_over_eager": dt_eager / dt_native}
    if rank == 0:
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(out) + "\n").encode())


def pam_bench(args, dev, json_fd):
    """Secondary line (--workload c2pam): the DENSE form of north_star's "APSP + K-medoids sweep" at configs[1]'s size --
    the all-pairs geodesic matrix of the 60 000-latent graph (14.4 GB, resident) and PAM swap evaluations over it.  An
    extension (the reference has neither): parity is against oracle/kmedoids.py's restatements in tests/.
    One step = one swap evaluation (all 512 x 59 488 exchanges): it reads the matrix exactly once, so the roofline is
    n^2 * 4 bytes / kernel time against 8 TB/s."""
    from vqvae_amd import _lib
    from vqvae_amd.geo.geo_shortest_paths import all_pairs_geodesic_device
    from vqvae_amd.geo.kmeans_optimized import fit_kmedoids_optimized, pam_swap_pass_device
    from vqvae_amd.geo.knn_graph_optimized import knn_graph_device
    n, d, k, K = 60000, 16, 20, 512
    z = torch.from_numpy(np.random.RandomState(0).randn(n, d).astype(np.float32)).to(dev)
    with contextlib.redirect_stdout(sys.stderr):
        G, _, _ = knn_graph_device(z, k, mode="distance", sym="union")
        med, _, _ = fit_kmedoids_optimized(G, K=K, init="kpp", seed=42)
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        D = all_pairs_geodesic_device(G)
        torch.cuda.synchronize(dev)
        t_apsp = time.perf_counter() - t0
        m = torch.from_numpy(med.astype(np.int32)).to(dev)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        kernel_ms = []
        orig = _lib.load().geo_pam_swap_deltas

        def timed(*a):                                   # HIP events on the stream the kernel is launched on
            e0.record()
            rc = orig(*a)
            e1.record()
            e1.synchronize()
            kernel_ms.append(e0.elapsed_time(e1))
            return rc

        for _ in range(args.warmup):
            pam_swap_pass_device(D, m, 2)
        torch.cuda.synchronize(dev)
        _lib.load().geo_pam_swap_deltas = timed
        try:
            t0 = time.perf_counter()
            for _ in range(args.steps):
                delta, i, x, total = pam_swap_pass_device(D, m, 2)
            torch.cuda.synchronize(dev)
            dt = time.perf_counter() - t0
        finally:
            _lib.load().geo_pam_swap_deltas = orig
    bytes_per_pass = float(n) * n * 4
    avg_kernel = sum(kernel_ms) / len(kernel_ms)
    achieved = bytes_per_pass / (avg_kernel * 1e-3) / 1e9
    out = {"metric": "PAM swap evaluations/sec over the dense 60 000 x 60 000 geodesic matrix (extension; secondary line)",
           "value": args.steps / dt, "unit": "swap passes/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup,
           "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f64 sums of f32 distances",
           "data": "synthetic", "config": {"workload": f"c2pam: N={n} d={d} k={k} K={K}, Euclidean kNN graph, power 2", "all_pairs_fill_s": t_apsp,
                                           "matrix_GB": bytes_per_pass / 1e9},
           "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": None,
                        "kernel": "pam_swap_kernel", "launches_per_step": 1, "avg_launch_ms": avg_kernel,
                        "algorithmic_bytes_per_launch": bytes_per_pass},
           "best_swap": {"delta": delta, "medoid_position": i, "candidate": x, "total_cost": total}}
    sys.stdout.flush()
    os.write(json_fd, (json.dumps(out) + "\n").encode())


def free_port() -> int:
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def launch_ranks(args) -> int:
    """--gpus N without a launcher: start N rank processes under torch.distributed.run (this process has not
    touched the GPU and never will) and relay their output; returns the exit code."""
    import subprocess
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()), os.path.abspath(__file__),
           "--gpus", str(args.gpus), "--steps", str(args.steps), "--warmup", str(args.warmup),
           "--workload", args.workload, "--pipeline", str(args.pipeline)] + (["--no-cpu-baseline"] if args.no_cpu_baseline else []) + \
          (["--replicas"] if args.replicas else [])
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    return subprocess.call(cmd, env=env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="c2", choices=sorted(WORKLOADS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--pipeline", type=int, default=int(os.environ.get("GEO_BENCH_PIPELINE", "0")),
                    help="independent builds in flight on one GPU, each on its own HIP stream and host thread (1 = one after the other; "
                         "0 = as many as fit: every build in flight owns its workspaces -- ~14 GB at the c2 size -- so the number is "
                         "taken from the memory one warmed-up build reserved, at most 8 on one GPU and 4 per rank of a multi-GPU run)")
    ap.add_argument("--replicas", action="store_true", default=os.environ.get("GEO_BENCH_REPLICAS", "0") == "1",
                    help="N > 1: every GPU builds its own codebooks (weak scaling, no data-path collective) instead of sharding each build")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        n_dev = torch.cuda.device_count()                 # counting devices does not initialise the GPU
        if n_dev < args.gpus and os.environ.get("GEO_BENCH_BACKEND", "nccl") == "nccl":
            log(f"bench.py: --gpus {args.gpus} but {n_dev} GPU(s) visible (set GEO_BENCH_BACKEND=gloo to rehearse the "
                f"{args.gpus}-rank path on fewer devices)")
            sys.exit(2)
        sys.exit(launch_ranks(args))

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    # stdout carries exactly ONE line, the JSON of rank 0: whatever libraries write to file descriptor 1 meanwhile (gloo
    # announces its connections there) goes to stderr
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    if world != args.gpus:
        log(f"bench.py: --gpus {args.gpus} but the launcher started {world} rank(s); reporting n_gpus={world}")
    assert torch.cuda.is_available(), "bench.py needs an MI355X"
    dev = torch.device("cuda", local % torch.cuda.device_count())
    torch.cuda.set_device(dev)
    backend = "none"
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = os.environ.get("GEO_BENCH_BACKEND", "nccl")      # "gloo" lets ranks rehearse on fewer GPUs
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    def barrier():
        if world > 1:
            import torch.distributed as dist
            dist.barrier()

    if args.workload == "c2pam":
        assert world == 1, "c2pam is a one-GPU line"
        pam_bench(args, dev, json_fd)
        return
    if args.workload == "c5prior":
        prior_bench(args, rank, world, dev, json_fd)
        if world > 1:
            import torch.distributed as dist
            dist.destroy_process_group()
        return
    z, dec, cfg = make_inputs(args.workload, dev)
    # Builds are independent of each other, and 40 % of one build is the k-means++ chain on ONE compute unit: with
    # --pipeline P (default: as many as fit in 80 % of the free HBM, at most 8; 8 at the c2 size) P builds are in flight, each driven
    # by its own host thread on its own HIP stream with its own workspace and decoder copy, so one build's chain runs beside the
    # next builds' kNN / JVP on the other CUs (measured on one box, c2: 31.5 ms per build at depth 4 and 5, 29.1-29.6 at 8, 29.5 at 10).  Every
    # step still does all of its work and is checked; `ms_per_step` is elapsed / steps (throughput), the latency of a single
    # build is reported beside it.  Ranks of a multi-GPU run issue collectives in program order: no pipelining there.
    import copy
    import threading
    replicas = world > 1 and args.replicas          # every rank on its own: builds are independent objects
    shard_world = 1 if replicas else world          # ranks ONE build is sharded over
    auto_depth = args.pipeline <= 0
    depth = max(1, args.pipeline)               # builds in flight per rank (sharded builds: collectives issued in ticket order); auto: set below
    solo = None
    if replicas:                                    # a process group of this rank alone: the sharding helpers then see world 1
        import torch.distributed as dist
        solo = [dist.new_group([r]) for r in range(world)][rank]
    if auto_depth or depth > 1:
        slots = [(torch.cuda.Stream(device=dev), copy.deepcopy(dec))]           # the others follow once the depth is known
    else:
        slots = [(torch.cuda.current_stream(dev), dec)]
    timers, prof, res = {}, (0.0, 0), None
    all_same = []

    from vqvae_amd.pipeline import run_pipelined
    streams = []                                     # (filled once every slot exists)

    from vqvae_amd.parallel import CollectiveOrder, OrderedGroup
    order = [None]                                   # sharded builds in flight: one ticket order per timed region

    def one_step(i, slot):
        grp = solo
        if order[0] is not None:
            grp = OrderedGroup(order[0], i)
        try:
            r, pr = hot_path_step(z, slots[slot][1], cfg, None, rank if not replicas else 0, shard_world, grp)
        except BaseException as e:                   # noqa: BLE001 -- the other builds must not wait for this one's tickets
            if order[0] is not None:
                order[0].abort(e)
            raise
        finally:
            if order[0] is not None:
                order[0].finish(i)
        ok = bool((r["assign_batched"].cpu().numpy() == r["assign_flat"][r["mask_lcc"]]).all())   # (syncs this stream only)
        return i, float(r["qe"]), pr, ok

    def timed_region(d):
        """EXACTLY args.steps builds with d of them in flight per rank, bracketed by barrier + synchronize; max over ranks."""
        barrier()
        torch.cuda.synchronize(dev)
        order[0] = CollectiveOrder(args.steps, d) if (shard_world > 1 and d > 1) else None
        t0 = time.perf_counter()
        if d > 1:
            done = run_pipelined(one_step, args.steps, d, dev, streams)
        else:
            with torch.cuda.stream(slots[0][0]):                  # (the stream slot 0 was warmed on)
                done = [one_step(i, 0) for i in range(args.steps)]
        torch.cuda.synchronize(dev)
        barrier()
        el = time.perf_counter() - t0
        order[0] = None
        assert len(done) == args.steps
        assert len({q for _, q, _, _ in done}) == 1, "builds of the timed region disagree"      # every build returned the same QE
        if world > 1:
            import torch.distributed as dist
            t = torch.tensor([el], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            el = float(t.item())
        region_prof[:] = [pr for _, _, pr, _ in done]            # per-build HIP-event sweep time measured INSIDE the timed region
        return el, [ok for _, _, _, ok in done]

    fallback_note = None
    region_prof = []
    depth_rule = "given on the command line"
    with contextlib.redirect_stdout(sys.stderr):
        sl = -1
        while True:                                               # every slot warms up its own stream and workspace
            sl += 1
            if sl >= len(slots):
                if sl >= depth:
                    break
                slots.append((torch.cuda.Stream(device=dev), copy.deepcopy(dec)))
            reserved0 = torch.cuda.memory_reserved(dev)
            with torch.cuda.stream(slots[sl][0]):
                for _ in range(max(1, args.warmup) if (auto_depth and sl == 0) else args.warmup):
                    hot_path_step(z, slots[sl][1], cfg, None, rank if not replicas else 0, shard_world, solo)
            if auto_depth and sl == 0:
                # one build's workspaces are reserved now: as many builds in flight as fit in 80 % of what is free, counting one
                # more set for the instrumented build after the timed region; at most 8 (4 per rank when builds are sharded)
                torch.cuda.synchronize(dev)
                per_slot = max(1, torch.cuda.memory_reserved(dev) - reserved0)
                free_now = torch.cuda.mem_get_info(dev)[0]
                cap = 8 if shard_world == 1 else 4
                depth = int(min(cap, max(1, 1 + (0.8 * free_now - per_slot) // per_slot)))
                if world > 1:                                     # every rank must run the same number of builds in flight
                    import torch.distributed as dist
                    t = torch.tensor([depth], dtype=torch.int64, device=dev)
                    dist.all_reduce(t, op=dist.ReduceOp.MIN)
                    depth = int(t.item())
                depth_rule = (f"auto: one warmed-up build reserved {per_slot / 2**30:.1f} GiB, {free_now / 2**30:.1f} GiB were free "
                              f"-> {depth} in flight (80 % of the free memory, one set kept for the instrumented build, cap {cap})")
        streams[:] = [st for st, _ in slots]
        guarded = shard_world > 1 and depth > 1
        depth_tried = depth
        if guarded:
            # Sharded builds in flight need every rank to issue its collectives in one order (parallel.CollectiveOrder).  That
            # order is by construction, but this repository's own runs never had more than one GPU: a region with one build
            # after the other is timed first, the pipelined region runs under a watchdog, and the line reports the faster of
            # the two -- or the plain one, if the pipelined region does not come back.
            elapsed_plain, same_plain = timed_region(1)
            prof_plain = list(region_prof)
        else:
            elapsed, all_same = timed_region(depth)
        # one more build, alone and instrumented (NOT part of a timed region): stage times and single-build latency
        if depth > 1:                                             # this stream's workspace has not been used yet: warm it
            hot_path_step(z, dec, cfg, None, rank if not replicas else 0, shard_world, solo)
        torch.cuda.synchronize(dev)
        t1 = time.perf_counter()
        res, prof = hot_path_step(z, dec, cfg, timers, rank if not replicas else 0, shard_world, solo)
        torch.cuda.synchronize(dev)
        latency_ms = (time.perf_counter() - t1) * 1e3
        # beside the K-source sweep (the roofline kernel this bench is about): the assignment alone, from ONE label-carrying
        # solve (geo_sssp_nearest_source) -- what assign_points_to_medoids runs; must give the same rows
        from vqvae_amd.geo.geo_shortest_paths import nearest_source_device
        srcs_ = torch.from_numpy(res["medoids"].astype(np.int32)).to(dev)
        nearest_source_device(res["W_lcc"], srcs_)
        torch.cuda.synchronize(dev)
        t2 = time.perf_counter()
        _, a1_, sw1_ = nearest_source_device(res["W_lcc"], srcs_)
        torch.cuda.synchronize(dev)
        one_solve = {"ms": (time.perf_counter() - t2) * 1e3, "sweeps": sw1_,
                     "equals_k_source_assignment": bool(torch.equal(a1_, res["assign_batched"]))}
        if guarded:
            box = {}

            def pipelined_region():
                try:
                    torch.cuda.set_device(dev)
                    box["out"] = timed_region(depth)
                except BaseException as e:                        # noqa: BLE001
                    box["err"] = e

            limit = max(120.0, 40.0 * elapsed_plain)
            th = threading.Thread(target=pipelined_region, name="geo-pipelined-region", daemon=True)
            th.start()
            th.join(limit)
            if "out" in box and box["out"][0] < elapsed_plain:
                elapsed, all_same = box["out"]
            else:
                if "out" in box:
                    fallback_note = f"{depth} builds in flight were not faster ({box['out'][0] / args.steps * 1e3:.1f} ms per build)"
                elif "err" in box:
                    fallback_note = f"the region with {depth} builds in flight failed: {box['err']!r}"
                else:
                    fallback_note = f"the region with {depth} builds in flight did not come back within {limit:.0f} s"
                log("bench.py: " + fallback_note + "; reporting the region with one build after the other")
                elapsed, all_same, depth = elapsed_plain, same_plain, 1
                region_prof[:] = prof_plain
            hung = th.is_alive()
        else:
            hung = False

    # the batched assignment stage must reproduce the fused chain's assignment
    same = bool((res["assign_batched"].cpu().numpy() == res["assign_flat"][res["mask_lcc"]]).all()) and all(all_same)
    G = res["W_lcc"]
    n, nnz, K = G.n, G.nnz, len(res["medoids"])
    sweep_ms, launches = prof
    algo_bytes = res["sources_this_rank"] * (16.0 * nnz + 16.0 * n)    # B_sssp x sources solved on this rank
    achieved = algo_bytes / (sweep_ms * 1e-3) / 1e9 if sweep_ms > 0 else 0.0
    # PMC-derived bytes per launch are NOT measured in this run (counters need their own rocprofv3 --pmc passes):
    # the committed summary of the stated PMC run is quoted, for the same kernel and workload only
    traffic, traffic_source = None, None
    tpath = os.path.join(ROOT, "profiles", "traffic_latest.json")
    if os.path.exists(tpath) and args.workload == "c2" and world == 1:
        with open(tpath) as f:
            tj = json.load(f)
        if tj.get("kernel") == res["sweep_kernel"]:
            traffic = tj.get("hbm_bytes_per_launch")
            traffic_source = f"profiles/traffic_latest.json (separate rocprofv3 --pmc passes, run {tj.get('tag', '?')}); not measured in this run"
    # what the DISPATCHED kernel must move per (edge, source) and per (node, source) -- beside SURVEY 8(d)'s 16 + 16 model:
    #   32-bit fixed point: 4 B distance gather + 8 B of CSR entry (column, weight units) shared by 32 sources; 4 B read + 4 B write
    #   fp64 kernels:       8 B distance gather + 8 B of CSR entry shared by 16 sources;                       8 B read + 8 B write
    per_edge, per_node = (4.0 + 8.0 / 32, 8.0) if res["sweep_kernel"] == "sweep_chunk32u_kernel" else (8.0 + 8.0 / 16, 16.0)
    kernel_bytes = res["sources_this_rank"] * (per_edge * nnz + per_node * n)
    ms_per_step = elapsed / args.steps * 1e3
    # the same kernel INSIDE the timed region (other builds' kNN / JVP / chain kernels share the chip while it runs): mean of the
    # per-build HIP-event sweep times that every timed build returned
    fl_ms = [m for m, l in region_prof if l > 0]
    in_flight_ms = sum(fl_ms) / len(fl_ms) if fl_ms else 0.0
    in_flight_launches = (sum(l for _, l in region_prof) / len(region_prof)) if region_prof else 0
    stages_ms = {k_: v * 1e3 for k_, v in timers.items()}          # of the one instrumented build after the timed region
    sharded = res.get("sharded", {})
    if replicas:
        parallelism = f"{world} independent replicas over {backend} (no data-path collective), {depth} builds in flight each"
    elif world > 1:
        parts = [name for name, on in (("kNN query rows", sharded.get("knn")), ("JVP chunks", sharded.get("jvp")),
                                       ("assignment sources", sharded.get("assign"))) if on]
        parallelism = (f"{world} ranks over {backend}: " + (" + ".join(parts) + " sharded (all-gather merges)" if parts else "nothing sharded")
                       + f"; k-means++ chain replicated on every rank ({stages_ms.get('kmedoids', 0.0):.1f} ms of a build do not shard)"
                       + (f"; {depth} sharded builds in flight per rank, collectives issued in ticket order on one communicator" if depth > 1 else ""))
    else:
        parallelism = "1 gpu"
    out = {
        "metric": "latents/sec through geodesic kNN+APSP+K-medoids", "value": cfg["n"] * (world if replicas else 1) / (elapsed / args.steps),
        "unit": "latents/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak" if replicas else "strong", "vs_baseline": None,
        "dtype": "f64-exact(u32 fixed point)" if res["sweep_kernel"] == "sweep_chunk32u_kernel" else "f64", "data": "synthetic",
        "config": {"workload": f"{args.workload}: N={cfg['n']} latents d={cfg['d']} k={cfg['k']} K={cfg['K']} "
                               f"{cfg['size']}px decoder BN-train batch 512 sym=union init=kpp seed=42",
                   "graph": {"nodes": n, "nnz": nnz, "edges_reweighted": res["n_edges"]},
                   "parallelism": parallelism, "pipeline_depth": depth, "pipeline_depth_rule": depth_rule,
                   "pipeline": (f"{depth} independent builds in flight, one host thread + HIP stream + workspace each" if depth > 1
                                else "one build after the other")},
        "latency_ms_single_build": latency_ms,
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_source,
                     "frac_measured_on": "one instrumented build alone, after the timed region (solo); frac_in_flight = the same kernel's "
                                         "mean over the builds OF the timed region, other builds' kernels sharing the chip",
                     "frac_in_flight": (algo_bytes / (in_flight_ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if in_flight_ms > 0 else None,
                     "in_flight": {"builds": len(region_prof), "pipeline_depth": depth, "mean_sweep_ms_per_build": in_flight_ms,
                                   "mean_launches_per_build": in_flight_launches,
                                   "avg_launch_ms": in_flight_ms / in_flight_launches if in_flight_launches else None,
                                   "min_sweep_ms": min(fl_ms) if fl_ms else None, "max_sweep_ms": max(fl_ms) if fl_ms else None},
                     "kernel": res["sweep_kernel"], "launches_per_step": launches,
                     "avg_launch_ms": sweep_ms / max(1, launches),
                     "algorithmic_bytes_per_launch": algo_bytes / max(1, launches),
                     # the kernel's own minimum traffic (ONE relaxation of every entry, see above) and what the run moved
                     # relative to it: label correcting evaluates a row several times, so `passes_over_minimum` > 1
                     "kernel_bytes": kernel_bytes, "frac_kernel_bytes": (kernel_bytes / (sweep_ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if sweep_ms > 0 else 0.0,
                     "frac_counters": (traffic * launches / (sweep_ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if (traffic and sweep_ms > 0) else None,
                     "passes_over_minimum": (traffic * launches / kernel_bytes) if (traffic and kernel_bytes) else None},
        "stages_ms": stages_ms,
        "parity_selfcheck": {"batched_assign_equals_fused": same, "qe": res["qe"], "one_solve_assignment": one_solve},
    }
    if guarded:
        out["config"]["regions_timed"] = {"one_build_after_the_other_ms_per_build": elapsed_plain / args.steps * 1e3,
                                          "builds_in_flight_tried": depth_tried, "reported": "in flight" if depth > 1 else "one after the other",
                                          "note": fallback_note}
    # compute-bound stages: kNN as 2*N^2*d flop vs the fp64 vector peak, JVP as 3.47 MFLOP/edge vs the
    # f32-MFMA peak; stage wall time (whole stage incl. graph assembly / BN statistics), this rank's share of the work
    share = 1.0 / shard_world
    if stages_ms.get("knn"):
        nn, dd = cfg["n"], cfg["d"]
        fl64 = 2.0 * nn ** 2 * dd * share
        t_knn = stages_ms["knn"] * 1e-3
        filtered = nn >= (40000 if dd <= 16 else 20000 if dd <= 32 else 16384)       # csrc/knn.hip: matrix-core filter + fp64 refinement
        if filtered:
            # what the matrix pipe is ISSUED: every (query, corpus) pair, 3 split products (hi*hi, hi*lo, lo*hi) of
            # v_mfma_f32_32x32x16_bf16 per 16 padded dimensions = 3 * 2 * N^2 * dp flop
            dp = ((dd + 15) // 16) * 16
            issued = 3.0 * 2.0 * nn ** 2 * dp * share
            out["roofline_knn"] = {"bound": "mfma-bf16", "achieved": issued / t_knn / 1e12, "peak": BF16_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                                   "frac": issued / t_knn / 1e12 / BF16_MFMA_PEAK_TFLOPS,
                                   "model": "issued bf16 flop of the filter scan (3 split MFMA products per pair and 16 dims) / kNN STAGE time "
                                            "(thresholds + scan + fp64 refinement + symmetrise + edge list); the exact fp64 work rides on the VALU beside it",
                                   "equivalent_fp64_tflops": fl64 / t_knn / 1e12,
                                   "equivalent_model": "SURVEY 8(d): 2*N^2*d flop / stage time -- the fp64 brute-force scan this stage REPLACES, "
                                                       f"not executed as such (fp64 vector peak {FP64_PEAK_TFLOPS} TFLOP/s); no fraction is formed from it"}
        else:
            out["roofline_knn"] = {"bound": "fp64-valu", "achieved": fl64 / t_knn / 1e12, "peak": FP64_PEAK_TFLOPS, "unit": "TFLOP/s",
                                   "frac": fl64 / t_knn / 1e12 / FP64_PEAK_TFLOPS,
                                   "model": "2*N^2*d fp64 flop (exact scan: this size runs no filter) / kNN stage time (search + symmetrise + edge list)"}
    if stages_ms.get("jvp"):
        per_edge_fl = jvp_flop_per_edge(cfg["d"], cfg["cout"], cfg["size"])
        c0, c1, c2 = 256, 128, 64
        pairs3 = 64 if int(cfg["size"]) == 28 else 196
        mfma_mac = 36 * c1 * c2 + pairs3 * c2 * cfg["cout"]           # ConvT2 + ConvT3: the layers that run on the matrix pipe
        t_jvp = stages_ms["jvp"] * 1e-3
        issued = 6.0 * 8.0 * mfma_mac * res["n_edges"] * share         # 2 ends x (primal + tangent) x 2 flop x 6 bf16 products per f32 product
        eff = per_edge_fl * res["n_edges"] * share / t_jvp / 1e12
        out["roofline_jvp"] = {"bound": "mfma-bf16", "achieved": issued / t_jvp / 1e12, "peak": BF16_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                               "frac": issued / t_jvp / 1e12 / BF16_MFMA_PEAK_TFLOPS,
                               "model": "issued bf16 flop: 6 MFMA products (exact 3-way bf16 split) per f32 MAC of ConvT2 + ConvT3, useful taps only "
                                        "/ JVP STAGE time (first layer, BatchNorm statistics, ConvT2, head); MFMA-busy cycles are in profiles/",
                               "flop_per_edge": per_edge_fl, "equivalent_f32_tflops": eff,
                               "equivalent_model": f"{per_edge_fl / 1e6:.2f} MFLOP/edge of f32 arithmetic from the decoder shape (d={cfg['d']}, {cfg['size']}-px, "
                                                   f"{cfg['cout']} ch) / stage time -- what an f32 implementation would execute (f32-MFMA peak "
                                                   f"{F32_MFMA_PEAK_TFLOPS} TFLOP/s); no fraction is formed from it"}
    # multi-GPU accounting: bytes each rank RECEIVES per step in the all-gather merges, and the part of the step
    # that does not shard (the k-means++ chain + component labelling are replicated) -- the first scaling line explains itself
    E = res["n_edges"]
    out["comm_bytes_per_step"] = {
        "latents_all_gather": int(cfg["n"] * cfg["d"] * 4 * (shard_world - 1) / shard_world) if shard_world > 1 else 0,
        "knn_lists_all_gather": int(cfg["n"] * (cfg["k"] + 1) * 4 * (shard_world - 1) / shard_world) if shard_world > 1 else 0,
        "edge_lengths_all_gather": int(E * 4 * (shard_world - 1) / shard_world) if shard_world > 1 else 0,
        "assign_min_argmin_all_gather": int(n * 8 * (shard_world - 1)) if shard_world > 1 else 0}
    serial_ms = stages_ms.get("kmedoids", 0.0) + stages_ms.get("lcc", 0.0)
    out["serial_fraction"] = {"ms_not_sharded": serial_ms, "of_single_build": serial_ms / latency_ms if latency_ms > 0 else None,
                              "amdahl_limit_8_gpus_single_build": (latency_ms / (serial_ms + (latency_ms - serial_ms) / 8.0)) if world == 1 and latency_ms > 0 else None}
    if hung:
        out["hang"] = True                            # the pipelined region never came back: the value above is the plain region's
    if rank == 0:
        if world == 1 and not args.no_cpu_baseline:
            mode = "sample"
            if args.workload in ("c1", "c2") and os.environ.get("GEO_BENCH_CPU_SAMPLE", "0") != "1":
                mode = "full" if os.environ.get("GEO_BENCH_CPU_FULL", "0") == "1" else "bounded"
            with contextlib.redirect_stdout(sys.stderr):
                out["cpu_baseline"] = cpu_baseline(res, z, dec, cfg, mode)
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(out) + "\n").encode())
    if hung:                                        # host threads parked inside the region that never came back: leave without joining
        sys.stderr.flush()                          # them, and say so in the exit code (the line above carries "hang": true)
        os._exit(3)
    if world > 1:
        import torch.distributed as dist
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

"""Riemannian codebook over VECTOR latents of the vanilla VAE, resident on the MI355X.

Drop-in for the reference's legacy builder (src/training/build_riemannian_codebook_legacy.py: `build_and_save(config)`
:169-291 and its `__main__`): same YAML keys and fall-backs, same four artefacts (knn_graph_euclidean.npz,
knn_graph_riemannian.npz, codebook.pt with medoid_indices / z_medoid / config / graph_stats / method, codes.npy with -1
outside the largest component) -- pinned by tests/golden/legacy_riemannian.npz, the reference's own output.

What the reference does, stated as the contract this file implements:
  1. Euclidean kNN graph of the latents (graph.k / metric / sym / mode), connectivity report, largest component;
  2. every STORED entry (i, j) of that graph -- both directions of an edge are stored -- gets the decoder pull-back length
     of its edge ("full"), or only a sample does ("subset": the entries are cut into five equal-probability bands of
     their Euclidean weight and at most max_edges // 5 entries are drawn from each band with numpy's GLOBAL generator,
     no replacement; untouched entries keep their Euclidean weight);
  3. the two directions of an edge are reconciled by the entry-wise maximum, non-finite lengths fall back to the
     Euclidean weight of the same entry;
  4. geodesic k-medoids (quantize.K / init / seed) on the re-weighted graph.

How it runs here: the graph never leaves HBM between steps 1 and 4.  The kNN graph, its components and the compaction to
the largest component are the `DeviceCSR` primitives of vqvae_amd.geo; an entry's row comes from one
`repeat_interleave` of the row pointer; lengths are computed on the device (HIP kernels for SpatialDecoder-shaped
decoders, autograd on the GPU for the Linear-first vanilla decoder) and scattered into the entry array; the maximum over
the two directions of an edge is a scatter-max / gather through the entry -> undirected-edge map that `upper_edges_device`
already provides.  The host sees only what the contract puts there: the band quantiles and the `np.random.choice` draws of
the subset mode (numpy's generator is host state by definition), the printed statistics, and the artefacts on disk.
"""
import argparse
import warnings
from dataclasses import dataclass
from pathlib import Path
from typing import Dict, Optional, Tuple

import numpy as np
import torch
from scipy import sparse

from .._device import DeviceCSR, device
from ..geo.kmeans_optimized import fit_kmedoids_optimized
from ..geo.knn_graph_optimized import (compact_device, connected_components_device, knn_graph_device, upper_edges_device)
from ..geo.riemannian_metric import edge_lengths_riemannian
from ..vae import decoder_from_vae_checkpoint

N_BANDS = 5                     # Euclidean-weight bands of the subset mode
_RUN_DIRS = {"mnist": "experiments/vae_mnist", "fashion": "experiments/vae_fashion", "cifar10": "experiments/vae_cifar10"}


@dataclass
class LegacyJob:
    """The reference's configuration dictionary, resolved (same keys, same fall-backs)."""
    latents: Path
    checkpoint: Path
    out_dir: Path
    vae_config: Dict
    k: int
    metric: str
    sym: str
    graph_mode: str
    reweight_mode: str
    max_edges: int
    batch_size: int
    K: int
    init: str
    seed: int

    @staticmethod
    def from_config(cfg: Dict) -> "LegacyJob":
        data = cfg.get("data") if isinstance(cfg.get("data"), dict) else {}

        def run_dir(default: str) -> str:
            name = str(data.get("dataset", default)).strip().lower()
            return _RUN_DIRS.get(name, _RUN_DIRS[default])

        model_cfg = cfg.get("model") if isinstance(cfg.get("model"), dict) else {}
        ckpt = cfg.get("checkpoint_path") or cfg.get("vae", {}).get("ckpt_path") or model_cfg.get("checkpoint_path")
        vae_config = cfg.get("vae_config") or cfg.get("model") or cfg.get("vae")
        if vae_config is None:
            raise ValueError("VAE configuration not found. Expected 'vae_config', 'model', or 'vae' key in config.")
        graph, riem, quant = cfg["graph"], cfg.get("riemannian", {}), cfg["quantize"]
        return LegacyJob(
            latents=Path(data.get("latents_path") or run_dir("mnist") + "/latents_train/z.pt"),
            checkpoint=Path(ckpt or run_dir("fashion") + "/checkpoints/best.pt"),
            out_dir=Path(cfg["out"]["dir"]), vae_config=vae_config,
            k=int(graph["k"]), metric=str(graph["metric"]), sym=str(graph["sym"]), graph_mode=str(graph["mode"]),
            reweight_mode=riem.get("mode", "subset"), max_edges=int(riem.get("max_edges", 5000)),
            batch_size=int(riem.get("batch_size", 512)),
            K=int(quant["K"]), init=str(quant["init"]), seed=int(quant["seed"]))


def read_latents(path: Path) -> torch.Tensor:
    """z.pt as written by the vanilla trainer: a tensor, or a dict holding one under "z"."""
    blob = torch.load(path, map_location="cpu")
    z = blob["z"] if isinstance(blob, dict) and "z" in blob else blob
    if not torch.is_tensor(z):
        raise ValueError("Expected dict with 'z' key or tensor")
    return z.float()


def read_decoder(path: Path, vae_config: Dict, dev: torch.device) -> torch.nn.Module:
    """The vanilla VAE's decoder in eval mode on `dev`; checkpoint = {"model_state_dict": ...} or a bare state dict."""
    blob = torch.load(path, map_location=dev)
    if not isinstance(blob, dict):
        raise ValueError("Expected checkpoint dict with model_state_dict")
    return decoder_from_vae_checkpoint(blob.get("model_state_dict", blob), **vae_config).to(dev).eval()


# ------------------------------------------------------------------------------------------------ device steps
def entry_rows(G: DeviceCSR) -> torch.Tensor:
    """Row of every stored entry (int64 [nnz]); entries are in row-major order, i.e. the order scipy's COO view has."""
    counts = (G.indptr[1:] - G.indptr[:-1]).long()
    return torch.repeat_interleave(torch.arange(G.n, device=G.indptr.device), counts)


def banded_entry_sample(weights: np.ndarray, max_edges: int) -> np.ndarray:
    """Entry indices of the subset mode.  Band b holds the entries whose weight lies between the b/5 and (b+1)/5
    quantiles, both ends included; each band contributes min(max_edges // 5, size) entries drawn without replacement
    from numpy's global generator, bands in ascending order (so that a seeded generator reproduces the reference's pick)."""
    cuts = np.quantile(weights, np.linspace(0.0, 1.0, N_BANDS + 1))
    picks = []
    for b in range(N_BANDS):
        members = np.flatnonzero((weights >= cuts[b]) & (weights <= cuts[b + 1]))
        take = min(max_edges // N_BANDS, members.size)
        if take > 0:
            picks.append(np.random.choice(members, size=take, replace=False))
    return np.concatenate(picks) if picks else np.empty(0, dtype=np.int64)


def reweight_graph_device(G: DeviceCSR, z: torch.Tensor, decoder: torch.nn.Module, mode: str = "subset",
                          max_edges: int = 5000, batch_size: int = 512) -> DeviceCSR:
    """Steps 2 and 3 on resident data: returns the re-weighted graph (same structure as G, new f32 data)."""
    dev = G.indptr.device
    rows, cols = entry_rows(G), G.indices.long()
    nnz = G.nnz
    print(f"Graph has {nnz} edges")
    if mode == "subset" and nnz > max_edges:
        chosen = torch.from_numpy(banded_entry_sample(G.data.cpu().numpy(), max_edges)).to(dev)
        print(f"Reweighting {chosen.numel()} edges (subset mode)")
    else:
        chosen = torch.arange(nnz, device=dev)
        print(f"Reweighting all {chosen.numel()} edges (full mode)")
    print(f"Computing Riemannian distances for {chosen.numel()} edges...")
    lengths = edge_lengths_riemannian(decoder, z[rows[chosen]], z[cols[chosen]], batch_size=batch_size).to(dev)
    data = G.data.clone()
    # an entry sitting exactly on a band boundary can be drawn twice: the later draw wins, as in a numpy fancy assignment
    order = torch.argsort(chosen, stable=True)
    last_of_run = torch.ones_like(order, dtype=torch.bool)
    last_of_run[:-1] = chosen[order][1:] != chosen[order][:-1]
    winners = order[last_of_run]
    data[chosen[winners]] = lengths[winners]
    # the two directions of an edge -> their maximum (structure is symmetric: both entries exist)
    _, _, entry_edge = upper_edges_device(G)
    edge = entry_edge.long()
    n_edges = int(edge.max()) + 1 if nnz else 0
    top = torch.full((n_edges,), float("-inf"), device=dev).scatter_reduce(0, edge, data, "amax", include_self=True)
    # (NaN propagates through amax like numpy's maximum; it is caught below)
    data = top[edge]
    bad = ~torch.isfinite(data)
    if bool(bad.any()):
        warnings.warn(f"Found {int(bad.sum())} non-finite Riemannian distances, keeping original Euclidean weights")
        data = torch.where(bad, G.data, data)
    ratio = float((lengths / G.data[chosen]).mean()) if chosen.numel() else float("nan")
    print(f"Riemannian reweighting complete. Edge weight ratio: mean={ratio:.3f}")
    return DeviceCSR(G.n, G.indptr, G.indices, data.contiguous())


def connectivity_report(G: DeviceCSR) -> Tuple[Dict, torch.Tensor]:
    """The reference's `analyze_graph_connectivity` dictionary and printout, and the mask of the largest component
    (first label on ties), from ONE component labelling of the resident graph."""
    n_comp, labels = connected_components_device(G)
    sizes = torch.bincount(labels.long(), minlength=max(n_comp, 1))
    largest = int(sizes.max()) if n_comp > 1 else G.n
    deg = torch.zeros(G.n, dtype=torch.float64, device=G.indptr.device).index_add_(0, entry_rows(G), G.data.double())
    stats = {"n_nodes": G.n, "n_edges": G.nnz, "n_components": n_comp, "largest_component_size": largest,
             "connectivity_ratio": largest / G.n if n_comp > 1 else 1.0,
             "avg_degree": np.float32(deg.mean().item()), "min_degree": np.float32(deg.min().item()),
             "max_degree": np.float32(deg.max().item())}
    print("Graph connectivity")
    print(f"nodes={G.n} edges={G.nnz} avg_deg={stats['avg_degree']:.1f}")
    print(f"components={n_comp} largest={largest} ({100 * stats['connectivity_ratio']:.1f}%)")
    if n_comp > 1:
        print("disconnected -> will use LCC")
        mask = labels == int(torch.argmax(sizes))              # first maximum, as np.argmax(bincount)
    else:
        mask = torch.ones(G.n, dtype=torch.bool, device=G.indptr.device)
    return stats, mask


# ------------------------------------------------------------------------------------------------ driver
def build_and_save(config: Dict, dev: Optional[torch.device] = None) -> Path:
    job = LegacyJob.from_config(config)
    job.out_dir.mkdir(parents=True, exist_ok=True)
    dev = dev or device()

    z_host = read_latents(job.latents)
    N, D = z_host.shape
    print(f"Loaded latents: N={N}, D={D}")
    print(f"Using device: {dev}")
    z = z_host.contiguous().to(dev)
    decoder = read_decoder(job.checkpoint, job.vae_config, dev)

    print(f"Building k-NN graph: k={job.k}, metric={job.metric}, sym={job.sym}")
    print(f"Building k-NN graph: N={N}, k={job.k}, method=hip")
    G, _, _ = knn_graph_device(z, job.k, mode=job.graph_mode, sym=job.sym, metric=job.metric)
    stats, mask = connectivity_report(G)
    n_lcc = int(mask.sum())
    if n_lcc < N:
        print(f"Using LCC: {n_lcc}/{N} nodes")
        G_lcc, _ = compact_device(G, mask, drop_zero=False)
        z_lcc = z[mask]
    else:
        G_lcc, z_lcc = G, z

    print(f"Re-weighting graph with Riemannian distances (mode={job.reweight_mode})")
    G_riem = reweight_graph_device(G_lcc, z_lcc, decoder, job.reweight_mode, job.max_edges, job.batch_size)
    sparse.save_npz(job.out_dir / "knn_graph_euclidean.npz", G_lcc.to_scipy())
    sparse.save_npz(job.out_dir / "knn_graph_riemannian.npz", G_riem.to_scipy())

    print(f"Running K-medoids on Riemannian graph: K={job.K}, init={job.init}")
    medoids, assign_lcc, qe = fit_kmedoids_optimized(G_riem, K=job.K, init=job.init, seed=job.seed)
    codes = assign_lcc
    if n_lcc < N:
        codes = np.full(N, -1, dtype=np.int32)
        codes[mask.cpu().numpy()] = assign_lcc
    torch.save({"medoid_indices": medoids.astype(np.int32),
                "z_medoid": z_lcc[torch.from_numpy(medoids).to(dev)].float().cpu(),
                "config": config, "graph_stats": stats, "method": "riemannian_geodesic"}, job.out_dir / "codebook.pt")
    np.save(job.out_dir / "codes.npy", codes)
    print(f"Riemannian quantization error: {qe:.3f}")
    print(f"Saved artifacts to: {job.out_dir}")
    return job.out_dir


if __name__ == "__main__":
    import yaml
    cli = argparse.ArgumentParser(description="Build Riemannian geodesic codebook")
    cli.add_argument("--config", type=str, default="configs/quantize.yaml", help="Configuration file path")
    with open(cli.parse_args().config, "r") as fh:
        print(f"Completed: {build_and_save(yaml.safe_load(fh))}")

"""Legacy Riemannian codebook builder for vector latents of the vanilla VAE -- drop-in for the reference's
src/training/build_riemannian_codebook_legacy.py (_reweight_graph_with_riemannian :67-166, build_and_save :169-291;
same YAML keys, same artefacts: knn_graph_euclidean.npz, knn_graph_riemannian.npz, codebook.pt, codes.npy).

Every numeric step is one of this package's src/geo replacements (kNN graph, components, decoder pull-back lengths,
geodesic k-medoids on the MI355X); this file is the reference's glue: Euclidean kNN graph -> largest component ->
re-weight all edges ("full") or a distance-stratified sample of max_edges of them ("subset", drawn with numpy's GLOBAL
generator like the reference: seed it with np.random.seed for reproducible subsets) -> W.maximum(W.T) -> k-medoids.
"""
import argparse
import warnings
from pathlib import Path
from typing import Dict

import numpy as np
import torch
from scipy import sparse

from ..geo.kmeans_optimized import fit_kmedoids_optimized
from ..geo.knn_graph_optimized import analyze_graph_connectivity, build_knn_graph_auto, largest_connected_component
from ..geo.riemannian_metric import edge_lengths_riemannian
from ..vae import VAE


def _load_latents(path: Path) -> torch.Tensor:
    obj = torch.load(path, map_location="cpu")
    if isinstance(obj, dict) and "z" in obj:
        return obj["z"].float()
    if torch.is_tensor(obj):
        return obj.float()
    raise ValueError("Expected dict with 'z' key or tensor")


def _load_vae_model(checkpoint_path: Path, vae_config: Dict, device: torch.device) -> VAE:
    checkpoint = torch.load(checkpoint_path, map_location=device)
    if not isinstance(checkpoint, dict):
        raise ValueError("Expected checkpoint dict with model_state_dict")
    model = VAE(**vae_config)
    model.load_state_dict(checkpoint.get("model_state_dict", checkpoint))
    return model.to(device).eval()


def _reweight_graph_with_riemannian(W: sparse.csr_matrix, z: np.ndarray, decoder: torch.nn.Module, mode: str = "subset",
                                    max_edges: int = 5000, batch_size: int = 512, device: torch.device = None):
    """Stored entries of W (in COO order) get the decoder pull-back length of their edge; "subset": only up to
    max_edges // 5 entries from each Euclidean-length quintile, the rest keep their Euclidean weight."""
    if device is None:
        device = next(decoder.parameters()).device
    z_tensor = torch.from_numpy(z).float().to(device)
    coo = W.tocoo()
    n_entries = len(coo.row)
    print(f"Graph has {n_entries} edges")
    if mode == "subset" and n_entries > max_edges:
        edges = coo.data
        cuts = np.linspace(0, 1, 6)
        picked = []
        for lo, hi in zip(cuts[:-1], cuts[1:]):
            q_lo, q_hi = np.quantile(edges, [lo, hi])
            pool = np.where((edges >= q_lo) & (edges <= q_hi))[0]
            take = min(max_edges // 5, len(pool))
            if take > 0:
                picked.extend(np.random.choice(pool, size=take, replace=False))
        chosen = np.array(picked)
        print(f"Reweighting {len(chosen)} edges (subset mode)")
    else:
        chosen = np.arange(n_entries)
        print(f"Reweighting all {len(chosen)} edges (full mode)")
    print(f"Computing Riemannian distances for {len(chosen)} edges...")
    with torch.no_grad():
        lengths = edge_lengths_riemannian(decoder, z_tensor[coo.row[chosen]], z_tensor[coo.col[chosen]],
                                          batch_size=batch_size).cpu().numpy()
    new = W.copy().astype(np.float32).tocoo()
    new.data[chosen] = lengths
    W_r = new.tocsr()
    W_r = W_r.maximum(W_r.T)
    finite = np.isfinite(W_r.data)
    if not finite.all():
        warnings.warn(f"Found {(~finite).sum()} non-finite Riemannian distances, keeping original Euclidean weights")
        W_r.data[~finite] = W.tocsr().data[~finite]
    print(f"Riemannian reweighting complete. Edge weight ratio: mean={np.mean(lengths / coo.data[chosen]):.3f}")
    return W_r


_DEFAULT_RUNS = {"mnist": "experiments/vae_mnist", "fashion": "experiments/vae_fashion", "cifar10": "experiments/vae_cifar10"}


def build_and_save(config: Dict) -> Path:
    data_cfg = config.get("data") if isinstance(config.get("data"), dict) else {}
    z_path = Path(data_cfg.get("latents_path") or
                  _DEFAULT_RUNS.get(str(data_cfg.get("dataset", "mnist")).strip().lower(), _DEFAULT_RUNS["mnist"]) + "/latents_train/z.pt")
    ckpt_cfg = (config.get("checkpoint_path") or config.get("vae", {}).get("ckpt_path")
                or (config.get("model", {}).get("checkpoint_path") if isinstance(config.get("model"), dict) else None))
    ckpt = Path(ckpt_cfg or _DEFAULT_RUNS.get(str(data_cfg.get("dataset", "fashion")).strip().lower(), _DEFAULT_RUNS["fashion"])
                + "/checkpoints/best.pt")
    out_dir = Path(config["out"]["dir"])
    out_dir.mkdir(parents=True, exist_ok=True)

    z = _load_latents(z_path).numpy()
    N, D = z.shape
    print(f"Loaded latents: N={N}, D={D}")
    device = torch.device("cuda" if torch.cuda.is_available() else "cpu")
    print(f"Using device: {device}")
    vae_config = config.get("vae_config") or config.get("model") or config.get("vae")
    if vae_config is None:
        raise ValueError("VAE configuration not found. Expected 'vae_config', 'model', or 'vae' key in config.")
    decoder = _load_vae_model(ckpt, vae_config, device).decoder

    g = config["graph"]
    k, metric, sym, mode = int(g["k"]), str(g["metric"]), str(g["sym"]), str(g["mode"])
    print(f"Building k-NN graph: k={k}, metric={metric}, sym={sym}")
    W_e, _ = build_knn_graph_auto(z, k=k, metric=metric, mode=mode, sym=sym)
    graph_stats = analyze_graph_connectivity(W_e)
    mask = largest_connected_component(W_e)
    if mask.sum() < W_e.shape[0]:
        print(f"Using LCC: {mask.sum()}/{W_e.shape[0]} nodes")
        W_e_lcc, z_lcc = W_e[mask][:, mask], z[mask]
    else:
        W_e_lcc, z_lcc = W_e, z

    r = config.get("riemannian", {})
    print(f"Re-weighting graph with Riemannian distances (mode={r.get('mode', 'subset')})")
    W_r = _reweight_graph_with_riemannian(W_e_lcc, z_lcc, decoder, mode=r.get("mode", "subset"),
                                          max_edges=int(r.get("max_edges", 5000)), batch_size=int(r.get("batch_size", 512)),
                                          device=device)
    sparse.save_npz(out_dir / "knn_graph_euclidean.npz", W_e_lcc)
    sparse.save_npz(out_dir / "knn_graph_riemannian.npz", W_r)

    q = config["quantize"]
    print(f"Running K-medoids on Riemannian graph: K={int(q['K'])}, init={q['init']}")
    medoids, assign_lcc, qe = fit_kmedoids_optimized(W_r, K=int(q["K"]), init=str(q["init"]), seed=int(q["seed"]))
    assign = np.full((N,), fill_value=-1, dtype=np.int32)
    if mask.sum() < N:
        assign[mask] = assign_lcc
    else:
        assign = assign_lcc
    torch.save({"medoid_indices": medoids.astype(np.int32), "z_medoid": torch.from_numpy(z_lcc[medoids]).float(),
                "config": config, "graph_stats": graph_stats, "method": "riemannian_geodesic"}, out_dir / "codebook.pt")
    np.save(out_dir / "codes.npy", assign)
    print(f"Riemannian quantization error: {qe:.3f}")
    print(f"Saved artifacts to: {out_dir}")
    return out_dir


if __name__ == "__main__":
    import yaml
    parser = argparse.ArgumentParser(description="Build Riemannian geodesic codebook")
    parser.add_argument("--config", type=str, default="configs/quantize.yaml", help="Configuration file path")
    with open(parser.parse_args().config, "r") as f:
        print(f"Completed: {build_and_save(yaml.safe_load(f))}")

"""Train the autoregressive prior over geodesic codes -- drop-in for the reference CLI
src/scripts/train_transformer.py (same YAML config: system / data / model / training / out), data-parallel when started
under torch.distributed.run (one process per GPU, RCCL):

    python -m vqvae_amd.scripts.train_transformer --config configs/.../transformer.yaml
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 -m vqvae_amd.scripts.train_transformer --config ...
"""
import argparse
import os
import random
from pathlib import Path

import numpy as np
import torch
import torch.distributed as dist
import yaml

from ..prior.codes_dataset import get_code_loaders
from ..prior.train import seed_dropout_stream, train_prior
from ..prior.transformer import Transformer


def set_seed(seed: int) -> None:
    random.seed(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed_all(seed)


def main(config_path: str):
    with open(config_path, "r") as f:
        cfg = yaml.safe_load(f)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    want = cfg["system"].get("device", "auto")
    use_gpu = torch.cuda.is_available() and want in ("auto", "cuda")
    device = torch.device("cuda", local % torch.cuda.device_count()) if use_gpu else torch.device("cpu")
    if use_gpu:
        torch.cuda.set_device(device)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # RCCL ("nccl") needs a GPU per rank; GEO_PRIOR_BACKEND=gloo lets ranks share one (rehearsals, 1-GPU boxes)
        if use_gpu and os.environ.get("GEO_PRIOR_BACKEND", "nccl") == "nccl":
            dist.init_process_group("nccl", device_id=device)
        else:
            dist.init_process_group("gloo")
    set_seed(cfg["system"]["seed"])          # every rank: same initial weights, same shuffling
    data_cfg, model_cfg, train_cfg = cfg["data"], cfg["model"], cfg["training"]
    # world > 1: the shuffling gets a generator of its own (seeded identically on every rank), so that the per-rank
    # dropout streams seeded below cannot pull the ranks' batch orders apart
    shuffler = None
    if world > 1:
        shuffler = torch.Generator()
        shuffler.manual_seed(int(cfg["system"]["seed"]))
    train_loader, val_loader = get_code_loaders(
        codes_path=data_cfg["codes_path"], labels_path=data_cfg.get("labels_path"), batch_size=data_cfg["batch_size"],
        num_workers=data_cfg["num_workers"], vanilla_vae=data_cfg.get("vanilla_vae", False),
        num_tokens=model_cfg.get("num_tokens"), device=device, generator=shuffler)
    model = Transformer(**model_cfg).to(device)
    if world > 1:
        seed_dropout_stream(cfg["system"]["seed"], dist.get_rank(), device)
    ckpt_dir = Path(cfg["out"]["dir"]) / "checkpoints"
    if not dist.is_initialized() or dist.get_rank() == 0:
        ckpt_dir.mkdir(parents=True, exist_ok=True)
    history = train_prior(model, train_loader, val_loader, epochs=int(train_cfg["epochs"]), lr=float(train_cfg["lr"]),
                          weight_decay=float(train_cfg["weight_decay"]), device=device, ckpt_dir=ckpt_dir)
    history["arena_sum"] = float(model.arena.detach().double().sum())       # one number to compare ranks by
    if dist.is_initialized():
        dist.destroy_process_group()
    return history


if __name__ == "__main__":
    parser = argparse.ArgumentParser()
    parser.add_argument("--config", type=str, required=True)
    main(parser.parse_args().config)

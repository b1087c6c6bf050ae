"""One data-parallel rank of tests/test_gpu_c5_pipeline.py: the prior's training CLI (vqvae_amd.scripts.train_transformer,
the reference's YAML) on the codes.npy a CIFAR-shaped codebook build wrote.  Fresh interpreter; RCCL when the box has a GPU
per rank, else the ranks share the GPU over gloo."""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

if __name__ == "__main__":
    from vqvae_amd.scripts import train_transformer as tt
    cfg_path, out = sys.argv[1], sys.argv[2]
    rank = int(os.environ["RANK"])
    history = tt.main(cfg_path)
    state = torch.load(os.path.join(os.path.dirname(cfg_path), "prior", "checkpoints", "latest.pt"), map_location="cpu") \
        if rank == 0 else None
    with open(os.path.join(out, f"c5_rank{rank}.json"), "w") as f:
        json.dump({"train_loss": history["train_loss"], "val_loss": history["val_loss"],
                   "arena_sum": history.get("arena_sum"), "keys": sorted(state.keys()) if state else None}, f)

"""One data-parallel rank of tests/test_gpu_prior_encoder.py (fresh interpreter; the ranks share the box's GPU, gloo)."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

if __name__ == "__main__":
    from test_prior_and_encoder import run_training
    rank = int(os.environ["RANK"])
    dev = torch.device("cuda", int(os.environ.get("LOCAL_RANK", "0")) % torch.cuda.device_count())
    torch.cuda.set_device(dev)
    dist.init_process_group("gloo")
    try:
        hist, norms = run_training(sys.argv[1], dev, epochs=2)
        np.savez(os.path.join(sys.argv[1], f"gpu_dp{rank}.npz"), train=np.array(hist["train_loss"]), norms=norms)
    finally:
        dist.destroy_process_group()

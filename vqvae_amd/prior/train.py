"""Data-parallel training of the code prior (reference loop: src/scripts/train_transformer.py:16-87, which
is single-process).  One process per GPU, `torch.distributed` over RCCL ("nccl" on ROCm; gloo in the CPU tests).

Every rank walks the SAME sequence of global batches (same seed -> same shuffling) and takes a contiguous slice of each
one.  Its loss is the SUM of its samples' token losses divided by the GLOBAL token count, so the all-reduced (summed)
gradients are exactly the gradients of the reference's mean loss over the global batch: N ranks reproduce the
single-process run up to float summation order -- for dropout = 0.  With dropout > 0 a rank draws the masks of ITS
slice from its own generator stream (`seed_dropout_stream`), so an N-rank run is a valid run of the same training
procedure but not a replay of the single-process one (whose masks depend on the batch being processed whole).
The model keeps all weights in one arena (vqvae_amd/prior/transformer.py), so all gradients are ONE flat buffer
(`model.arena.grad`), reduced with a single all-reduce per step: 3.3 M parameters = 13 MB, far below what a ring over xGMI
needs to be bandwidth-bound, so bucketing or overlap with backward would buy nothing here.
"""
import math
from pathlib import Path
from typing import Callable, Dict, Optional

import torch
import torch.distributed as dist
import torch.nn.functional as F
from torch.optim import AdamW
from torch.optim.lr_scheduler import CosineAnnealingLR

from ..parallel import block_range, world_info
from .native import ArenaAdamW, GraphedStep
from .transformer import Transformer


def _reduce_sum(t: torch.Tensor, group=None) -> None:
    if world_info(group)[1] > 1:
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)


def seed_dropout_stream(seed: int, rank: int, device: torch.device) -> None:
    """Give this rank its own dropout stream (call AFTER the model is built and the loaders hold their own generator):
    ranks must not repeat rank 0's masks on their slices."""
    if device.type == "cuda":
        torch.cuda.manual_seed(int(seed) + 7919 * (rank + 1))
    else:
        torch.manual_seed(int(seed) + 7919 * (rank + 1))


def _batch_to(batch, device, lo, hi, with_labels):
    x, y = batch[0][lo:hi].to(device, non_blocking=True), batch[1][lo:hi].to(device, non_blocking=True)
    labels = batch[2][lo:hi].to(device, non_blocking=True) if with_labels else None
    return x, y, labels


def train_prior(model: Transformer, train_loader, val_loader, *, epochs: int, lr: float, weight_decay: float,
                device: torch.device, ckpt_dir: Optional[Path] = None, group=None,
                on_step: Optional[Callable[[int, float], None]] = None, native: Optional[bool] = None) -> Dict[str, list]:
    """The reference's loop (AdamW, CosineAnnealingLR(T_max=epochs) stepped per epoch, cross-entropy over all positions,
    validation = mean of the per-batch losses over the un-shuffled data, best / latest state dicts) run data-parallel.
    Returns {"train_loss": per-step global mean losses, "val_loss": per-epoch}.

    On the GPU (`native`, default there): forward + backward of the full-size batch replay from ONE HIP graph
    (vqvae_amd/prior/native.py: GraphedStep), attention runs in the fused csrc/prior.hip kernels, the optimiser step is one
    launch over the arena (ArenaAdamW), and the per-step losses stay on the device until the epoch ends -- a step costs no
    host synchronisation.  The ragged last batch of an epoch runs eagerly.  On the CPU (tests, gloo) the same loop runs on
    torch's own AdamW."""
    rank, world = world_info(group)
    native = (device.type == "cuda") if native is None else (native and device.type == "cuda")
    if model.arena.grad is None:
        model.arena.grad = torch.zeros_like(model.arena)          # the one gradient buffer: zeroed, reduced, consumed in place
    grads = model.arena.grad
    if native:
        optimizer = ArenaAdamW(model.arena, lr=lr, weight_decay=weight_decay)
        scheduler = None
    else:
        optimizer = AdamW(model.parameters(), lr=lr, weight_decay=weight_decay)
        scheduler = CosineAnnealingLR(optimizer, T_max=int(epochs))
    with_labels = model.num_classes > 0
    seq = train_loader.dataset.seq_len - 1
    history = {"train_loss": [], "val_loss": []}
    best = float("inf")
    step = 0
    graphed, graphed_B = None, None

    def scaled_loss(B_global):
        def fn(x, y, labels):
            logits = model(x, y=labels)
            return F.cross_entropy(logits.reshape(-1, logits.size(-1)), y.reshape(-1), reduction="sum") / (B_global * seq)
        return fn

    for epoch in range(int(epochs)):
        model.train()
        pending = []                                              # per-step losses, still on the device
        for batch in train_loader:
            B = batch[0].shape[0]
            lo, hi = block_range(B, rank, world)
            x, y, labels = _batch_to(batch, device, lo, hi, with_labels)
            loss = torch.zeros((), device=device)
            if hi > lo and native and (graphed is None or graphed_B == B):
                if graphed is None:                               # first full batch: capture forward + backward once
                    graphed, graphed_B = GraphedStep(model, scaled_loss(B), x, y, labels), B
                loss = graphed.run(x, y, labels)
            else:
                grads.zero_()
                if hi > lo:
                    loss = scaled_loss(B)(x, y, labels)
                    loss.backward()
            _reduce_sum(grads, group)
            optimizer.step()
            total = loss.detach().clone()
            _reduce_sum(total, group)                             # sum of the ranks' shares of the global mean loss
            pending.append(total)
            if on_step is not None:
                on_step(step, float(total))
            step += 1
        history["train_loss"].extend(torch.stack(pending).tolist() if pending else [])
        if scheduler is not None:
            scheduler.step()
        else:                                                     # CosineAnnealingLR(T_max=epochs, eta_min=0), closed form
            optimizer.param_groups[0]["lr"] = 0.5 * lr * (1.0 + math.cos(math.pi * (epoch + 1) / int(epochs)))
        model.eval()
        val, n_batches = torch.zeros((), device=device), 0
        with torch.no_grad():
            for batch in val_loader:                 # identical on every rank (tiny model): no collective needed
                x, y, labels = _batch_to(batch, device, 0, batch[0].shape[0], with_labels)
                logits = model(x, y=labels)
                val += F.cross_entropy(logits.reshape(-1, logits.size(-1)), y.reshape(-1))
                n_batches += 1
        val = float(val) / max(1, n_batches)
        history["val_loss"].append(val)
        if rank == 0:
            print(f"Epoch {len(history['val_loss'])}: Val Loss = {val:.4f}")
            if ckpt_dir is not None:
                if val < best:
                    torch.save(model.state_dict(), Path(ckpt_dir) / "best.pt")
                torch.save(model.state_dict(), Path(ckpt_dir) / "latest.pt")
        best = min(best, val)
    return history

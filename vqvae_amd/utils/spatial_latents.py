"""Latent production on the MI355X: the reference's save_spatial_latents
(src/utils/spatial_latents.py:10-36) plus a resident variant that hands the latents to the codebook builder without
leaving HBM (the reference writes z.pt and build_codebook.py reads it back: a host round trip of N*C*H*W*4 bytes)."""
from pathlib import Path
from typing import Iterable, Tuple

import torch


@torch.no_grad()
def encode_latents_device(model, loader: Iterable, device: torch.device):
    """Runs the VAE in eval mode over `loader` ((x, y) batches) on `device`; returns (z, mu, logvar, y) with the three
    latent tensors (N, C, H, W) RESIDENT on the device and y on the host."""
    model.eval()
    zs, mus, logvars, ys = [], [], [], []
    for x, y in loader:
        _, mu, logvar, z = model(x.to(device, non_blocking=True))
        zs.append(z), mus.append(mu), logvars.append(logvar), ys.append(y)
    return torch.cat(zs), torch.cat(mus), torch.cat(logvars), torch.cat(ys)


def flatten_latents_device(z: torch.Tensor) -> torch.Tensor:
    """(N, C, H, W) -> (N*H*W, C), row = (n, h, w): the node order of build_codebook.py:35, on the device."""
    return z.permute(0, 2, 3, 1).reshape(-1, z.shape[1]).contiguous()


def save_spatial_latents(model, loader: Iterable, device: torch.device, out_dir: Path) -> None:
    """Same files as the reference: z.pt, mu.pt, logvar.pt, y.pt (4-D CPU tensors) in out_dir."""
    z, mu, logvar, y = encode_latents_device(model, loader, device)
    out_dir = Path(out_dir)
    out_dir.mkdir(parents=True, exist_ok=True)
    torch.save(z.cpu(), out_dir / "z.pt")
    torch.save(mu.cpu(), out_dir / "mu.pt")
    torch.save(logvar.cpu(), out_dir / "logvar.pt")
    torch.save(y, out_dir / "y.pt")
    print(f"Saved spatial latents to {out_dir}")


def latents_to_codebook_device(model, loader: Iterable, device: torch.device, **codebook_args) -> Tuple[dict, torch.Tensor]:
    """encoder -> z (resident) -> geodesic codebook, with the model's own decoder left in training mode as
    src/scripts/build_codebook.py:27-29 leaves it.  Returns (build_codebook_device result, y)."""
    from ..scripts.build_codebook import build_codebook_device
    z, _, _, y = encode_latents_device(model, loader, device)
    model.decoder.train()
    res = build_codebook_device(flatten_latents_device(z), model.decoder, **codebook_args)
    res["latent_shape"] = tuple(z.shape)
    return res, y

"""The decoder of the vanilla (vector-latent) VAE of the legacy builders (reference src/models/vae.py:53-85): Linear -> ConvT
stack with the reference's parameter names, so the `decoder.*` entries of a reference checkpoint load unchanged.  The legacy
Riemannian builder differentiates this module -- Linear-first, so edge_lengths_riemannian takes its autograd path on the GPU
(riemannian_metric.py:18-22).  Encoder, loss and sampling of the reference's VAE are not part of the hot path and not built."""
from typing import Dict, Sequence

import torch
import torch.nn as nn

from .spatial_decoder import make_norm


class Decoder(nn.Module):
    """fc to a 4x4 grid, ConvT(k3,s2,p1[,output_padding for 32-px]) -> 7x7 | 8x8, ConvT(k4,s2,p1) x 2 -> 28 | 32 px."""

    def __init__(self, out_channels: int = 1, channels: Sequence[int] = (128, 64, 32), latent_dim: int = 16,
                 output_image_size: int = 28, norm_type: str = "none"):
        super().__init__()
        self.fc = nn.Linear(latent_dim, channels[0] * 4 * 4)
        self.deconv1 = nn.Sequential(
            nn.ConvTranspose2d(channels[0], channels[1], 3, stride=2, padding=1,
                               output_padding=1 if output_image_size == 32 else 0),
            make_norm(norm_type, channels[1]), nn.ReLU(inplace=True))
        self.deconv2 = nn.Sequential(nn.ConvTranspose2d(channels[1], channels[2], 4, stride=2, padding=1),
                                     make_norm(norm_type, channels[2]), nn.ReLU(inplace=True))
        self.output_layer = nn.ConvTranspose2d(channels[2], out_channels, 4, stride=2, padding=1)

    def forward(self, z: torch.Tensor) -> torch.Tensor:
        h = self.fc(z).view(z.size(0), -1, 4, 4)
        return self.output_layer(self.deconv2(self.deconv1(h)))


def decoder_from_vae_checkpoint(state: Dict[str, torch.Tensor], in_channels: int = 1, dec_channels: Sequence[int] = (128, 64, 32),
                                latent_dim: int = 16, output_image_size: int = 28, norm_type: str = "none", **_other) -> Decoder:
    """The decoder of a vanilla-VAE checkpoint (reference `VAE.state_dict()`: `encoder.*` and `decoder.*` entries).  Only the
    `decoder.*` entries are read -- the builders differentiate the decoder and never encode -- and they must match exactly."""
    dec = Decoder(in_channels, tuple(dec_channels), latent_dim, output_image_size, norm_type)
    own = {k[len("decoder."):]: v for k, v in state.items() if k.startswith("decoder.")}
    dec.load_state_dict(own if own else state)           # (a bare decoder state dict is accepted too)
    return dec

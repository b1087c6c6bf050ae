"""SpatialVAE for latent production on the MI355X: the reference's encoder half
(src/models/spatial_vae.py:22-44) next to the decoder of vqvae_amd/spatial_decoder.py, with the SAME parameter names,
so a `best.pt` written by the reference's trainer ({'model_state_dict', 'epoch'}, spatial_engine.py:142) loads unchanged.
Only inference is kept (encode -> mu, logvar, z): training the VAE is outside the geodesic-codebook path.  The layers
are PyTorch-ROCm modules (device memory and convolutions are plumbing here; the path's own arithmetic is in csrc/)."""
from typing import Sequence, Tuple

import torch
import torch.nn as nn

from .spatial_decoder import SpatialDecoder, make_norm


class SpatialEncoder(nn.Module):
    """Stride-2 3x3 convolutions with norm + ReLU, then 1x1 heads for the mean and log-variance grids."""

    def __init__(self, input_channels: int, channels: Sequence[int], latent_dim: int, norm_type: str):
        super().__init__()
        layers, prev = [], input_channels
        for ch in channels:
            layers += [nn.Conv2d(prev, ch, 3, stride=2, padding=1), make_norm(norm_type, ch), nn.ReLU(inplace=True)]
            prev = ch
        self.conv_layers = nn.Sequential(*layers)
        self.fc_mu = nn.Conv2d(prev, latent_dim, 1)
        self.fc_logvar = nn.Conv2d(prev, latent_dim, 1)

    def forward(self, x: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
        h = self.conv_layers(x)
        return self.fc_mu(h), self.fc_logvar(h)


class SpatialVAE(nn.Module):
    """encoder + decoder under the reference's attribute names (spatial_vae.py:84-104); forward returns
    (x_logits, mu, logvar, z) with z = mu + eps * exp(logvar / 2), eps ~ N(0, 1) from torch's generator."""

    def __init__(self, in_channels, enc_channels, dec_channels, latent_dim, recon_loss, output_image_size, norm_type,
                 **kwargs):
        super().__init__()
        assert recon_loss in {"bce", "mse"}
        self.encoder = SpatialEncoder(in_channels, tuple(enc_channels), latent_dim, norm_type)
        self.decoder = SpatialDecoder(in_channels, tuple(dec_channels), latent_dim, output_image_size, norm_type)
        self.recon_loss = recon_loss
        self.mse_use_sigmoid = kwargs.get("mse_use_sigmoid", True)

    @staticmethod
    def reparameterize(mu: torch.Tensor, logvar: torch.Tensor) -> torch.Tensor:
        std = torch.exp(0.5 * logvar)
        return mu + torch.randn_like(std) * std

    def forward(self, x: torch.Tensor):
        mu, logvar = self.encoder(x)
        z = self.reparameterize(mu, logvar)
        return self.decoder(z), mu, logvar, z

"""The vanilla (vector-latent) VAE of the legacy builders (reference src/models/vae.py:22-85,88-122): Linear -> ConvT
decoder, conv encoder with Linear heads, same parameter names (a reference checkpoint loads unchanged).  Inference only
(encode / decode): the legacy Riemannian builder differentiates `decoder` -- a Linear-first module, so
edge_lengths_riemannian takes its autograd path on the GPU (riemannian_metric.py:18-22)."""
from typing import Sequence, Tuple

import torch
import torch.nn as nn

from .spatial_decoder import make_norm


class Encoder(nn.Module):
    def __init__(self, input_channels: int = 1, channels: Sequence[int] = (32, 64, 128), latent_dim: int = 16,
                 norm_type: str = "none"):
        super().__init__()
        layers, prev = [], input_channels
        for ch in channels:
            layers += [nn.Conv2d(prev, ch, 3, stride=2, padding=1), make_norm(norm_type, ch), nn.ReLU(inplace=True)]
            prev = ch
        self.conv_layers = nn.Sequential(*layers)
        self.feature_dim = channels[-1] * 4 * 4
        self.fc_mu = nn.Linear(self.feature_dim, latent_dim)
        self.fc_logvar = nn.Linear(self.feature_dim, latent_dim)

    def forward(self, x: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
        h = self.conv_layers(x).flatten(1)
        return self.fc_mu(h), self.fc_logvar(h)


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


class VAE(nn.Module):
    def __init__(self, in_channels=1, enc_channels=(32, 64, 128), dec_channels=(128, 64, 32), latent_dim=16,
                 recon_loss="bce", output_image_size: int = 28, norm_type: str = "none", mse_use_sigmoid: bool = True,
                 **_training_defaults):
        super().__init__()
        assert recon_loss in {"bce", "mse"}, f"recon_loss must be 'bce' or 'mse', got {recon_loss}"
        self.encoder = Encoder(in_channels, tuple(enc_channels), latent_dim, norm_type)
        self.decoder = Decoder(in_channels, tuple(dec_channels), latent_dim, output_image_size, norm_type)
        self.recon_loss, self.mse_use_sigmoid = recon_loss, mse_use_sigmoid

    @staticmethod
    def reparameterize(mu: torch.Tensor, logvar: torch.Tensor) -> torch.Tensor:
        std = torch.exp(0.5 * logvar)
        return mu + torch.randn_like(std) * std

    def forward(self, x: torch.Tensor):
        mu, logvar = self.encoder(x)
        z = self.reparameterize(mu, logvar)
        return self.decoder(z), mu, logvar, z

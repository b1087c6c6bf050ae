"""Host-side description of the decoder the JVP kernels differentiate.

The function being differentiated is the reference's SpatialDecoder
(src/models/spatial_vae.py:47-81; norm factory :8-19).  Only what the geodesic-codebook path needs
is kept here: a module with the SAME parameter names (so `best.pt` state dicts load and the
reference's own SpatialDecoder instances are accepted by duck typing) and the export of its
weights into the `geo_decoder_desc` of include/geo_hip.h.  Encoder, losses and training are out
of scope.
"""
import ctypes
from typing import Mapping, Optional, Sequence, Tuple

import torch
import torch.nn as nn

from . import _lib

_NORM_CODE = {"none": 0, "batch": 1, "group": 2}


def make_norm(norm_type: Optional[str], channels: int) -> nn.Module:
    """batch -> BatchNorm2d, group -> GroupNorm with the largest divisor of `channels` <= 32, else identity."""
    kind = (norm_type or "none").lower()
    if kind == "batch":
        return nn.BatchNorm2d(channels)
    if kind == "group":
        groups = next(g for g in range(max(1, min(32, channels)), 0, -1) if channels % g == 0)
        return nn.GroupNorm(groups, channels)
    return nn.Identity()


class SpatialDecoder(nn.Module):
    """1x1 conv to channels[0], two stride-2 transposed convs with norm + ReLU, one output transposed conv."""

    LAST_PADDING = {32: 1, 28: 3}

    def __init__(self, out_channels: int, channels: Sequence[int], latent_dim: int, output_image_size: int,
                 norm_type: Optional[str]):
        super().__init__()
        if output_image_size not in self.LAST_PADDING:
            raise ValueError(f"Unsupported output size: {output_image_size}")
        c0, c1, c2 = channels
        self.conv_in = nn.Conv2d(latent_dim, c0, 1)
        self.deconv_layers = nn.Sequential(
            nn.ConvTranspose2d(c0, c1, 4, stride=2, padding=1), make_norm(norm_type, c1), nn.ReLU(inplace=True),
            nn.ConvTranspose2d(c1, c2, 4, stride=2, padding=1), make_norm(norm_type, c2), nn.ReLU(inplace=True),
            nn.ConvTranspose2d(c2, out_channels, 4, stride=2, padding=self.LAST_PADDING[output_image_size]),
        )

    def forward(self, z: torch.Tensor) -> torch.Tensor:
        return self.deconv_layers(self.conv_in(z))


def load_decoder_from_checkpoint(ckpt_path: str, *, in_channels: int, dec_channels: Sequence[int], latent_dim: int,
                                 output_image_size: int, norm_type: str, device) -> SpatialDecoder:
    """Decoder half of a SpatialVAE checkpoint ({'model_state_dict', 'epoch'}, spatial_engine.py:142).
    The module is left in training mode, as src/scripts/build_codebook.py:27-29 leaves it."""
    state = torch.load(ckpt_path, map_location="cpu")["model_state_dict"]
    dec_state = {k[len("decoder."):]: v for k, v in state.items() if k.startswith("decoder.")}
    dec = SpatialDecoder(in_channels, tuple(dec_channels), latent_dim, output_image_size, norm_type)
    dec.load_state_dict(dec_state)
    return dec.to(device)


def looks_like_spatial_decoder(m: nn.Module) -> bool:
    """True for this class and for any module with the reference SpatialDecoder's layer layout."""
    conv_in, seq = getattr(m, "conv_in", None), getattr(m, "deconv_layers", None)
    if not isinstance(conv_in, nn.Conv2d) or not isinstance(seq, nn.Sequential) or len(seq) != 7:
        return False
    if conv_in.kernel_size != (1, 1) or conv_in.stride != (1, 1) or conv_in.padding != (0, 0):
        return False
    if not all(isinstance(seq[i], nn.ConvTranspose2d) for i in (0, 3, 6)):
        return False
    if not all(isinstance(seq[i], nn.ReLU) for i in (2, 5)):
        return False
    for i, pads in ((0, (1,)), (3, (1,)), (6, (1, 3))):
        c = seq[i]
        if (c.kernel_size != (4, 4) or c.stride != (2, 2) or c.padding[0] not in pads or c.padding[0] != c.padding[1]
                or c.output_padding != (0, 0) or c.dilation != (1, 1) or c.groups != 1):
            return False
    kinds = {type(seq[1]), type(seq[4])}
    return len(kinds) == 1 and kinds <= {nn.BatchNorm2d, nn.GroupNorm, nn.Identity}


def hip_kernels_cover(m: nn.Module) -> bool:
    """Whether csrc/jvp.hip implements this SpatialDecoder-shaped module (else the caller differentiates it
    with autograd): dec_channels[1] in {32,64,128}, dec_channels[2] a multiple of 16 dividing 128,
    latent_dim <= 64, LDS budget of the output stage; GroupNorm with 32 groups per layer and
    dec_channels[2] == 64 (the reference's default decoder 256-128-64)."""
    seq = m.deconv_layers
    d, c1, c2, co = m.conv_in.in_channels, seq[0].out_channels, seq[3].out_channels, seq[6].out_channels
    if isinstance(seq[1], nn.GroupNorm):
        if seq[1].num_groups != 32 or seq[4].num_groups != 32 or c2 != 64:
            return False
    s_out = 8 if seq[6].padding[0] == 1 else 4
    back_lds = (2 * 8 * 16 * (c2 + 4) + 16 * co * (c2 + 4) + 8 * co * s_out * s_out) * 4
    return (d <= 64 and c1 in (32, 64, 128) and c2 % 16 == 0 and 128 % c2 == 0 and back_lds <= 160 * 1024
            and all(getattr(l, "bias", None) is not None for l in (m.conv_in, seq[0], seq[3], seq[6])))


class DecoderExport:
    """f32 contiguous copies of a decoder's parameters on `dev` plus the ctypes descriptor over them."""

    def __init__(self, dec: nn.Module, dev: torch.device):
        seq = dec.deconv_layers
        norm = seq[1]

        def f(t):
            return None if t is None else t.detach().to(dev, torch.float32).contiguous()

        self.tensors = {
            "w_in": f(dec.conv_in.weight.view(dec.conv_in.out_channels, -1)), "b_in": f(dec.conv_in.bias),
            "w1": f(seq[0].weight), "b1": f(seq[0].bias), "w2": f(seq[3].weight), "b2": f(seq[3].bias),
            "w3": f(seq[6].weight), "b3": f(seq[6].bias),
        }
        for tag, layer in (("1", seq[1]), ("2", seq[4])):
            affine = getattr(layer, "weight", None) is not None
            self.tensors["g" + tag] = f(layer.weight) if affine else None
            self.tensors["be" + tag] = f(layer.bias) if affine else None
            self.tensors["rm" + tag] = f(getattr(layer, "running_mean", None))
            self.tensors["rv" + tag] = f(getattr(layer, "running_var", None))
        for name in ("b_in", "b1", "b2", "b3"):
            if self.tensors[name] is None:
                raise ValueError("decoder convolutions without bias are not supported")
        d = _lib.DecoderDesc()
        d.latent_dim = dec.conv_in.in_channels
        d.c0, d.c1, d.c2 = dec.conv_in.out_channels, seq[0].out_channels, seq[3].out_channels
        d.out_channels = seq[6].out_channels
        d.out_size = {1: 32, 3: 28}[seq[6].padding[0]]
        if isinstance(norm, nn.BatchNorm2d):
            d.norm, d.eps = 1, norm.eps
            # batch statistics are used when the NORM layer is in training mode (torch looks at the layer's own flag, not
            # at its parent's), or when no running statistics are tracked
            d.bn_train = 1 if norm.training or norm.running_mean is None else 0
            if norm.weight is None:
                ones = {t: torch.ones(c, device=dev) for t, c in (("1", d.c1), ("2", d.c2))}
                for t in ("1", "2"):
                    self.tensors["g" + t], self.tensors["be" + t] = ones[t], torch.zeros_like(ones[t])
        elif isinstance(norm, nn.GroupNorm):
            d.norm, d.eps, d.bn_train = 2, norm.eps, 0
            d.groups1, d.groups2 = seq[1].num_groups, seq[4].num_groups
            if norm.weight is None:                      # affine=False
                for t, c in (("1", d.c1), ("2", d.c2)):
                    self.tensors["g" + t] = torch.ones(c, device=dev)
                    self.tensors["be" + t] = torch.zeros(c, device=dev)
        else:
            d.norm, d.eps, d.bn_train = 0, 1e-5, 0
        # train-mode BatchNorm with tracked statistics: the kernels fold every batch into running_mean / running_var
        # as torch does (riemannian_metric.py:57-58 runs the decoder in whatever mode it is in); a cumulative average
        # (momentum=None) is not reproduced
        self._bn_layers = [seq[1], seq[4]] if isinstance(norm, nn.BatchNorm2d) else []
        self.tracks_running = bool(d.bn_train and self._bn_layers and norm.running_mean is not None
                                   and norm.momentum is not None)
        d.update_running = 1 if self.tracks_running else 0
        d.momentum = float(norm.momentum) if self.tracks_running else 0.0
        for name, t in self.tensors.items():
            setattr(d, name, None if t is None else ctypes.c_void_p(t.data_ptr()))
        self.desc = d

    def commit_running_stats(self, n_calls: int) -> None:
        """After a JVP run over `n_calls` decoder calls (2 per chunk): write the updated statistics back when the export
        holds copies (decoder on another device / dtype) and advance num_batches_tracked."""
        if not self.tracks_running:
            return
        with torch.no_grad():
            for tag, layer in zip(("1", "2"), self._bn_layers):
                for attr, key in (("running_mean", "rm" + tag), ("running_var", "rv" + tag)):
                    buf, mine = getattr(layer, attr), self.tensors[key]
                    if buf.data_ptr() != mine.data_ptr():
                        buf.copy_(mine.to(buf.device, buf.dtype))
                if layer.num_batches_tracked is not None:
                    layer.num_batches_tracked += n_calls

"""Riemannian (decoder pull-back) edge lengths on the MI355X -- same API as the reference's
src/geo/riemannian_metric.py (decoder_logits_to_img :7, edge_lengths_riemannian :37).

For decoders with the SpatialDecoder layer layout the Jacobian-vector products run in the HIP
kernels of csrc/jvp.hip (forward-mode tangent propagation, MFMA for the dominant ConvT layer),
chunked by `batch_size` exactly like riemannian_metric.py:50-58 so that train-mode BatchNorm sees
the same batches.  Any other nn.Module (e.g. the Linear test decoder of the reference's
tests/test_riemannian_metric.py, the vanilla VAE decoder, or a SpatialDecoder variant outside the kernels'
coverage, see spatial_decoder.hip_kernels_cover) has no kernel: it is differentiated by autograd on the
decoder's own device.  BatchNorm (train and eval), GroupNorm and no normalisation are all in the kernels.
"""
import torch

from .. import _lib
from .._device import device, ptr, stream_ptr, workspace
from ..spatial_decoder import DecoderExport, hip_kernels_cover, looks_like_spatial_decoder


@torch.no_grad()
def decoder_logits_to_img(logits: torch.Tensor) -> torch.Tensor:
    """Decoder logits -> image space [0,1] (riemannian_metric.py:7-10)."""
    return torch.sigmoid(logits)


def edge_lengths_device(export: DecoderExport, z_start: torch.Tensor, z_end: torch.Tensor,
                        batch_size: int = 512) -> torch.Tensor:
    """HIP path on explicit endpoint arrays (f32, contiguous, on the export's device)."""
    lib = _lib.load()
    dev = z_start.device
    E = z_start.shape[0]
    out = torch.empty(E, dtype=torch.float32, device=dev)
    if E == 0:
        return out
    nbytes = lib.geo_jvp_workspace_bytes(export.desc, E, int(batch_size))
    if nbytes == 0:
        raise _lib.GeoHipError("geo_jvp_workspace_bytes: decoder configuration not supported by the HIP path")
    ws = workspace(nbytes, dev)
    with torch.cuda.device(dev):
        _lib.check(lib.geo_decoder_jvp_pairs(export.desc, ptr(z_start), ptr(z_end), E, int(batch_size), ptr(out),
                                             ptr(ws), ws.numel(), stream_ptr()), "geo_decoder_jvp_pairs")
    export.commit_running_stats(2 * ((E + int(batch_size) - 1) // int(batch_size)))
    return out


def edge_lengths_graph_device(export: DecoderExport, z: torch.Tensor, src: torch.Tensor, dst: torch.Tensor,
                              batch_size: int = 512) -> torch.Tensor:
    """HIP path on an edge list over resident latents: len[e] for (src[e], dst[e]) without gathers in HBM."""
    lib = _lib.load()
    dev = z.device
    E = int(src.numel())
    out = torch.empty(E, dtype=torch.float32, device=dev)
    if E == 0:
        return out
    # (with the per-latent buffers: decoders with fixed statistics run their primal pass once per latent, not per edge end)
    nbytes = lib.geo_jvp_edges_workspace_bytes(export.desc, int(z.shape[0]), E, int(batch_size))
    if nbytes == 0:
        raise _lib.GeoHipError("geo_jvp_edges_workspace_bytes: decoder configuration not supported by the HIP path")
    ws = workspace(nbytes, dev)
    with torch.cuda.device(dev):
        _lib.check(lib.geo_decoder_jvp_edges(export.desc, ptr(z), z.shape[0], ptr(src), ptr(dst), E, int(batch_size),
                                             ptr(out), ptr(ws), ws.numel(), stream_ptr()), "geo_decoder_jvp_edges")
    export.commit_running_stats(2 * ((E + int(batch_size) - 1) // int(batch_size)))
    return out


def _generic_jvp_norms(decoder, z: torch.Tensor, direction: torch.Tensor) -> torch.Tensor:
    """|J(z) v| for a decoder the kernels cannot represent, by autograd on the decoder's own device
    (works for any module, including train-mode BatchNorm with its running-statistic updates).
    Linear-first decoders take 2-D input, conv-first ones a 1x1 latent image (riemannian_metric.py:18-27)."""
    first = next(decoder.children())
    flat_input = hasattr(first, "in_features")

    def image(latent):
        if not flat_input and latent.ndim == 2:
            latent = latent[:, :, None, None]
        return torch.sigmoid(decoder(latent)).flatten(1)

    _, tangent = torch.autograd.functional.jvp(image, (z,), (direction,))
    return torch.linalg.vector_norm(tangent, dim=1)


@torch.no_grad()
def edge_lengths_riemannian(decoder, z_start: torch.Tensor, z_end: torch.Tensor, batch_size: int = 512) -> torch.Tensor:
    """0.5 * (|J(z_i) dz| + |J(z_j) dz|) per edge, float32, on the decoder's device (riemannian_metric.py:37-66)."""
    assert z_start.shape == z_end.shape, "Start and end points must have same shape"
    dec_dev = next(decoder.parameters()).device
    if looks_like_spatial_decoder(decoder) and z_start.ndim == 2 and hip_kernels_cover(decoder):
        dev = dec_dev if dec_dev.type == "cuda" else device()
        export = DecoderExport(decoder, dev)
        zs = z_start.detach().to(dev, torch.float32).contiguous()
        ze = z_end.detach().to(dev, torch.float32).contiguous()
        return edge_lengths_device(export, zs, ze, batch_size).to(dec_dev)
    z_start, z_end = z_start.to(dec_dev), z_end.to(dec_dev)
    delta = z_end - z_start
    pieces = []
    for lo in range(0, z_start.size(0), batch_size):
        sl = slice(lo, min(lo + batch_size, z_start.size(0)))
        pieces.append(0.5 * (_generic_jvp_norms(decoder, z_start[sl], delta[sl])
                             + _generic_jvp_norms(decoder, z_end[sl], delta[sl])))
    if not pieces:
        return torch.empty(0, dtype=torch.float32, device=dec_dev)
    return torch.cat(pieces).to(torch.float32)

"""CPU restatement of src/geo/riemannian_metric.py over src/models/spatial_vae.py:47-81 (reference).
TEST INFRASTRUCTURE ONLY.

The reference obtains J(z)·δ by the double-backward trick of torch.autograd.functional.jvp
(riemannian_metric.py:32).  This file states the same quantity in closed form: the tangent is
pushed forward layer by layer next to the primal (forward-mode), for the SpatialDecoder layer
sequence  conv1x1 -> [ConvT(k4,s2,p1) -> norm -> ReLU] x2 -> ConvT(k4,s2,p=3|1) -> sigmoid.

The decoder is described by its state_dict (the reference's key names: "conv_in.weight",
"deconv_layers.{0,3,6}.weight/bias", "deconv_layers.{1,4}.weight/bias/running_mean/running_var"),
the norm type and the output image size -- no nn.Module is needed.
"""
from typing import Mapping

import numpy as np
import torch
import torch.nn.functional as F

BN_EPS = 1e-5
GN_EPS = 1e-5


def group_count(channels: int) -> int:
    """spatial_vae.py:13-16."""
    g = max(1, min(32, channels))
    while channels % g != 0 and g > 1:
        g -= 1
    return g


_DEVICE = "cpu"     # where the restatement runs; edge_lengths(device=...) may move the CHECKER to the GPU (fp64 torch)


def _t(sd: Mapping, key: str, dtype):
    v = sd[key]
    if not torch.is_tensor(v):
        v = torch.from_numpy(np.asarray(v))
    return v.detach().to(_DEVICE, dtype)


def _norm_push(x, t, sd, prefix, norm_type, training, dtype):
    """Primal and tangent through one normalisation layer."""
    if norm_type == "none":
        return x, t
    gamma = _t(sd, prefix + ".weight", dtype).view(1, -1, 1, 1)
    beta = _t(sd, prefix + ".bias", dtype).view(1, -1, 1, 1)
    if norm_type == "batch" and not training:
        rm = _t(sd, prefix + ".running_mean", dtype).view(1, -1, 1, 1)
        rv = _t(sd, prefix + ".running_var", dtype).view(1, -1, 1, 1)
        inv = 1.0 / torch.sqrt(rv + BN_EPS)
        return (x - rm) * inv * gamma + beta, t * inv * gamma
    if norm_type == "batch":
        dims, eps = (0, 2, 3), BN_EPS
        xs, ts = x, t
    elif norm_type == "group":
        B, C, H, Wd = x.shape
        g = group_count(C)
        xs, ts = x.reshape(B, g, -1), t.reshape(B, g, -1)
        dims, eps = (2,), GN_EPS
    else:
        raise ValueError(norm_type)
    # statistics are accumulated in float64 and rounded once, like torch's CPU kernels (acc_type<float> is
    # double there); plain float32 means lose up to 1e-2 on batches with |mean| >> std (few distinct samples)
    acc = torch.float64
    mu = xs.to(acc).mean(dim=dims, keepdim=True)
    var = ((xs.to(acc) - mu) ** 2).mean(dim=dims, keepdim=True)          # biased, as torch normalises
    inv = (1.0 / torch.sqrt(var + eps))
    xhat64 = (xs.to(acc) - mu) * inv
    mean_t = ts.to(acc).mean(dim=dims, keepdim=True)
    mean_xt = (xhat64 * ts.to(acc)).mean(dim=dims, keepdim=True)
    mu, inv, mean_t, mean_xt = (v.to(dtype) for v in (mu, inv, mean_t, mean_xt))
    xhat = (xs - mu) * inv
    that = inv * (ts - mean_t - xhat * mean_xt)
    xhat, that = xhat.reshape(x.shape), that.reshape(x.shape)
    return xhat * gamma + beta, that * gamma


def jvp_norms(sd: Mapping, norm_type: str, output_image_size: int, z, delta, training: bool,
              dtype=torch.float32) -> torch.Tensor:
    """|| d/de sigmoid(decoder(z + e*delta)) ||_2 per row, for one batch (BN statistics are taken over it)."""
    norm_type = (norm_type or "none").lower()
    if norm_type not in ("batch", "group"):
        norm_type = "none"
    pad_last = {28: 3, 32: 1}[int(output_image_size)]
    z = torch.as_tensor(z).to(_DEVICE, dtype)
    delta = torch.as_tensor(delta).to(_DEVICE, dtype)
    x = z.view(z.shape[0], -1, 1, 1)
    t = delta.view(delta.shape[0], -1, 1, 1)
    w = _t(sd, "conv_in.weight", dtype)
    x = F.conv2d(x, w, _t(sd, "conv_in.bias", dtype))
    t = F.conv2d(t, w)
    for conv, norm in ((0, 1), (3, 4)):
        w = _t(sd, f"deconv_layers.{conv}.weight", dtype)
        x = F.conv_transpose2d(x, w, _t(sd, f"deconv_layers.{conv}.bias", dtype), stride=2, padding=1)
        t = F.conv_transpose2d(t, w, None, stride=2, padding=1)
        x, t = _norm_push(x, t, sd, f"deconv_layers.{norm}", norm_type, training, dtype)
        t = t * (x > 0).to(dtype)
        x = torch.relu(x)
    w = _t(sd, "deconv_layers.6.weight", dtype)
    x = F.conv_transpose2d(x, w, _t(sd, "deconv_layers.6.bias", dtype), stride=2, padding=pad_last)
    t = F.conv_transpose2d(t, w, None, stride=2, padding=pad_last)
    s = torch.sigmoid(x)
    jt = (t * s * (1.0 - s)).reshape(z.shape[0], -1)
    return torch.linalg.vector_norm(jt, dim=1)


def edge_lengths(sd: Mapping, norm_type: str, output_image_size: int, z_start, z_end, batch_size: int = 512,
                 training: bool = True, dtype=torch.float32, device: str = "cpu") -> torch.Tensor:
    """riemannian_metric.py:37-66: 0.5*(|J(z_i)δ| + |J(z_j)δ|), chunked exactly like the reference.
    `device` lets a full-size GPU test run this same closed form in fp64 torch on the GPU (a torch reference of
    the floating-point kernel, tied to the CPU run on sample chunks by the test); the result returns to the CPU."""
    global _DEVICE
    prev, _DEVICE = _DEVICE, device
    try:
        sd = {k: _t(sd, k, torch.float64 if k.endswith("num_batches_tracked") else dtype) for k in sd}
        z_start = torch.as_tensor(z_start).to(device, dtype)
        z_end = torch.as_tensor(z_end).to(device, dtype)
        assert z_start.shape == z_end.shape, "Start and end points must have same shape"
        delta = z_end - z_start
        out = []
        with torch.no_grad():
            for lo in range(0, z_start.shape[0], batch_size):
                hi = min(lo + batch_size, z_start.shape[0])
                a = jvp_norms(sd, norm_type, output_image_size, z_start[lo:hi], delta[lo:hi], training, dtype)
                b = jvp_norms(sd, norm_type, output_image_size, z_end[lo:hi], delta[lo:hi], training, dtype)
                out.append(0.5 * (a + b))
        if not out:
            return torch.empty(0, dtype=torch.float32)
        return torch.cat(out).to("cpu", torch.float32)
    finally:
        _DEVICE = prev


def _dense_maps(sd: Mapping, output_image_size: int, dtype):
    """The three transposed convolutions on the decoder's 1x1 latent image as dense matrices (an exact copy of the
    weights: each row is the layer applied to a one-hot input, bias removed).  Shapes (c_in*h*w, c_out*2h*2w)."""
    pad_last = {28: 3, 32: 1}[int(output_image_size)]
    maps, hw = [], 1
    for conv, pad in ((0, 1), (3, 1), (6, pad_last)):
        w = _t(sd, f"deconv_layers.{conv}.weight", dtype)
        cin = w.shape[0]
        eye = torch.eye(cin * hw * hw, dtype=dtype, device=w.device).view(cin * hw * hw, cin, hw, hw)
        out = F.conv_transpose2d(eye, w, None, stride=2, padding=pad)
        maps.append((out.reshape(cin * hw * hw, -1), out.shape[1], out.shape[2]))
        hw = out.shape[2]
    return maps


def edge_lengths_dense(sd: Mapping, output_image_size: int, z_start, z_end, batch_size: int = 512,
                       dtype=torch.float64, device: str = "cpu", with_conditioning: bool = False):
    """The same closed form as edge_lengths(norm_type="batch", training=True) with every layer written as a dense
    matrix product over whole stacks of chunks (the decoder sees a 1x1 latent image, so each transposed convolution
    is a small matrix): identical in exact arithmetic, fp64 rounding differs at 1e-16.  This is what makes the fp64
    check of EVERY BatchNorm chunk of a full-size run affordable (tests tie it to edge_lengths on sample chunks).
    with_conditioning=True also returns a (chunks, 2) array: column 0 = R, the maximum over (endpoint side, BatchNorm
    layer, channel) of |batch mean| / batch std of the layer's input (x - mean cancels log2(R) leading bits); column 1 =
    the smallest |pre-activation| entering a ReLU anywhere in the chunk.  A pre-activation within float32 rounding reach
    of zero makes the ReLU mask of that sample implementation-dependent, and in train mode the flipped tangent enters
    the NEXT layer's batch means of t and of xhat*t, i.e. it moves every edge of the chunk, not only its own."""
    global _DEVICE
    prev, _DEVICE = _DEVICE, device
    try:
        with torch.no_grad():
            zs = torch.as_tensor(z_start).to(device, dtype)
            ze = torch.as_tensor(z_end).to(device, dtype)
            E = zs.shape[0]
            w_in = _t(sd, "conv_in.weight", dtype).view(-1, zs.shape[1])
            b_in = _t(sd, "conv_in.bias", dtype)
            maps = _dense_maps(sd, output_image_size, dtype)
            out = torch.empty(E, dtype=dtype, device=device)
            full = (E // batch_size) * batch_size
            stack = 64 * batch_size                                  # chunks processed together

            cond = []

            def run(zs_, ze_, nb, bs):                               # nb chunks of bs edges
                delta = ze_ - zs_
                res = 0.0
                worst = torch.zeros(nb, dtype=dtype, device=device)
                near0 = torch.full((nb,), float("inf"), dtype=dtype, device=device)
                for z0 in (zs_, ze_):
                    x = z0 @ w_in.t() + b_in                         # (nb*bs, c0)
                    t = delta @ w_in.t()
                    for li, (norm, bias) in enumerate(((1, 0), (4, 3), (None, 6))):
                        M, cout, hw = maps[li]
                        b = _t(sd, f"deconv_layers.{bias}.bias", dtype)
                        x = (x @ M).view(nb, bs, cout, hw * hw) + b.view(1, 1, -1, 1)
                        t = (t @ M).view(nb, bs, cout, hw * hw)
                        if norm is None:
                            break
                        gamma = _t(sd, f"deconv_layers.{norm}.weight", dtype).view(1, 1, -1, 1)
                        beta = _t(sd, f"deconv_layers.{norm}.bias", dtype).view(1, 1, -1, 1)
                        mu = x.mean(dim=(1, 3), keepdim=True)
                        var = ((x - mu) ** 2).mean(dim=(1, 3), keepdim=True)
                        inv = 1.0 / torch.sqrt(var + BN_EPS)
                        worst = torch.maximum(worst, (mu.abs() * inv).amax(dim=(1, 2, 3)))
                        xhat = (x - mu) * inv
                        that = inv * (t - t.mean(dim=(1, 3), keepdim=True) - xhat * (xhat * t).mean(dim=(1, 3), keepdim=True))
                        x = xhat * gamma + beta
                        near0 = torch.minimum(near0, x.abs().amin(dim=(1, 2, 3)))
                        t = (that * gamma) * (x > 0).to(dtype)
                        x = torch.relu(x)
                        x, t = x.reshape(nb * bs, -1), t.reshape(nb * bs, -1)
                    sg = torch.sigmoid(x)
                    res = res + 0.5 * torch.linalg.vector_norm((t * sg * (1.0 - sg)).reshape(nb * bs, -1), dim=1)
                cond.append(torch.stack([worst, near0], dim=1))
                return res

            for lo in range(0, full, stack):
                hi = min(lo + stack, full)
                out[lo:hi] = run(zs[lo:hi], ze[lo:hi], (hi - lo) // batch_size, batch_size)
            if full < E:
                out[full:] = run(zs[full:], ze[full:], 1, E - full)
            if with_conditioning:
                return out.to("cpu", torch.float32), torch.cat(cond).to("cpu", torch.float64)
            return out.to("cpu", torch.float32)
    finally:
        _DEVICE = prev


def make_decoder_state(seed: int, latent_dim: int, out_channels: int, channels=(256, 128, 64),
                       norm_type: str = "batch") -> dict:
    """Seeded synthetic SpatialDecoder weights (golden fixture G3): numpy RandomState, so the same
    state_dict can be regenerated on the GPU box without shipping megabytes of weights."""
    r = np.random.RandomState(seed)
    c0, c1, c2 = channels

    def u(shape, fan_in):
        b = 1.0 / np.sqrt(fan_in)
        return r.uniform(-b, b, size=shape).astype(np.float32)

    sd = {
        "conv_in.weight": u((c0, latent_dim, 1, 1), latent_dim),
        "conv_in.bias": u((c0,), latent_dim),
        "deconv_layers.0.weight": u((c0, c1, 4, 4), c1 * 16),
        "deconv_layers.0.bias": u((c1,), c1 * 16),
        "deconv_layers.3.weight": u((c1, c2, 4, 4), c2 * 16),
        "deconv_layers.3.bias": u((c2,), c2 * 16),
        "deconv_layers.6.weight": u((c2, out_channels, 4, 4), out_channels * 16),
        "deconv_layers.6.bias": u((out_channels,), out_channels * 16),
    }
    if norm_type in ("batch", "group"):
        for idx, c in ((1, c1), (4, c2)):
            sd[f"deconv_layers.{idx}.weight"] = r.uniform(0.5, 1.5, size=c).astype(np.float32)
            sd[f"deconv_layers.{idx}.bias"] = r.uniform(-0.3, 0.3, size=c).astype(np.float32)
            if norm_type == "batch":
                sd[f"deconv_layers.{idx}.running_mean"] = r.uniform(-0.2, 0.2, size=c).astype(np.float32)
                sd[f"deconv_layers.{idx}.running_var"] = r.uniform(0.5, 1.5, size=c).astype(np.float32)
                sd[f"deconv_layers.{idx}.num_batches_tracked"] = np.array(0, dtype=np.int64)
    return sd

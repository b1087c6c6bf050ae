"""GPU side of the rows either side of the hot path: latent production on the MI355X feeding the codebook
builder without a host round trip, and the Transformer prior trained on the GPU -- single process against the
reference's loss curve, and two ranks sharing the box's GPU over gloo against the single-process run."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest
import torch

from test_prior_and_encoder import PRIOR_CFG, run_training, write_prior_inputs

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_encoder_on_gpu_and_resident_hand_over_to_the_codebook(golden, tmp_path):
    from oracle import synthetic as syn
    from vqvae_amd._device import device
    from vqvae_amd.scripts.build_codebook import build_codebook_device
    from vqvae_amd.spatial_vae import SpatialVAE
    from vqvae_amd.utils.spatial_latents import (encode_latents_device, flatten_latents_device, latents_to_codebook_device,
                                                 save_spatial_latents)
    dev = device()
    g = golden("encoder")
    for name, (cin, size, d, norm) in {"fm": (1, 28, 16, "batch"), "cf": (3, 32, 32, "group")}.items():
        vae = SpatialVAE(cin, [64, 128, 256], [256, 128, 64], d, "mse", size, norm, mse_use_sigmoid=True)
        vae.load_state_dict(syn.seeded_state_dict(vae.state_dict(), 5))
        vae = vae.to(dev)
        x = torch.from_numpy(np.random.RandomState(6).rand(24, cin, size, size).astype(np.float32))
        loader = [(x[:16], torch.arange(16)), (x[16:], torch.arange(16, 24))]
        z, mu, logvar, y = encode_latents_device(vae, loader, dev)
        assert z.is_cuda and z.shape == (24, d, 4, 4) and y.tolist() == list(range(24))
        np.testing.assert_allclose(mu.cpu().numpy(), g[f"{name}/mu"], rtol=2e-4, atol=2e-5)
        np.testing.assert_allclose(logvar.cpu().numpy(), g[f"{name}/logvar"], rtol=2e-4, atol=2e-5)
    # the reference's four files, then: codebook from the resident latents == codebook from z.pt read back
    vae = SpatialVAE(1, [64, 128, 256], [256, 128, 64], 16, "mse", 28, "batch", mse_use_sigmoid=True)
    vae.load_state_dict(syn.seeded_state_dict(vae.state_dict(), 5))
    vae = vae.to(dev)
    x = torch.from_numpy(np.random.RandomState(8).rand(64, 1, 28, 28).astype(np.float32))
    loader = [(x[i:i + 32], torch.zeros(32, dtype=torch.int64)) for i in (0, 32)]
    torch.manual_seed(3)
    save_spatial_latents(vae, loader, dev, tmp_path / "lat")
    z_file = torch.load(tmp_path / "lat" / "z.pt")
    assert z_file.shape == (64, 16, 4, 4) and not z_file.is_cuda
    assert all((tmp_path / "lat" / f).exists() for f in ("mu.pt", "logvar.pt", "y.pt"))
    torch.manual_seed(3)                                        # same eps -> same z as in the files
    res, y = latents_to_codebook_device(vae, loader, dev, k=8, sym="union", K=16, init="kpp", seed=42, batch_size=512)
    ref = build_codebook_device(flatten_latents_device(z_file.to(dev)), vae.decoder.train(), k=8, sym="union", K=16,
                                init="kpp", seed=42, batch_size=512)
    np.testing.assert_array_equal(res["medoids"], ref["medoids"])
    np.testing.assert_array_equal(res["assign_flat"], ref["assign_flat"])
    assert res["qe"] == ref["qe"] and res["latent_shape"] == (64, 16, 4, 4)


def test_prior_training_on_gpu_follows_the_reference_curve(golden, tmp_path):
    g = golden("prior")
    write_prior_inputs(str(tmp_path))
    hist, norms = run_training(str(tmp_path), torch.device("cuda", 0))
    np.testing.assert_allclose(hist["train_loss"], g["train/step_losses"], rtol=2e-4)
    np.testing.assert_allclose(hist["val_loss"], g["train/val_losses"], rtol=2e-4)
    np.testing.assert_allclose(norms, g["train/param_norms"], rtol=2e-4)


def test_two_ranks_on_the_gpu_equal_single_process(tmp_path):
    write_prior_inputs(str(tmp_path))
    port = socket.socket()
    port.bind(("127.0.0.1", 0))
    p = port.getsockname()[1]
    port.close()
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE="2", LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(p), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "_gpu_prior_rank.py"), str(tmp_path)], env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    outs = [q.communicate(timeout=400)[0] for q in procs]
    assert all(q.returncode == 0 for q in procs), "\n".join(outs)[-3000:]
    hist, norms = run_training(str(tmp_path), torch.device("cuda", 0), epochs=2)
    for rank in range(2):
        got = np.load(os.path.join(str(tmp_path), f"gpu_dp{rank}.npz"))
        np.testing.assert_allclose(got["train"], hist["train_loss"], rtol=2e-5)
        np.testing.assert_allclose(got["norms"], norms, rtol=2e-5)


def _torch_attention(qkv, n_head, keep, p):
    B, T, C3 = qkv.shape
    C = C3 // 3
    q, k, v = qkv.view(B, T, 3, n_head, C // n_head).permute(2, 0, 3, 1, 4)
    w = (q @ k.transpose(-2, -1)) * (1.0 / np.sqrt(C // n_head))
    w = w.masked_fill(torch.triu(torch.ones(T, T, device=qkv.device), 1).bool(), float("-inf"))
    w = torch.softmax(w, dim=-1)
    if keep is not None:
        w = w * keep / (1.0 - p)
    return (w @ v).transpose(1, 2).reshape(B, T, C)


@pytest.mark.parametrize("C,H,T,B", [(256, 4, 15, 37), (64, 4, 15, 64), (128, 4, 16, 5), (64, 4, 7, 3)])
def test_fused_attention_kernels_equal_torch_forward_and_backward(C, H, T, B):
    """csrc/prior.hip's attention (transformer.py:121-129 in one kernel, and its whole backward in another) against the
    same arithmetic written with torch ops in float64: output and the gradient with respect to qkv, with and without a
    fixed dropout mask (head_dim 64 / 16 / 32, ragged wave counts, T = 16 and short sequences)."""
    from vqvae_amd.prior import native
    dev = torch.device("cuda", 0)
    g = torch.Generator(device="cpu").manual_seed(C + T)
    qkv = torch.randn(B, T, 3 * C, generator=g).to(dev)
    dout = torch.randn(B, T, C, generator=g).to(dev)
    for p in (0.0, 0.25):
        keep = (torch.rand(B, H, T, T, generator=g) >= p).to(dev) if p > 0 else None
        x = qkv.clone().requires_grad_(True)
        out = native.causal_attention(x, H, p, keep)
        out.backward(dout)
        x64 = qkv.double().clone().requires_grad_(True)
        ref = _torch_attention(x64, H, keep.double() if keep is not None else None, p)
        ref.backward(dout.double())
        np.testing.assert_allclose(out.detach().cpu().numpy(), ref.detach().cpu().numpy(), rtol=2e-5, atol=2e-6)
        np.testing.assert_allclose(x.grad.cpu().numpy(), x64.grad.cpu().numpy(), rtol=2e-4, atol=2e-5)


def test_arena_adamw_equals_torch_adamw():
    """One-launch AdamW over the flat arena against torch.optim.AdamW on the same gradients, 40 steps with the learning rate
    moved twice (as the cosine schedule does per epoch)."""
    from vqvae_amd.prior.native import ArenaAdamW
    dev = torch.device("cuda", 0)
    g = torch.Generator().manual_seed(3)
    p0 = torch.randn(100003, generator=g)
    a = torch.nn.Parameter(p0.clone().to(dev))
    b = torch.nn.Parameter(p0.clone().to(dev))
    mine, ref = ArenaAdamW(a, lr=3e-4, weight_decay=0.01), torch.optim.AdamW([b], lr=3e-4, weight_decay=0.01)
    for step in range(40):
        grad = (torch.randn(100003, generator=g) * (1.0 + step % 3)).to(dev)
        a.grad, b.grad = grad.clone(), grad.clone()
        if step in (15, 30):
            for opt in (mine, ref):
                opt.param_groups[0]["lr"] *= 0.5
        mine.step()
        ref.step()
    np.testing.assert_allclose(a.detach().cpu().numpy(), b.detach().cpu().numpy(), rtol=2e-6, atol=1e-7)


def test_native_training_step_equals_eager_torch_step(tmp_path):
    """HIP-graph forward/backward + fused attention + ArenaAdamW against the same loop on stock torch ops (native=False)."""
    from vqvae_amd.prior.codes_dataset import get_code_loaders
    from vqvae_amd.prior.train import train_prior
    from vqvae_amd.prior.transformer import Transformer
    from vqvae_amd.scripts.train_transformer import set_seed
    write_prior_inputs(str(tmp_path))
    dev = torch.device("cuda", 0)
    runs = {}
    for native in (True, False):
        set_seed(42)
        tl, vl = get_code_loaders(os.path.join(str(tmp_path), "codes.npy"), os.path.join(str(tmp_path), "y.pt"), batch_size=64,
                                  num_workers=0, pin_memory=False, device=dev)
        model = Transformer(**PRIOR_CFG).to(dev)
        model.fused_attention = native
        runs[native] = train_prior(model, tl, vl, epochs=2, lr=3e-4, weight_decay=0.01, device=dev, native=native)
    np.testing.assert_allclose(runs[True]["train_loss"], runs[False]["train_loss"], rtol=2e-5)
    np.testing.assert_allclose(runs[True]["val_loss"], runs[False]["val_loss"], rtol=2e-5)

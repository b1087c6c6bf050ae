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

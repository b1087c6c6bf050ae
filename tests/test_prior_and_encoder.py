"""The rows either side of the hot path: latent production (SpatialEncoder) and the Transformer prior over
codes with its data-parallel training loop, against fixtures generated from the reference import
(tests/golden/encoder.npz, prior.npz; oracle/gen_golden_c2.py --part encoder|prior).  CPU: model parity, the
single-process loss curve, and world-2 / world-3 gloo runs that must reproduce the single-process run."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PRIOR_CFG = dict(num_classes=10, num_tokens=64, embed_dim=64, n_layers=2, n_head=4, max_seq_len=16, dropout=0.0)


def prior_inputs():
    r = np.random.RandomState(7)
    codes = r.randint(0, 64, size=(512, 4, 4)).astype(np.int32)
    codes[5, 1, 2] = codes[77, 0, 0] = codes[300, 3, 3] = -1
    return codes, torch.from_numpy(r.randint(0, 10, size=512).astype(np.int64))


def write_prior_inputs(tmp):
    codes, labels = prior_inputs()
    np.save(os.path.join(tmp, "codes.npy"), codes)
    torch.save(labels, os.path.join(tmp, "y.pt"))
    return os.path.join(tmp, "codes.npy"), os.path.join(tmp, "y.pt")


def run_training(tmp, device, epochs=3, group=None):
    """The CLI's sequence (seed, loaders, model, loop) on the fixture's inputs."""
    from vqvae_amd.prior.codes_dataset import get_code_loaders
    from vqvae_amd.prior.train import train_prior
    from vqvae_amd.prior.transformer import Transformer
    from vqvae_amd.scripts.train_transformer import set_seed
    codes_path, labels_path = os.path.join(tmp, "codes.npy"), os.path.join(tmp, "y.pt")
    set_seed(42)
    tl, vl = get_code_loaders(codes_path, labels_path, batch_size=64, num_workers=0, pin_memory=False)
    model = Transformer(**PRIOR_CFG).to(device)
    hist = train_prior(model, tl, vl, epochs=epochs, lr=3e-4, weight_decay=0.01, device=device, group=group)
    sd = model.state_dict()
    norms = np.array([float(sd[k].float().norm()) for k in sorted(sd) if sd[k].dtype.is_floating_point])
    return hist, norms


def test_encoder_outputs_equal_reference(golden):
    from oracle import synthetic as syn
    from vqvae_amd.spatial_vae import SpatialVAE
    g = golden("encoder")
    for name, (cin, size, d, norm) in {"fm": (1, 28, 16, "batch"), "cf": (3, 32, 32, "group")}.items():
        vae = SpatialVAE(cin, [64, 128, 256], [256, 128, 64], d, "mse", size, norm, mse_use_sigmoid=True)
        vae.load_state_dict(syn.seeded_state_dict(vae.state_dict(), 5))         # same keys as the reference's SpatialVAE
        x = torch.from_numpy(np.random.RandomState(6).rand(24, cin, size, size).astype(np.float32))
        vae.eval()
        with torch.no_grad():
            mu, logvar = vae.encoder(x)
        np.testing.assert_allclose(mu.numpy(), g[f"{name}/mu"], rtol=1e-5, atol=1e-6)
        np.testing.assert_allclose(logvar.numpy(), g[f"{name}/logvar"], rtol=1e-5, atol=1e-6)


def test_prior_forward_and_dataset_contract(golden, tmp_path):
    from oracle import synthetic as syn
    from vqvae_amd.prior.codes_dataset import CodesDataset, VanillaCodesDataset, get_code_loaders
    from vqvae_amd.prior.transformer import Transformer
    g = golden("prior")
    model = Transformer(**PRIOR_CFG)
    model.load_state_dict(syn.seeded_state_dict(model.state_dict(), 11))       # strict: the reference's names and buffers
    model.eval()
    r = np.random.RandomState(12)
    idx = torch.from_numpy(r.randint(0, 64, size=(8, 15)).astype(np.int64))
    cls = torch.from_numpy(r.randint(0, 10, size=8).astype(np.int64))
    with torch.no_grad():
        np.testing.assert_allclose(model(idx, y=cls).numpy(), g["forward/logits"], rtol=1e-5, atol=1e-6)
    codes_path, labels_path = write_prior_inputs(str(tmp_path))
    ds = CodesDataset(codes_path, labels_path)
    assert len(ds) == 509 and ds.seq_len == 16                         # images with a -1 are dropped (codes_dataset.py:15-17)
    x, y, lab = ds[0]
    assert x.shape == (15,) and y.shape == (15,) and x.dtype == torch.int64 and torch.equal(x[1:], y[:-1])
    flat = np.where(prior_inputs()[0].reshape(512, -1)[:, 0] >= 0, prior_inputs()[0].reshape(512, -1)[:, 0], 0)
    np.save(tmp_path / "flat.npy", flat.astype(np.int32))
    vd = VanillaCodesDataset(str(tmp_path / "flat.npy"), None, num_tokens=65)
    assert vd[3][0].tolist() == [64] and vd.seq_len == 2
    with pytest.raises(ValueError):
        get_code_loaders(str(tmp_path / "flat.npy"), vanilla_vae=True)


def test_prior_training_curve_equals_reference(golden, tmp_path):
    g = golden("prior")
    write_prior_inputs(str(tmp_path))
    hist, norms = run_training(str(tmp_path), torch.device("cpu"))
    np.testing.assert_allclose(hist["train_loss"], g["train/step_losses"], rtol=2e-6)
    np.testing.assert_allclose(hist["val_loss"], g["train/val_losses"], rtol=2e-6)
    np.testing.assert_allclose(norms, g["train/param_norms"], rtol=2e-6)


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _dp_worker(rank, world, port, tmp):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch.distributed as dist
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.set_num_threads(2)
        hist, norms = run_training(tmp, torch.device("cpu"), epochs=2)
        np.savez(os.path.join(tmp, f"dp{rank}.npz"), train=np.array(hist["train_loss"]), val=np.array(hist["val_loss"]), norms=norms)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_data_parallel_run_equals_single_process(tmp_path, world):
    """Ranks take slices of the same global batches and all-reduce one flat gradient buffer: every rank's loss curve and
    final weights equal the single-process run (float summation order aside)."""
    write_prior_inputs(str(tmp_path))
    mp.spawn(_dp_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    torch.set_num_threads(2)
    hist, norms = run_training(str(tmp_path), torch.device("cpu"), epochs=2)
    for rank in range(world):
        got = np.load(os.path.join(str(tmp_path), f"dp{rank}.npz"))
        np.testing.assert_allclose(got["train"], hist["train_loss"], rtol=1e-5)
        np.testing.assert_allclose(got["val"], hist["val_loss"], rtol=1e-5)
        np.testing.assert_allclose(got["norms"], norms, rtol=1e-5)


def _dp_dropout_worker(rank, world, port, tmp):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    from vqvae_amd.prior.codes_dataset import get_code_loaders
    from vqvae_amd.prior.train import seed_dropout_stream, train_prior
    from vqvae_amd.prior.transformer import Transformer
    from vqvae_amd.scripts.train_transformer import set_seed
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.set_num_threads(2)
        set_seed(42)
        shuffler = torch.Generator()
        shuffler.manual_seed(42)
        tl, vl = get_code_loaders(os.path.join(tmp, "codes.npy"), os.path.join(tmp, "y.pt"), batch_size=96, num_workers=0,
                                  pin_memory=False, generator=shuffler)
        model = Transformer(**dict(PRIOR_CFG, dropout=0.2))
        seed_dropout_stream(42, rank, torch.device("cpu"))
        seen = []
        orig_iter = tl._order

        def spy():
            o = orig_iter()
            seen.append(o[:8].tolist())
            return o
        tl._order = spy
        hist = train_prior(model, tl, vl, epochs=2, lr=3e-4, weight_decay=0.01, device=torch.device("cpu"))
        first_mask = torch.rand(4).tolist()                                   # the rank's own stream continues differently
        np.savez(os.path.join(tmp, f"drop{rank}.npz"), train=np.array(hist["train_loss"]), arena=model.arena.detach().numpy(),
                 order=np.array(seen), probe=np.array(first_mask))
    finally:
        dist.destroy_process_group()


def test_data_parallel_with_dropout_keeps_the_ranks_in_step(tmp_path):
    """dropout > 0 under data parallelism (advisor, round 2): the shuffling has its own generator, seeded identically on
    every rank, so the ranks walk the same batches whatever their dropout streams do; each rank draws its masks from its own
    stream; the all-reduced gradients keep the weights identical on all ranks."""
    write_prior_inputs(str(tmp_path))
    mp.spawn(_dp_dropout_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    a, b = (np.load(os.path.join(str(tmp_path), f"drop{r}.npz")) for r in range(2))
    np.testing.assert_array_equal(a["order"], b["order"])                     # same batches on both ranks, both epochs
    np.testing.assert_array_equal(a["arena"], b["arena"])                     # identical weights after every all-reduce
    np.testing.assert_array_equal(a["train"], b["train"])
    assert np.isfinite(a["train"]).all() and a["train"][-1] < a["train"][0]
    assert not np.array_equal(a["probe"], b["probe"])                         # ... while the dropout streams differ

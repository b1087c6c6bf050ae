"""The sharded hot path on the HIP kernels with more than one rank: N fresh child processes, one rank each,
all on this box's GPU(s), gloo rendezvous on 127.0.0.1 (the driver's 8-GPU run uses the same code over RCCL).  Every
rank must end with exactly the single-process result."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _single_process(n=2048, K=64):
    from oracle import metric as om
    from oracle import synthetic as syn
    from vqvae_amd._device import device
    from vqvae_amd.scripts.build_codebook import build_codebook_device
    from vqvae_amd.spatial_decoder import SpatialDecoder
    dev = device()
    z_h = syn.gauss_latents(n, 16, 0)
    sd = om.make_decoder_state(0, 16, 1, norm_type="batch")
    dec = SpatialDecoder(1, (256, 128, 64), 16, 28, "batch")
    dec.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()})
    dec = dec.to(dev).train()
    res = build_codebook_device(torch.from_numpy(z_h).to(dev), dec, k=20, sym="union", K=K,
                                init="kpp", seed=42, batch_size=512)
    res["bn"] = {k: v.cpu().numpy() for k, v in dec.state_dict().items() if "running" in k or "tracked" in k}
    return z_h, res


def _backend(world):
    """RCCL ("nccl") when the box has a GPU per rank -- the driver's multi-GPU node --, else gloo with the ranks sharing
    the GPU(s) that are there (RCCL refuses two ranks on one device)."""
    return "nccl" if torch.cuda.device_count() >= world else "gloo"


@pytest.mark.parametrize("world,shape", [(2, (2048, 16, 64)), (3, (2048, 16, 64)), (2, (60000, 16, 512))],
                         ids=["w2-c1", "w3-c1", "w2-c2"])
def test_ranks_on_hip_kernels_equal_single_process(tmp_path, world, shape):
    port = _free_port()
    procs = []
    for rank in range(world):           # fresh interpreters: nothing that touched the GPU is forked or re-executed
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"),
                   GEO_TEST_BACKEND=_backend(world), GEO_TEST_SHAPE=",".join(str(v) for v in shape))
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "_gpu_rank_worker.py"), str(tmp_path)],
                                      env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    logs = []
    for p in procs:
        try:
            out, _ = p.communicate(timeout=420)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        logs.append(out)
    assert all(p.returncode == 0 for p in procs), "\n".join(logs)[-4000:]
    z_h, res = _single_process(shape[0], shape[2])
    G = res["W_lcc"]
    for rank in range(world):
        got = np.load(os.path.join(str(tmp_path), f"rank{rank}.npz"))
        assert got["sharded"].tolist() == [1, 1]                     # the ranks really split the kNN rows and the chunks
        np.testing.assert_array_equal(got["z"], z_h)
        np.testing.assert_array_equal(got["indptr"], G.indptr.cpu().numpy())
        np.testing.assert_array_equal(got["indices"], G.indices.cpu().numpy())
        np.testing.assert_array_equal(got["lengths"], res["edge_lengths"].cpu().numpy())   # chunk-aligned: same BN batches
        np.testing.assert_array_equal(got["data"], G.data.cpu().numpy())
        np.testing.assert_array_equal(got["medoids"], res["medoids"])
        np.testing.assert_array_equal(got["assign"], res["assign_flat"])
        assert float(got["qe"]) == res["qe"]
        np.testing.assert_array_equal(got["arg"], res["assign_flat"])            # sharded assignment == fused chain
        # BatchNorm buffers: identical on every rank, and the single-process values up to float32 re-association
        for key, want in res["bn"].items():
            np.testing.assert_array_equal(got["bn/" + key], np.load(os.path.join(str(tmp_path), "rank0.npz"))["bn/" + key])
            if "tracked" in key:
                assert int(got["bn/" + key]) == int(want)
            else:
                np.testing.assert_allclose(got["bn/" + key], want, rtol=2e-5, atol=1e-7)


def test_sharded_builds_in_flight_equal_single_process(tmp_path):
    """build_codebooks_pipelined under a process group: every build sharded over two ranks AND three builds in flight per
    rank (collectives in ticket order, parallel.CollectiveOrder) -- five latent sets of different sizes; every rank must
    return, for every set, exactly what one process returns for it."""
    from oracle import metric as om
    from oracle import synthetic as syn
    from vqvae_amd._device import device
    from vqvae_amd.scripts.build_codebook import build_codebook_device
    from vqvae_amd.spatial_decoder import SpatialDecoder
    world, n, K = 2, 2048, 64
    port = _free_port()
    procs = []
    for rank in range(world):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"),
                   GEO_TEST_BACKEND=_backend(world), GEO_TEST_SHAPE=f"{n},16,{K}", GEO_TEST_MODE="in_flight")
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "_gpu_rank_worker.py"), str(tmp_path)],
                                      env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    logs = []
    for p in procs:
        try:
            out, _ = p.communicate(timeout=420)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        logs.append(out)
    assert all(p.returncode == 0 for p in procs), "\n".join(logs)[-4000:]
    dev = device()
    sd = om.make_decoder_state(0, 16, 1, norm_type="batch")
    sizes = [n, n - 301, n // 2, n - 77, n // 2 + 13]
    for i, m in enumerate(sizes):
        dec = SpatialDecoder(1, (256, 128, 64), 16, 28, "batch")
        dec.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()})
        want = build_codebook_device(torch.from_numpy(syn.gauss_latents(m, 16, 10 + i)).to(dev), dec.to(dev).train(), k=20,
                                     sym="union", K=K, init="kpp", seed=42, batch_size=512)
        for rank in range(world):
            got = np.load(os.path.join(str(tmp_path), f"rank{rank}.npz"))
            assert got["sharded"].tolist() == [1]
            np.testing.assert_array_equal(got[f"{i}/medoids"], want["medoids"])
            np.testing.assert_array_equal(got[f"{i}/assign_flat"], want["assign_flat"])
            np.testing.assert_array_equal(got[f"{i}/lengths"], want["edge_lengths"].cpu().numpy())
            assert float(got[f"{i}/qe"]) == want["qe"]


def test_rccl_accepts_ticket_ordered_collectives_from_pipeline_threads():
    """What one GPU can say about the RCCL side of sharded builds in flight: a ONE-rank nccl (= RCCL) group, three host
    threads on their own HIP streams, their builds' all-gathers issued through CollectiveOrder on the one communicator --
    ProcessGroupNCCL must take the calls from several threads, and the gathers must be ordered with the kernels of the
    calling thread's stream (tests/_gpu_rccl_threads_worker.py).  The multi-rank order itself is tested over gloo."""
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()),
               HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "_gpu_rccl_threads_worker.py")], env=env,
                         capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and out.stdout.strip().endswith("OK"), (out.stdout[-1500:], out.stderr[-3000:])

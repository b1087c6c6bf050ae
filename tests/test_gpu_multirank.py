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


def _single_process():
    from oracle import metric as om
    from oracle import synthetic as syn
    from vqvae_amd._device import device
    from vqvae_amd.scripts.build_codebook import build_codebook_device
    from vqvae_amd.spatial_decoder import SpatialDecoder
    dev = device()
    z_h = syn.gauss_latents(2048, 16, 0)
    sd = om.make_decoder_state(0, 16, 1, norm_type="batch")
    dec = SpatialDecoder(1, (256, 128, 64), 16, 28, "batch")
    dec.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()})
    res = build_codebook_device(torch.from_numpy(z_h).to(dev), dec.to(dev).train(), k=20, sym="union", K=64,
                                init="kpp", seed=42, batch_size=512)
    return z_h, res


@pytest.mark.parametrize("world", [2, 3])
def test_ranks_on_hip_kernels_equal_single_process(tmp_path, world):
    port = _free_port()
    procs = []
    for rank in range(world):           # fresh interpreters: nothing that touched the GPU is forked or re-executed
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
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
    z_h, res = _single_process()
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

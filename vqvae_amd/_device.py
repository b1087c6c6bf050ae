"""Device plumbing shared by the vqvae_amd.geo wrappers: PyTorch-ROCm owns device memory and
streams, libgeo_hip.so does the arithmetic.  Nothing here computes on the CPU."""
import ctypes
import os
import threading
from dataclasses import dataclass
from typing import Optional

import numpy as np
import torch
from scipy import sparse

from . import _lib


def device() -> torch.device:
    """The GPU of this process (one process per GPU: LOCAL_RANK selects it).  Raises without one."""
    if not torch.cuda.is_available():
        raise _lib.GeoHipError("vqvae_amd needs an MI355X (torch.cuda.is_available() is False); "
                               "there is no CPU fallback for the geodesic-codebook path")
    return torch.device("cuda", int(os.environ.get("LOCAL_RANK", "0")) % torch.cuda.device_count())


def ptr(t: Optional[torch.Tensor]):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def stream_ptr():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


_ws_cache = {}
_ws_lock = threading.Lock()        # pipeline slots (host threads) look up / grow / release concurrently
_slot_streams = {}


def slot_streams(dev: torch.device, depth: int):
    """The `depth` streams builds in flight run on -- ONE pool per device for the life of the process.  torch hands out fresh
    `torch.cuda.Stream` objects round-robin from 32 handles per device and the workspace cache below is keyed by handle: a new
    set of streams per pipelined call would pin one grow-only buffer (25 GB for the JVP stage at 60 000 x 16) per handle ever
    seen (advisor, round 3).  Reusing the same streams bounds the cache at `depth` + 1 buffers."""
    key = (dev.type, dev.index)
    with _ws_lock:
        pool = _slot_streams.setdefault(key, [])
        while len(pool) < depth:
            pool.append(torch.cuda.Stream(device=dev))
        return pool[:depth]


def workspace(nbytes: int, dev: torch.device) -> torch.Tensor:
    """Grow-only scratch buffer per device and stream (288 GB of HBM: keep it resident between calls)."""
    nbytes = int(nbytes) + 256
    # one buffer per (device, stream): builds pipelined on two streams (bench.py) must not share scratch
    key = (dev.type, dev.index, torch.cuda.current_stream(dev).cuda_stream)
    with _ws_lock:
        buf = _ws_cache.get(key)
        if buf is None or buf.numel() < nbytes:
            _ws_cache.pop(key, None)
            buf = None
            buf = torch.empty(nbytes, dtype=torch.uint8, device=dev)
            _ws_cache[key] = buf
        return buf


def release_workspace(above_bytes: int = 0) -> None:
    """Hand scratch buffers larger than `above_bytes` back to torch's caching allocator (stream ordered: kernels
    already queued on the current stream keep their memory).  The pipeline itself never calls this: the JVP stage's
    scratch (25 GB at 60 000 x 16, batch 512) stays resident between builds -- measured, handing it back after the
    stage lets the smaller SSSP workspace split the block and the next build pays a fresh 25 GB allocation (+2.2 ms
    per 55 ms step).  A caller that builds one codebook and then needs the memory calls it afterwards."""
    with _ws_lock:
        for key in [k for k, b in _ws_cache.items() if b.numel() > above_bytes]:
            del _ws_cache[key]


def workspace_buffers() -> int:
    """Number of cached scratch buffers (tests: pipelined calls must not grow it without bound)."""
    with _ws_lock:
        return len(_ws_cache)


@dataclass
class DeviceCSR:
    """CSR on the GPU: int32 indptr [n+1], int32 indices [nnz], optional f32 data [nnz]."""
    n: int
    indptr: torch.Tensor
    indices: torch.Tensor
    data: Optional[torch.Tensor]

    @property
    def nnz(self) -> int:
        return int(self.indices.numel())

    @staticmethod
    def from_scipy(W: sparse.spmatrix, dev: torch.device, with_data: bool = True) -> "DeviceCSR":
        W = W.tocsr()
        if W.nnz >= 2 ** 31 or W.shape[0] >= 2 ** 31 - 1:
            raise ValueError("graph too large for int32 CSR")
        indptr = torch.from_numpy(np.ascontiguousarray(W.indptr, dtype=np.int32)).to(dev)
        indices = torch.from_numpy(np.ascontiguousarray(W.indices, dtype=np.int32)).to(dev)
        data = None
        if with_data:
            data = torch.from_numpy(np.ascontiguousarray(W.data, dtype=np.float32)).to(dev)
        return DeviceCSR(W.shape[0], indptr, indices, data)

    def to_scipy(self) -> sparse.csr_matrix:
        data = (self.data.cpu().numpy() if self.data is not None
                else np.ones(self.nnz, dtype=np.float32))
        return sparse.csr_matrix((data, self.indices.cpu().numpy(), self.indptr.cpu().numpy()),
                                 shape=(self.n, self.n))

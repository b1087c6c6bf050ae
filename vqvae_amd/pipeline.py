"""Several independent codebook builds in flight on ONE GPU.

40 % of a build is the k-means++ chain, a serial sequence of small solves that keeps one compute unit busy and 255 idle
(DESIGN.md section 4).  Builds of different latent sets do not depend on each other, so `run_pipelined` drives `depth` of them
at once: one host thread per slot, each inside its own HIP stream (`torch.cuda.stream`), hence with its own scratch workspace
(`_device.workspace` is keyed by stream) and its own outputs; the library's per-call state (error text, sweep profile) is per
host thread.  One build's chain then runs beside another build's kNN / JVP kernels.  Nothing is shared between slots but
read-only inputs -- give every slot its own decoder module when BatchNorm running statistics are updated in place.

Measured at the 60 000-latent configuration (bench.py): 52.4 ms per build one after the other, 39.7 ms with two in flight,
35.8 ms with three, 33.8 ms with four, 34.6 ms with six.  The latency of a single build does not improve (it grows by a few
per cent).
"""
import threading
from typing import Callable, List, Optional

import torch

from ._device import slot_streams


def run_pipelined(fn: Callable[[int, int], object], n_items: int, depth: int, device: torch.device,
                  streams: Optional[List[torch.cuda.Stream]] = None) -> list:
    """Calls fn(item, slot) for item = 0 .. n_items-1 from `depth` host threads, thread s inside stream s; items are handed
    out from a shared counter (a slot that finishes early takes the next one).  Returns [fn(0, .), fn(1, .), ...].
    Exceptions of a worker are re-raised here.  depth <= 1: plain loop on the current stream."""
    results: list = [None] * n_items
    if depth <= 1 or n_items <= 1:
        for i in range(n_items):
            results[i] = fn(i, 0)
        return results
    depth = min(depth, n_items)
    streams = streams or slot_streams(device, depth)             # one pool per device: bounded workspace cache
    lock, cursor, errors = threading.Lock(), [0], []
    ready = torch.cuda.Event()
    ready.record(torch.cuda.current_stream(device))              # work queued before the call is visible to every slot

    def worker(slot: int) -> None:
        try:
            with torch.cuda.stream(streams[slot]):
                streams[slot].wait_event(ready)
                while not errors:
                    with lock:
                        i = cursor[0]
                        cursor[0] += 1
                    if i >= n_items:
                        break
                    results[i] = fn(i, slot)
                streams[slot].synchronize()
        except BaseException as e:                               # noqa: BLE001 -- handed to the caller
            errors.append(e)

    threads = [threading.Thread(target=worker, args=(s,), name=f"geo-build-{s}") for s in range(depth)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    if errors:
        raise errors[0]
    return results

"""RCCL itself from the pipeline's host threads, as far as one GPU allows: a one-rank "nccl" process group, three host threads
each inside its own HIP stream, every thread's builds issuing their all-gathers through parallel.CollectiveOrder (one
communicator, ticket order).  Checks stream ordering (the gathered tensor is produced and consumed by kernels of the thread's
stream) and that ProcessGroupNCCL accepts calls from several threads.  Prints OK."""
import os
import sys
import threading

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    from vqvae_amd.parallel import STAGES, CollectiveOrder, OrderedGroup, _all_gather
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    n_builds, depth, n = 12, 3, 1 << 20
    order = CollectiveOrder(n_builds, depth)
    streams = [torch.cuda.Stream(device=dev) for _ in range(depth)]
    lock, cursor, errors, sums = threading.Lock(), [0], [], {}

    def slot(s):
        try:
            with torch.cuda.stream(streams[s]):
                while True:
                    with lock:
                        i = cursor[0]
                        cursor[0] += 1
                    if i >= n_builds:
                        break
                    grp = OrderedGroup(order, i)
                    acc = torch.zeros((), dtype=torch.float64, device=dev)
                    try:
                        for si, stage in enumerate(STAGES):
                            x = torch.full((n,), float(i * 10 + si), device=dev) * 2.0 + 1.0        # kernels of this stream
                            out = torch.empty(n, device=dev)
                            _all_gather(out, x, grp, stage)
                            acc += out.double().sum()                                               # consumed on this stream
                    finally:
                        order.finish(i)
                    sums[i] = acc
                streams[s].synchronize()
        except BaseException as e:                               # noqa: BLE001
            errors.append(repr(e))
            order.abort(e)

    threads = [threading.Thread(target=slot, args=(s,)) for s in range(depth)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(120)
    assert not any(t.is_alive() for t in threads), "a pipeline thread is parked"
    assert not errors, errors
    for i in range(n_builds):
        want = sum((float(i * 10 + si) * 2.0 + 1.0) * n for si in range(len(STAGES)))
        assert float(sums[i]) == want, (i, float(sums[i]), want)
    dist.barrier()
    dist.destroy_process_group()
    print("OK")


if __name__ == "__main__":
    main()

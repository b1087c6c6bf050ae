"""Several sharded builds in flight per rank: `parallel.CollectiveOrder` makes every rank issue the collectives of all
builds in one order on ONE communicator.  CPU test over gloo: 2 ranks x 3 builds in flight x 9 builds; every build's host
thread dawdles a random (rank- and build-dependent) time before each collective and skips the optional stages by a rule all
ranks share -- the gathers must still pair up (same build, same stage on both ranks) and nothing may hang."""
import os
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, n_builds, depth, q):
    import random
    import threading
    import time
    import torch.distributed as dist
    sys.path.insert(0, ROOT)
    from vqvae_amd.parallel import STAGES, CollectiveOrder, OrderedGroup, _all_gather
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    order = CollectiveOrder(n_builds, depth)
    results, errors = {}, []
    lock, cursor = threading.Lock(), [0]

    def build(i):
        rnd = random.Random(1000 * rank + i)                      # different dawdling on every rank
        grp = OrderedGroup(order, i)
        got = {}
        try:
            for si, stage in enumerate(STAGES):
                if stage in ("knn_d2", "bn_fold") and (i + si) % 3 == 0:     # skipped alike on all ranks
                    continue
                time.sleep(rnd.random() * (0.03 if stage.startswith("assign") else 0.01))
                inp = torch.tensor([float(1000 * i + 10 * si + rank)])
                out = torch.empty(world)
                _all_gather(out, inp, grp, stage)
                got[stage] = out.tolist()
        finally:
            order.finish(i)
        return got

    def slot():
        try:
            while True:
                with lock:
                    i = cursor[0]
                    cursor[0] += 1
                if i >= n_builds:
                    return
                results[i] = build(i)
        except BaseException as e:                                # noqa: BLE001
            errors.append(repr(e))
            order.abort(e)

    threads = [threading.Thread(target=slot) for _ in range(depth)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(120)
    hung = any(t.is_alive() for t in threads)
    q.put((rank, results, errors, hung))
    if not hung:
        dist.destroy_process_group()


@pytest.mark.parametrize("depth", [1, 3])
def test_builds_in_flight_issue_their_collectives_in_one_order(depth):
    import torch.multiprocessing as mp
    from vqvae_amd.parallel import STAGES
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    world, n_builds, port = 2, 9, 29500 + (os.getpid() + depth) % 400
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_builds, depth, q)) for r in range(world)]
    for p in procs:
        p.start()
    outs = [q.get(timeout=240) for _ in range(world)]
    for p in procs:
        p.join(30)
        if p.is_alive():
            p.kill()
    for rank, results, errors, hung in outs:
        assert not errors and not hung, (rank, errors, hung)
        assert sorted(results) == list(range(n_builds))
        for i, got in results.items():
            for si, stage in enumerate(STAGES):
                if stage in ("knn_d2", "bn_fold") and (i + si) % 3 == 0:
                    assert stage not in got
                else:                                               # both ranks' contributions of THIS build and stage
                    assert got[stage] == [float(1000 * i + 10 * si + r) for r in range(world)], (rank, i, stage, got[stage])


def test_ticket_order_puts_the_final_exchange_of_a_build_behind_the_early_ones_of_later_builds():
    from vqvae_amd.parallel import STAGES, CollectiveOrder
    o = CollectiveOrder(8, 4)
    early, late = STAGES.index("edge_lengths"), STAGES.index("assign_d")
    assert o._ticket(0, late) > o._ticket(3, early)                # build 3 starts beside build 0: its JVP exchange goes first
    assert o._ticket(0, late) < o._ticket(4, STAGES.index("latents"))   # build 4 starts only after build 0 has ended
    assert o._ticket(2, early) < o._ticket(2, late)
    tickets = sorted(o._ticket(b, s) for b in range(8) for s in range(len(STAGES)))
    assert len(set(tickets)) == len(tickets)


def test_a_failed_build_wakes_the_builds_waiting_behind_its_tickets():
    """Build 0 fails before its first collective; build 1 is parked behind build 0's early tickets (its own ticket is larger):
    `abort` must make it raise instead of waiting for a ticket that never comes; `finish` alone (a build that simply skipped
    everything) must let it through."""
    import threading
    import time
    from vqvae_amd.parallel import CollectiveOrder
    for how in ("abort", "finish"):
        o = CollectiveOrder(3, 2)
        out = {}

        def waiter():
            try:
                out["value"] = o.issue(1, "knn_idx", lambda: "issued")
            except RuntimeError as e:
                out["error"] = e

        t = threading.Thread(target=waiter)
        t.start()
        time.sleep(0.3)
        assert t.is_alive() and not out                        # parked: build 0 has not issued or given up anything
        if how == "abort":
            o.abort(ValueError("build 0 failed"))
        else:
            o.finish(0)
        t.join(5)
        assert not t.is_alive()
        if how == "abort":
            assert "error" in out and isinstance(out["error"].__cause__, ValueError)
        else:
            assert out.get("value") == "issued"

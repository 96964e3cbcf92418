"""CPU, world_size 2, gloo: the bucketed gradient exchange sums every element exactly once and respects
the frozen prefix, when driven in backward order (the N>1 path of bench.py / TrainStep)."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from multigriddet_amd.dp import GradBuckets
    sizes = [1000, 50, 4000, 7, 12000, 300, 90000, 64]
    offs = np.concatenate([[0], np.cumsum(sizes)[:-1]]).tolist()
    n = sum(sizes)
    ok = True
    for lo in (0, offs[3]):
        grads = torch.arange(n, dtype=torch.float32) * (rank + 1)
        gb = GradBuckets(grads, offs, world, bucket_mb=0.05)
        gb.reset(lo=lo)
        for i in range(len(sizes) - 1, -1, -1):       # backward order
            gb.on_layer_done(i)
        gb.finish()
        expect = torch.arange(n, dtype=torch.float32) * sum(r + 1 for r in range(world))
        expect[:lo] = torch.arange(lo, dtype=torch.float32) * (rank + 1)     # frozen prefix untouched
        ok = ok and bool(torch.equal(grads, expect))
        ok = ok and len(gb.buckets) > 2
    q.put((rank, ok))
    dist.destroy_process_group()


def test_bucketed_allreduce_world2():
    world = 2
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in ps:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in ps:
        p.join(timeout=60)
    assert all(ok for _, ok in res), res

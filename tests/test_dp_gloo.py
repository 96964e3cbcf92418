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


def _trainer_helpers_worker(rank, world, port, q):
    """The collectives MultiGridTrainer / MultiGridDetTrainModel use around the step (dp.py): equal shards, rank 0's
    state for everyone, mean of a scalar, rank 0's stop flag - and the generator's random streams under data
    parallelism (own permutation per rank, common multi-scale shapes)."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from multigriddet_amd.dp import all_reduce_mean_scalar, broadcast_flag, broadcast_tensors, shard_lines
    from multigriddet_amd.data.generators import MultiGridDataGenerator
    out = {}
    lines = [f"img{i}.png 1,2,3,4,0" for i in range(11)]
    mine = shard_lines(lines, rank, world)
    out["shard"] = mine
    params = torch.full((1000,), float(rank + 1))
    moving = torch.arange(10, dtype=torch.float32) * (rank + 1)
    broadcast_tensors([params, None, moving, torch.zeros(0)], world)
    out["params_ok"] = bool((params == 1.0).all()) and bool(torch.equal(moving, torch.arange(10, dtype=torch.float32)))
    out["mean"] = all_reduce_mean_scalar(3.0 * (rank + 1), world)
    out["flag"] = broadcast_flag(rank == 0, world)
    anchors = [np.ones((3, 2), np.float32)] * 3
    gen = MultiGridDataGenerator(mine, 2, (416, 416), anchors, 80, rescale_interval=1, seed=5 + rank, shape_seed=5,
                                 num_workers=1)
    out["shapes"] = [tuple(gen.next_shape()) for _ in range(12)]
    out["perm"] = gen.indexes.tolist()
    out["len"] = len(gen)
    q.put((rank, out))
    dist.destroy_process_group()


def test_trainer_data_parallel_helpers_world2():
    world = 2
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_trainer_helpers_worker, args=(r, world, port, q)) for r in range(world)]
    for p in ps:
        p.start()
    res = dict(q.get(timeout=120) for _ in range(world))
    for p in ps:
        p.join(timeout=60)
    a, b = res[0], res[1]
    assert len(a["shard"]) == len(b["shard"]) == 5 and not set(a["shard"]) & set(b["shard"])   # equal, disjoint shards
    assert a["len"] == b["len"]                                  # same steps per epoch -> same sequence of collectives
    assert a["params_ok"] and b["params_ok"]                     # rank 0's state everywhere
    assert a["mean"] == b["mean"] == 4.5
    assert a["flag"] is True and b["flag"] is True               # rank 0 decides
    assert a["shapes"] == b["shapes"] and len(set(a["shapes"])) > 1      # one resolution per step on every rank
    assert sorted(a["perm"]) == sorted(b["perm"]) == list(range(5))

"""Multi-GPU path: sequences shard with no data-path collective; covered on CPU with 2 gloo ranks."""
import os

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import chd_amd
from chd_amd import sharding


def test_lpt_assign_balances_and_covers():
    costs = [90, 600, 60, 90, 90, 120, 30, 90]
    parts = sharding.lpt_assign(costs, 3)
    assert sorted(i for p in parts for i in p) == list(range(len(costs)))
    loads = [sum(costs[i] for i in p) for p in parts]
    assert max(loads) == 600 and min(loads) >= 270
    assert sharding.lpt_assign(costs, 3) == parts          # deterministic


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    costs = [90] * 7 + [60]
    mine = sharding.my_shard(costs)
    recs = [(i, {'seq': i, 'rank': rank, 'value': float(i) * 2}) for i in mine]       # stand-in for per-sequence results
    merged = sharding.gather_records(recs, dist)
    t = torch.tensor([float(len(mine))]); dist.all_reduce(t)                           # the only collective bench.py needs: counters
    if rank == 0:
        q.put((sorted(merged.keys()), t.item(), [merged[i]['value'] for i in sorted(merged)]))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_shard_and_gather():
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    keys, total, vals = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert keys == list(range(8)) and total == 8.0 and vals == [2.0 * i for i in range(8)]

"""Multi-GPU sharding of independent sequences.

Every video is a self-contained NLP (``scripts/run_phys_mocap.py:80`` loops over directories with no shared
state), so the path shards with no data-path collective: one process per GPU, a static
longest-processing-time-first assignment by frame count, results returned through files or a final
``gather_object`` of small per-sequence records.
"""
import os


def lpt_assign(costs, n_shards):
    """Longest-processing-time-first: returns ``n_shards`` lists of item indices."""
    order = sorted(range(len(costs)), key=lambda i: (-costs[i], i))
    loads = [0.0] * n_shards
    out = [[] for _ in range(n_shards)]
    for i in order:
        k = min(range(n_shards), key=lambda s: (loads[s], s))
        out[k].append(i)
        loads[k] += costs[i]
    for lst in out:
        lst.sort()
    return out


def rank_world():
    return int(os.environ.get('RANK', '0')), int(os.environ.get('WORLD_SIZE', '1')), int(os.environ.get('LOCAL_RANK', '0'))


def my_shard(costs, rank=None, world=None):
    r, w, _ = rank_world()
    rank = r if rank is None else rank
    world = w if world is None else world
    return lpt_assign(costs, world)[rank]


def gather_records(records, dist=None):
    """All ranks contribute a list of (index, record); rank 0 gets the merged dict (others get None).
    ``dist`` is ``torch.distributed`` when running under torch.distributed.run, else None."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return dict(records)
    bucket = [None] * dist.get_world_size() if dist.get_rank() == 0 else None
    dist.gather_object(list(records), bucket, dst=0)
    if dist.get_rank() != 0:
        return None
    merged = {}
    for part in bucket:
        merged.update(dict(part))
    return merged

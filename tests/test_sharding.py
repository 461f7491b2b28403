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


# ---- shard determinism through the real driver: run_phys_mocap.main over 8 directories of mixed length, once as a
#      single process and once as two ranks; every directory's output files must be byte-identical.  The solver handle is
#      replaced by the host emulation of the kernel source (tests/host_emu) behind the same solve_dirs interface, because
#      this container has no GPU; sharding, directory handling and file I/O are the driver's own code.
class _EmuSolver:
    def __init__(self, device=0, config=None, **kw):
        self.cfg = config

    def solve_dirs(self, in_dirs, out_dirs, nframes):
        import sys
        sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), 'host_emu'))
        import emu
        from chd_amd import io_formats as iof
        from chd_amd.phys_capi import default_config
        st = []
        for din, dout, F in zip(in_dirs, out_dirs, nframes):
            seq = iof.read_inputs(din, F)
            e = emu.EmuProblem(seq, default_config(max_iter=[15] * 6))
            e.solve(0, 1)                                       # the two kinematic stages are enough to produce sol_out_no_dynamics
            stats, snaps = e.results()
            for k, name in enumerate(('sol_out_no_dynamics.txt', 'sol_out_dynamics.txt', 'sol_out_durations.txt')):
                s = snaps[0]
                iof.write_solution(iof.Solution(dt=seq.dt, num_frames=s['num_frames'], base_lin=s['base_lin'], base_ang_deg=s['base_ang_deg'],
                                                ee_pos=s['ee_pos'], ee_force=s['ee_force'], contact=s['contact']), os.path.join(dout, name))
            open(os.path.join(dout, 'success_log.txt'), 'w').write('dynamics 0\ndurations 0\n')
            st.append(0)
        return st

    def close(self):
        pass


def _driver_rank(rank, world, port, root, q):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    if world > 1:
        dist.init_process_group('gloo', rank=rank, world_size=world)
    from chd_amd import run_phys_mocap
    run_phys_mocap.PhysOptim = _EmuSolver
    rc = run_phys_mocap.main(['--data', root, '--character', 'ybot'])
    n = torch.tensor([float(len(sharding.my_shard([1] * 8)))])
    if world > 1:
        dist.all_reduce(n)
        dist.barrier()
        dist.destroy_process_group()
    q.put((rank, rc, n.item()))


def _make_tree(root):
    from chd_amd import io_formats as iof
    from chd_amd.synth import make_walk
    frames = [24, 40, 24, 32, 40, 24, 32, 28]
    for i, F in enumerate(frames):
        d = os.path.join(root, 'video_%02d' % i, 'phys_optim_in_ybot')
        iof.write_inputs(make_walk(seed=10 + i, F=F, randomize=True), d)
    return frames


def test_driver_outputs_do_not_depend_on_the_number_of_ranks(tmp_path):
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), 'host_emu'))
    import emu
    emu.build()
    roots = {}
    ctx = mp.get_context('spawn')
    for world in (1, 2, 8):          # (8: the driver's largest configuration -- one directory per rank here)
        root = str(tmp_path / ('w%d' % world)); os.makedirs(root)
        _make_tree(root)
        roots[world] = root
        q = ctx.Queue()
        port = 29500 + ((os.getpid() + 17 * world) % 2000)
        procs = [ctx.Process(target=_driver_rank, args=(r, world, port, root, q)) for r in range(world)]
        for p in procs:
            p.start()
        got = [q.get(timeout=600) for _ in range(world)]
        for p in procs:
            p.join(timeout=60)
            assert p.exitcode == 0
        assert all(g[1] == 0 for g in got) and got[0][2] == 8.0           # every rank succeeded; the shards cover all 8 directories
    names = ['sol_out_durations.txt', 'sol_out_dynamics.txt', 'sol_out_no_dynamics.txt', 'success_log.txt']
    for i in range(8):
        a = os.path.join(roots[1], 'video_%02d' % i, 'phys_optim_out_ybot')
        for world in (2, 8):
            b = os.path.join(roots[world], 'video_%02d' % i, 'phys_optim_out_ybot')
            assert sorted(os.listdir(a)) == names and sorted(os.listdir(b)) == names          # every directory written exactly once, by whichever rank owns it
            for n in names:
                assert open(os.path.join(a, n), 'rb').read() == open(os.path.join(b, n), 'rb').read(), (world, i, n)

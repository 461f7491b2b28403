"""bench.py's multi-rank path without hardware: `bench.main` as TWO ranks on the gloo backend, the solver handle replaced by the host
emulation of the kernel source (tests/host_emu) behind the same upload / solve / fetch interface.  What runs is bench.py's own code:
process-group set-up from the launcher's environment, barrier, the timed region, the max / sum all-reduces, rank 0's single JSON line
with the contract's keys and whole-job aggregates.  (On a GPU box the backend is nccl = RCCL; no collective touches the data path.)"""
import io
import json
import os
import sys
import contextlib

import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


class _EmuBatch:
    def __init__(self, seqs):
        self.seqs = list(seqs)
        self.res = None

    def solve(self):
        import time
        import types
        sys.path.insert(0, os.path.join(HERE, 'host_emu'))
        import emu
        from chd_amd.phys_capi import default_config
        t0 = time.perf_counter()
        self.res = []
        iters = 0
        for seq in self.seqs:
            e = emu.EmuProblem(seq, default_config(max_iter=[15] * 6))
            e.solve(0, 1)
            stats, snaps = e.results()
            sz = e.sizes(1)
            it = int(stats[0, 1] + stats[1, 1]); iters += it
            self.res.append(types.SimpleNamespace(total_iters=it, dynamics_succeed=False, durations_succeed=False,
                                                  sizes=dict(n=sz['n'], m=sz['m'], kkt_dim=sz['n'] + sz['m'], halfband=sz['w'], border=sz['bc'], nnz_jac=sz['nnz_jac'])))
        ms = 1e3 * (time.perf_counter() - t0)
        return dict(kernel_ms=[ms, 0.0], host_ms=0.0, total_iters=iters, total_factorizations=iters, alg_bytes=1e6 * iters, n_fallback=0,
                    phase_ms=[ms] * 24, max_seq_ms=ms, n_stalled=0, n_rejected=0, n_workgroups=1)

    def fetch(self):
        return self.res

    def free(self):
        pass


class _EmuSolver:
    def __init__(self, device=0, config=None):
        self.cfg = config

    def upload(self, seqs):
        return _EmuBatch(seqs)

    def close(self):
        pass


def _rank(rank, world, port, q):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    sys.path.insert(0, ROOT)
    import bench
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        bench.main(['--gpus', str(world), '--steps', '1', '--warmup', '1', '--batch', '2', '--frames', '24', '--no-cpu-baseline', '--no-side-metrics', '--strong-total', '0'],
                   solver_factory=_EmuSolver)
    q.put((rank, buf.getvalue()))


def test_bench_main_as_two_gloo_ranks():
    sys.path.insert(0, os.path.join(HERE, 'host_emu'))
    import emu
    emu.build()
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = 29500 + ((os.getpid() + 911) % 2000)
    procs = [ctx.Process(target=_rank, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=600) for _ in range(2))
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert got[1].strip() == ''                                   # only rank 0 prints
    lines = [ln for ln in got[0].splitlines() if ln.strip()]
    assert len(lines) == 1
    o = json.loads(lines[0])
    for k in ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling', 'vs_baseline', 'dtype', 'data', 'config', 'roofline'):
        assert k in o
    assert o['n_gpus'] == 2 and o['steps'] == 1 and o['scaling'] == 'weak' and o['dtype'] == 'f64'
    assert o['config']['sequences_per_gpu'] == 2
    # whole-job aggregates over both ranks: 4 sequences in the timed region, iterations summed by the all-reduce
    assert abs(o['value'] * o['ms_per_step'] * 1e-3 - 4.0) < 1e-6
    assert o['config']['ipm_iterations_per_sequence'] > 0
    assert o['roofline']['peak'] == 16000.0
    assert 'cpu_baseline' not in o                                 # N > 1: rank 0 does not time the CPU baseline


def _check_line(o, n):
    for k in ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling', 'vs_baseline', 'dtype', 'data', 'config', 'roofline'):
        assert k in o
    assert o['n_gpus'] == n and o['scaling'] == 'weak' and o['roofline']['peak'] == 8000.0 * n
    assert abs(o['value'] * o['ms_per_step'] * 1e-3 - 2.0 * n) < 1e-6          # batch 2, one step, per rank


def test_bench_launches_its_own_ranks_from_a_plain_shell(capfd, monkeypatch):
    """`python bench.py --gpus 2` WITHOUT a launcher around it (no RANK / WORLD_SIZE in the environment -- VERDICT r04: until round 5 `--gpus` was parsed and
    never used, and this ran one rank printing n_gpus 1): bench.main starts the two ranks itself, rank 0 prints the line, n_gpus comes from the live process
    group, and the strong-scaling leg (a fixed total, LPT-sharded over the ranks) rides along."""
    sys.path.insert(0, os.path.join(HERE, 'host_emu'))
    import emu
    emu.build()
    for k in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT'):
        monkeypatch.delenv(k, raising=False)
    sys.path.insert(0, ROOT)
    import bench
    capfd.readouterr()
    bench.main(['--gpus', '2', '--steps', '1', '--warmup', '1', '--batch', '2', '--frames', '24', '--no-cpu-baseline', '--no-side-metrics', '--strong-total', '6'],
               solver_factory=_EmuSolver)
    out = capfd.readouterr().out
    lines = [ln for ln in out.splitlines() if ln.strip().startswith('{')]
    assert len(lines) == 1, out
    o = json.loads(lines[0])
    _check_line(o, 2)
    st = o['strong_scaling']
    assert st['scaling'] == 'strong' and st['total_sequences'] == 6 and st['sequences_rank0'] == 3 and st['n_gpus'] == 2
    assert abs(st['value'] * st['seconds'] - 6.0) < 1e-9


def test_bench_refuses_a_world_that_is_not_gpus(monkeypatch):
    """--gpus 4 under a launcher that started 2 ranks is an error, not a silent 2-rank (or 1-rank) run"""
    import pytest
    sys.path.insert(0, ROOT)
    import bench
    monkeypatch.setenv('WORLD_SIZE', '2'); monkeypatch.setenv('RANK', '0')
    with pytest.raises(SystemExit):
        bench.main(['--gpus', '4', '--steps', '1', '--batch', '2', '--frames', '24'], solver_factory=_EmuSolver)


def test_bench_as_eight_ranks_from_a_plain_shell(capfd, monkeypatch):
    """The driver's largest configuration without hardware: `bench.main(['--gpus', '8', ...])` starts EIGHT gloo ranks itself.  One JSON line,
    n_gpus 8, the weak-scaling aggregate covers 8 x batch sequences, and the strong-scaling leg's LPT shards cover its fixed total exactly once (16 sequences: two
    per rank)."""
    sys.path.insert(0, os.path.join(HERE, 'host_emu'))
    import emu
    emu.build()
    for k in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT', 'LOCAL_WORLD_SIZE'):
        monkeypatch.delenv(k, raising=False)
    sys.path.insert(0, ROOT)
    import bench
    capfd.readouterr()
    bench.main(['--gpus', '8', '--steps', '1', '--warmup', '0', '--batch', '1', '--frames', '24', '--no-cpu-baseline', '--no-side-metrics', '--strong-total', '16'],
               solver_factory=_EmuSolver)
    out = capfd.readouterr().out
    lines = [ln for ln in out.splitlines() if ln.strip().startswith('{')]
    assert len(lines) == 1, out
    o = json.loads(lines[0])
    assert o['n_gpus'] == 8 and o['scaling'] == 'weak' and o['roofline']['peak'] == 8000.0 * 8
    assert abs(o['value'] * o['ms_per_step'] * 1e-3 - 8.0) < 1e-6          # batch 1, one step, per rank: 8 sequences in the timed region
    st = o['strong_scaling']
    assert st['total_sequences'] == 16 and st['sequences_rank0'] == 2 and st['n_gpus'] == 8
    assert abs(st['value'] * st['seconds'] - 16.0) < 1e-9
    assert o['strong_scaling_value'] == st['value']                       # (top-level copy for parsers that keep first-level numbers)


def test_rank_local_workloads_are_disjoint_and_shards_cover_the_total_once():
    """What makes the 8-rank numbers meaningful: rank r's weak-scaling seeds are r*K*128 .. (no sequence is solved twice), and `sharding.lpt_assign` of the 4 000
    strong-scaling sequences over 8 ranks is a partition (every index exactly once, loads within one sequence of each other for equal lengths)."""
    import chd_amd  # noqa: F401
    from chd_amd.sharding import lpt_assign
    K, B, W = 20, 128, 8
    seeds = [set(range(r * K * B, (r + 1) * K * B)) for r in range(W)]
    assert sum(len(s) for s in seeds) == len(set().union(*seeds)) == W * K * B
    shards = lpt_assign([90] * 4000, W)
    flat = sorted(i for sh in shards for i in sh)
    assert flat == list(range(4000)) and max(len(sh) for sh in shards) - min(len(sh) for sh in shards) <= 1
    mixed = lpt_assign([60, 120] * 999 + [600, 90], W)
    assert sorted(i for sh in mixed for i in sh) == list(range(2000))

"""Iteration statistics of the solver on the fixture's sequences, on the CPU: runs the host emulation of the kernel source (tests/host_emu, in lockstep with the
oracle) on the cases of tests/golden/make_bench_parity_golden.py and on any physics input directories given, and compares stage statuses / iteration counts with the
committed fixture.  What an algorithm change does to the workload is known minutes later, without the GPU box.

    python tests/tools/emu_sweep.py [small|hard|all] [--dirs DIR:FRAMES ...] [--workers 8] [--cap 7000]
"""
import argparse
import multiprocessing as mp
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
TESTS = os.path.dirname(HERE)
ROOT = os.path.dirname(TESTS)
for p in (ROOT, TESTS, os.path.join(TESTS, 'golden'), os.path.join(TESTS, 'host_emu')):
    sys.path.insert(0, p)


def work(job):
    import chd_amd  # noqa: F401
    import emu
    import make_bench_parity_golden as G
    from chd_amd.phys_capi import default_config
    case, caps = job
    t0 = time.time()
    if case[0] == 'dir':
        from chd_amd.io_formats import read_inputs
        seq = read_inputs(case[1], case[2]); key = os.path.basename(case[1].rstrip('/'))
        if key.startswith('phys_optim_in'):
            key = os.path.basename(os.path.dirname(case[1].rstrip('/')))
    else:
        seq = G.make_case(*case[1:]); key = G.case_key(*case[1:])
    e = emu.EmuProblem(seq, default_config(max_iter=caps))
    e.solve(0, 4)
    st, _ = e.results()
    n_st = 5
    if int(st[4][0]) != 0:
        if e.rebuild_fallback():
            e.solve(5, 5); st, _ = e.results()
        n_st = 6
    return key, [(int(st[s][0]), int(st[s][1])) for s in range(n_st)], [float(st[s][4]) for s in range(n_st)], [int(st[s][6]) for s in range(n_st)], time.time() - t0


if __name__ == '__main__':
    import make_bench_parity_golden as G
    ap = argparse.ArgumentParser()
    ap.add_argument('which', nargs='?', default='small')
    ap.add_argument('--dirs', nargs='*', default=[])
    ap.add_argument('--workers', type=int, default=8)
    ap.add_argument('--cap', type=int, default=0)
    a = ap.parse_args()
    import emu
    emu.build()
    caps = [a.cap] * 6 if a.cap > 0 else G.CAPS
    sets = {'small': G.FLAT[:16] + G.TILTED[:8] + G.HARD[:12] + G.PIPE, 'hard': G.HARD + G.PIPE, 'all': G.FLAT + G.TILTED + G.HARD + G.PIPE, 'none': []}
    cases = [('seed',) + c for c in sets[a.which]] + [('dir', d.split(':')[0], int(d.split(':')[1])) for d in a.dirs]
    g = np.load(os.path.join(TESTS, 'golden', 'bench_parity_golden.npz'))
    tot_new = tot_old = fail_new = fail_old = 0
    its = []; nf_tot = 0
    with mp.get_context('spawn').Pool(a.workers) as pool:
        for key, st, obj, nf, dt in pool.imap_unordered(work, [(c, caps) for c in cases]):
            it_new = sum(s[1] for s in st)
            line = '%-18s %5.0fs  %4d %s' % (key, dt, it_new, st)
            its.append(it_new); nf_tot += sum(nf)
            if key + '_iters' in g.files:
                io = [int(v) for v in g[key + '_iters']]; so = [int(v) for v in g[key + '_status']]
                tot_new += it_new; tot_old += sum(io)
                fail_new += sum(1 for s in st if s[0] != 0); fail_old += sum(1 for s in so if s != 0)
                line += ' | fixture %4d %s' % (sum(io), list(zip(so, io)))
            print(line, flush=True)
    its = np.array(its)
    print('TOTAL iterations %d (fixture %d), failed stages %d (fixture %d); all cases: mean %.1f p50 %d p90 %d max %d, factorisations / iteration %.2f'
          % (tot_new, tot_old, fail_new, fail_old, its.mean(), np.percentile(its, 50), np.percentile(its, 90), its.max(), nf_tot / max(1, its.sum())))

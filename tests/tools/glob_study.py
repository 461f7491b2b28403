"""Globalisation / damping studies on the CPU (round 6): runs the host emulation of the kernel source with the study switches of chd_kernels.hpp set through the
environment (CHD_GLOB_FILTER, CHD_FILTER_MAXBT, CHD_L1_FLOOR, ...) on
  * the lockstep fixture's 200 sequences (+ the four pipeline clips it holds),
  * W: 256 fresh walks (seeds 3000..3255) -- the fixture's 36 hard seeds are the PREVIOUS rules' stragglers and flatter any change,
  * the bench workload's known stragglers (seeds 1688, 88: 199 / 231 iterations in the duration stage),
  * physics input directories the repo's own upstream stages produced (tools/gpu_r06_start.sh: 192 clips x 100 frames = sets A, B, C of 64),
and prints ONE summary row per set: what VERDICT r05 next-1 asks a rule to be judged by (total, max, stage-3 p99, clips above 800 / at the cap, failed stages).

    python tests/tools/glob_study.py --name filter CHD_GLOB_FILTER=1 [--sets fixture stragglers A B C] [--clips /tmp/pipe192] [--workers 8]
"""
import argparse
import json
import multiprocessing as mp
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
TESTS = os.path.dirname(HERE)
ROOT = os.path.dirname(TESTS)
for p in (ROOT, TESTS, os.path.join(TESTS, 'golden'), os.path.join(TESTS, 'host_emu'), HERE):
    sys.path.insert(0, p)


def summarise(rows, cap_dur=2000):
    tot = np.array([sum(s[1] for s in st) for _, st in rows])
    dur = np.array([st[4][1] for _, st in rows])                  # the duration stage (phys_optim.cpp:664-712)
    failed = sum(1 for _, st in rows for s in st if s[0] != 0)
    fallbacks = sum(1 for _, st in rows if len(st) > 5)
    at_cap = sum(1 for _, st in rows for s in st if s[0] == -1)
    return {'n': len(rows), 'iterations': int(tot.sum()), 'mean': float(tot.mean()), 'p50': int(np.percentile(tot, 50)), 'p90': int(np.percentile(tot, 90)), 'p99': int(np.percentile(tot, 99)),
            'max': int(tot.max()), 'dur_p50': int(np.percentile(dur, 50)), 'dur_p99': int(np.percentile(dur, 99)), 'dur_max': int(dur.max()), 'above_800': int((tot > 800).sum()),
            'stages_at_cap': at_cap, 'failed_stages': failed, 'fallbacks': fallbacks, 'worst': [k for _, k in sorted(((int(t), k) for t, (k, _) in zip(tot, rows)), reverse=True)[:4]]}


if __name__ == '__main__':
    import make_bench_parity_golden as G
    ap = argparse.ArgumentParser()
    ap.add_argument('env', nargs='*', help='KEY=VALUE study switches')
    ap.add_argument('--name', default='study')
    ap.add_argument('--sets', nargs='*', default=['fixture', 'W', 'stragglers', 'A', 'B', 'C'])
    ap.add_argument('--clips', default='/tmp/pipe192')
    ap.add_argument('--workers', type=int, default=8)
    ap.add_argument('--out', default='/tmp/glob_study')
    a = ap.parse_args()
    for kv in a.env:
        k, v = kv.split('=', 1); os.environ[k] = v
    import emu
    emu.build()
    from emu_sweep import work
    sets = {}
    if 'fixture' in a.sets:
        sets['fixture'] = [('seed',) + c for c in G.FLAT + G.TILTED + G.HARD + G.PIPE]
    if 'small' in a.sets:
        sets['small'] = [('seed',) + c for c in G.FLAT[:16] + G.TILTED[:8] + G.HARD[:12] + G.PIPE]
    if 'W' in a.sets:          # 256 FRESH walks (seeds outside every fixture): what the fixture's hand-picked hard seeds cannot tell (profiles/r06_globalisation_study.md, ratio rule)
        sets['W'] = [('seed', 3000 + i, 90, 0.0) for i in range(256)]
    if 'stragglers' in a.sets:
        sets['stragglers'] = [('seed', 1688, 90, 0.0), ('seed', 88, 90, 0.0)]
    for name, lo in (('A', 0), ('B', 64), ('C', 128)):
        if name in a.sets:
            sets[name] = [('dir', os.path.join(a.clips, 'video_%03d' % i, 'phys_optim_in_combined'), 100) for i in range(lo, lo + 64)]
    os.makedirs(a.out, exist_ok=True)
    result = {'name': a.name, 'env': a.env}
    with mp.get_context('spawn').Pool(a.workers) as pool:
        for sname, cases in sets.items():
            t0 = time.time()
            rows = []
            jobs = [(c, G.CAPS) for c in cases]
            for k, (key, st, obj, nf, dt) in enumerate(pool.imap_unordered(work, jobs)):
                rows.append((key, st))
            result[sname] = summarise(rows); result[sname]['seconds'] = round(time.time() - t0, 1)
            result[sname + '_rows'] = {k: st for k, st in rows}
            r = result[sname]
            print('%-14s %-10s n %3d  iters %6d  p50 %4d p90 %4d p99 %4d max %5d | dur stage p50 %3d p99 %4d max %4d | >800: %d  at cap: %d  failed stages %d  fallbacks %d  (%.0f s)'
                  % (a.name, sname, r['n'], r['iterations'], r['p50'], r['p90'], r['p99'], r['max'], r['dur_p50'], r['dur_p99'], r['dur_max'], r['above_800'], r['stages_at_cap'], r['failed_stages'], r['fallbacks'], r['seconds']), flush=True)
    json.dump(result, open(os.path.join(a.out, a.name + '.json'), 'w'), indent=1)

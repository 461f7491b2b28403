"""Generates tests/golden/quality_golden.npz: the ACCEPTANCE side of the physics parity fixtures, independent of the solver's rules.

For the 32 bench seeds of the quality study (profiles/r03_solution_quality.md, tests/tools/solution_quality.py) it stores, per output snapshot (stages 1.2, 2.2, 3),
the objective of the CONVERGED staged solve (tol 1e-6, reference iteration caps) -- a property of the NLP and its staging, not of how fast a solver gets there -- next
to the shipped solve's objective and largest constraint violation at the reference's tol 1e-3 when the file was made.

tests/test_quality_gate.py holds every later build to it: largest constraint violation <= 1e-4 (IPOPT's constr_viol_tol, the solver's own acceptance test) and
objective <= GATE x converged objective, so that a rule which makes the solver faster by stopping WORSE fails a test -- the lockstep fixtures
(bench_parity_golden.npz) follow the solver and cannot object.  This file is made ONCE; regenerate it only when the NLP itself changes.

    python tests/golden/make_quality_golden.py [--workers 8] [--seeds 32]          (about 8 CPU-minutes per seed at 90 frames)
"""
import argparse
import multiprocessing as mp
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests')); sys.path.insert(0, os.path.join(ROOT, 'tests', 'tools'))
FRAMES = 90


def work(seed):
    import chd_amd  # noqa: F401
    from chd_amd.synth import make_walk
    from solution_quality import staged
    seq = make_walk(seed=seed, F=FRAMES, randomize=True)
    t0 = time.time()
    sa, a, _ = staged(seq, 1e-3)
    sc, c, _ = staged(seq, 1e-6)
    return seed, sa, [r['obj'] for r in a], [r['viol'] for r in a], sc, [r['obj'] for r in c], [r['viol'] for r in c], time.time() - t0


if __name__ == '__main__':
    ap = argparse.ArgumentParser()
    ap.add_argument('--workers', type=int, default=8)
    ap.add_argument('--seeds', type=int, default=32)
    ap.add_argument('--out', default=os.path.join(HERE, 'quality_golden.npz'))
    a = ap.parse_args()
    from oracle import oracle
    oracle.build()
    n = a.seeds
    obj_c = np.zeros((n, 3)); obj_a = np.zeros((n, 3)); viol_a = np.zeros((n, 3)); viol_c = np.zeros((n, 3)); it_a = np.zeros(n, dtype=np.int32); it_c = np.zeros(n, dtype=np.int32)
    with mp.get_context('spawn').Pool(a.workers) as pool:
        for seed, sa, oa, va, sc, oc, vc, dt in pool.imap_unordered(work, range(n)):
            obj_a[seed] = oa; viol_a[seed] = va; obj_c[seed] = oc; viol_c[seed] = vc
            it_a[seed] = sum(s[1] for s in sa); it_c[seed] = sum(s[1] for s in sc)
            print('seed %2d  %5.0f s  objective ratio %s  violation %s  iterations %d / %d' % (seed, dt, np.round(np.array(oa) / np.array(oc), 4), ['%.1e' % v for v in va], it_a[seed], it_c[seed]), flush=True)
    np.savez_compressed(a.out, seeds=np.arange(n), frames=FRAMES, objective_converged=obj_c, violation_converged=viol_c,
                        objective_at_tol_1e3_when_made=obj_a, violation_at_tol_1e3_when_made=viol_a, iterations_at_tol_1e3_when_made=it_a, iterations_converged=it_c)
    r = obj_a / obj_c
    print('wrote %s: objective ratio median %s max %s; largest violation %.1e' % (a.out, np.round(np.median(r, axis=0), 4), np.round(r.max(axis=0), 4), viol_a.max()))

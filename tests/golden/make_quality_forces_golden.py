"""Generates tests/golden/quality_forces_golden.npz: the FORCE side of the acceptance gate.

The objective of the NLP has no force term and 30 fps position data pin the centre-of-mass acceleration weakly, so at the reference's tol 1e-3 the ground
reaction forces are the least determined output: the round-3 / round-4 studies (profiles/r04_solution_quality.md) found them ~73 % away from the solution of the
same algorithm at tol 1e-6 while objective, COM and feet agree to 1e-2 .. 1e-3.  Until round 5 no test saw that number.  This file stores, for the 32 seeds of
tests/golden/quality_golden.npz, the trajectories of the staged solve at tol 1e-6 (the kernel SOURCE through tests/host_emu: the same algorithm as the oracle
at ~40 x its speed; its objective is stored next to quality_golden.npz's, which came from the oracle -- they agree where every stage converges and differ by a
few per cent where the duration stage fails at 1e-6 and the stage-4 fallback takes over) and the distance of the tol-1e-3 solve to them when the file was made.
tests/test_quality_gate.py holds every later build to: dynamics-row residual <= 1e-4 (recomputed by the oracle's model at the returned point) and a force
distance no worse than when this file was made.

    python tests/golden/make_quality_forces_golden.py [--workers 8] [--seeds 32]          (about 1 CPU-minute per seed)
"""
import argparse
import multiprocessing as mp
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests')); sys.path.insert(0, os.path.join(ROOT, 'tests', 'host_emu'))
FRAMES = 90
CAPS = [7000, 7000, 7000, 2500, 2000, 7000]
KEYS = ('base_lin', 'base_ang_deg', 'ee_pos', 'ee_force')


def staged(seq, tol):
    import emu
    from chd_amd.phys_capi import default_config
    e = emu.EmuProblem(seq, default_config(max_iter=CAPS, tol=tol))
    e.solve(0, 4)
    st, sn = e.results()
    if int(st[4][0]) != 0 and e.rebuild_fallback():
        e.solve(5, 5); st, sn = e.results()
    last = 5 if int(st[4][0]) != 0 else 4
    return [int(s[0]) for s in st], int(sum(s[1] for s in st)), [st[1][4], st[3][4], st[last][4]], [st[1][3], st[3][3], st[last][3]], sn


def work(seed):
    import chd_amd  # noqa: F401
    from chd_amd.synth import make_walk
    from common import rel_l2
    seq = make_walk(seed=seed, F=FRAMES, randomize=True)
    t0 = time.time()
    sa, ia, oa, va, na = staged(seq, 1e-3)
    sc, ic, oc, vc, nc = staged(seq, 1e-6)
    dist = np.zeros((3, len(KEYS)))
    for k in range(3):
        for q, name in enumerate(KEYS):
            a, c = np.asarray(na[k][name]), np.asarray(nc[k][name])
            dist[k, q] = rel_l2(a, c) if a.shape == c.shape and np.linalg.norm(c) > 0 else 0.0
    return seed, sc, ic, oc, vc, nc, ia, oa, dist, time.time() - t0


if __name__ == '__main__':
    ap = argparse.ArgumentParser()
    ap.add_argument('--workers', type=int, default=8)
    ap.add_argument('--seeds', type=int, default=32)
    ap.add_argument('--out', default=os.path.join(HERE, 'quality_forces_golden.npz'))
    a = ap.parse_args()
    import emu
    emu.build()
    n = a.seeds
    out = dict(seeds=np.arange(n), frames=FRAMES, keys=np.array(KEYS), objective_converged=np.zeros((n, 3)), violation_converged=np.zeros((n, 3)),
               status_converged=np.zeros((n, 6), dtype=np.int32), iterations_converged=np.zeros(n, dtype=np.int32), iterations_at_tol_1e3_when_made=np.zeros(n, dtype=np.int32),
               objective_at_tol_1e3_when_made=np.zeros((n, 3)), distance_at_tol_1e3_when_made=np.zeros((n, 3, len(KEYS))))
    with mp.get_context('spawn').Pool(a.workers) as pool:
        for seed, sc, ic, oc, vc, nc, ia, oa, dist, dt in pool.imap_unordered(work, range(n)):
            out['objective_converged'][seed] = oc; out['violation_converged'][seed] = vc; out['status_converged'][seed] = sc; out['iterations_converged'][seed] = ic
            out['iterations_at_tol_1e3_when_made'][seed] = ia; out['objective_at_tol_1e3_when_made'][seed] = oa; out['distance_at_tol_1e3_when_made'][seed] = dist
            for k in range(3):
                for name in KEYS:
                    out['s%d_snap%d_%s' % (seed, k, name)] = np.asarray(nc[k][name])
            print('seed %2d  %4.0f s  iterations %d / %d  force distance %s  COM %s' % (seed, dt, ia, ic, np.round(dist[:, 3], 3), np.round(dist[:, 0], 4)), flush=True)
    np.savez_compressed(a.out, **out)
    d = out['distance_at_tol_1e3_when_made']
    print('wrote %s: forces vs converged median %s max %s; COM median %s' % (a.out, np.round(np.median(d[:, :, 3], axis=0), 3), np.round(d[:, :, 3].max(axis=0), 3), np.round(np.median(d[:, :, 0], axis=0), 4)))

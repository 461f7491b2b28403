"""Timing of the two factorisations and the substitution on single KKT matrices of the bench workload (analysis tool):
    python tests/tools/gpu_factor_bench.py [--seeds 0 1 2 3] [--reps 20]
One workgroup on an otherwise idle GPU -- per-call latency, not throughput under load."""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import chd_amd  # noqa: E402,F401
from chd_amd.phys_optim import PhysOptim, default_config  # noqa: E402
from chd_amd.synth import make_walk  # noqa: E402

if __name__ == '__main__':
    ap = argparse.ArgumentParser()
    ap.add_argument('--seeds', type=int, nargs='+', default=[0, 1, 2, 3])
    ap.add_argument('--reps', type=int, default=20)
    ap.add_argument('--frames', type=int, default=90)
    a = ap.parse_args()
    s = PhysOptim(device=0, config=default_config())
    seqs = [make_walk(seed=sd, F=a.frames, randomize=True) for sd in a.seeds]
    b = s.upload(seqs)
    rng = np.random.default_rng(0)
    print('| seed | stage | N | w | border | register front us | left-looking us | right-looking us | solve us | rel. difference front / left-looking vs right-looking |')
    print('|---|---|---|---|---|---|---|---|---|---|')
    for q, sd in enumerate(a.seeds):
        for stage in range(5):
            sz = b.sizes(q, stage)
            rhs = rng.normal(size=sz['kkt_dim'])
            x0, i0 = b.debug_linsolve(q, stage, rhs, dw=1e-2, dval=1e-3, which=0, reps=a.reps)
            x1, i1 = b.debug_linsolve(q, stage, rhs, dw=1e-2, dval=1e-3, which=1, reps=a.reps)
            x2, i2 = b.debug_linsolve(q, stage, rhs, dw=1e-2, dval=1e-3, which=2, reps=a.reps)
            print('| %d | %d | %d | %d | %d | %.0f%s | %.0f%s | %.0f | %.0f | %.1e / %.1e |' % (sd, stage, sz['kkt_dim'], sz['halfband'], sz['border'], i2['factor_us'], '' if i2['ran'] == 2 else ' (fell back)',
                                                                                      i0['factor_us'], '' if i0['ran'] == 0 else ' (fell back)', i1['factor_us'], i0['solve_us'],
                                                                                      np.linalg.norm(x2 - x1) / np.linalg.norm(x1), np.linalg.norm(x0 - x1) / np.linalg.norm(x1)), flush=True)
            if stage in (2, 4):
                print('|   | phases (us) | front: extract %.0f (tile wavefronts %.0f), diagonal+slots %.0f (slots alone %.0f), rows %.0f, update %.0f (tile wavefronts %.0f), border %.0f | right-looking: copy %.0f, load %.0f, rows %.0f, look-ahead wavefront %.0f, store + wait %.0f, border %.0f; look-ahead wavefront (CHD_DIAG_TIMING builds): loads + tiles %.0f, hand-over (waits for the block) %.0f, stores %.0f, chain %.0f | | | | |'
                      % tuple([i2['phase_us'][k] for k in (8, 15, 9, 13, 10, 11, 14, 12)] + [i1['phase_us'][k] for k in (6, 8, 9, 11, 10, 12, 7, 13, 14, 15)]), flush=True)
    b.free(); s.close()

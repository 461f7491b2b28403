"""Timing of the factorisation and the substitution on single KKT matrices of the bench workload (analysis tool):
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
    print('| seed | stage | N | w | border | factorisation us | solve us | phases (us): copy, panel load, row solve, look-ahead wavefront, store + wait, border |')
    print('|---|---|---|---|---|---|---|---|')
    for q, sd in enumerate(a.seeds):
        for stage in range(5):
            sz = b.sizes(q, stage)
            rhs = rng.normal(size=sz['kkt_dim'])
            x1, i1 = b.debug_linsolve(q, stage, rhs, dw=1e-2, dval=1e-3, reps=a.reps)
            print('| %d | %d | %d | %d | %d | %.0f | %.0f | %s |' % (sd, stage, sz['kkt_dim'], sz['halfband'], sz['border'], i1['factor_us'], i1['solve_us'],
                                                                   ', '.join('%.0f' % i1['phase_us'][k] for k in (6, 8, 9, 11, 10, 12))), flush=True)
    b.free(); s.close()

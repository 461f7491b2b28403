"""Generates tests/golden/phys_golden.npz: the CPU oracle's three output snapshots and stage statistics for
seeded synthetic sequences.  The reference binary cannot be built, so these vectors pin
HIP-vs-oracle parity and oracle regressions, NOT oracle-vs-IPOPT parity ("parity unpinned")."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import chd_amd  # noqa: E402,F401
from chd_amd.synth import make_walk  # noqa: E402
from common import oracle_run  # noqa: E402

CASES = [(0, 60), (5, 90), (2, 40)]      # (seed, frames)
CAPS = [300] * 6

if __name__ == '__main__':
    out = {}
    for seed, F in CASES:
        seq = make_walk(seed=seed, F=F, randomize=True)
        stats, snaps = oracle_run(seq, CAPS)
        key = 's%d_F%d' % (seed, F)
        out[key + '_status'] = np.array([s[0] for s in stats]); out[key + '_iters'] = np.array([s[1] for s in stats])
        out[key + '_obj'] = np.array([s[2] for s in stats])
        for k, sn in enumerate(snaps):
            for name in ('base_lin', 'base_ang_deg', 'ee_pos', 'ee_force', 'contact'):
                out['%s_snap%d_%s' % (key, k, name)] = np.asarray(sn[name])
        print(key, stats)
    np.savez_compressed(os.path.join(HERE, 'phys_golden.npz'), **out)

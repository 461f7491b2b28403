"""Generates tests/golden/bench_parity_golden.npz: the CPU oracle's results on the BASELINE workload itself, so that the
GPU parity test (tests/test_gpu_parity.py::test_bench_workload_matches_oracle) and bench.py's `parity` block can hold the
HIP path to the oracle on EVERY bench sequence without spending GPU-box minutes on CPU solves.

  flat    seeds 0..127, F = 90, flat floor        (BASELINE.json configs[1]: exactly bench.py's batch of rank 0)
  tilted  seeds 200..231, F = 90, floor tilted by 2..9.75 degrees about x (the sequences on which round 1's
          parity holes showed up were mostly tilted ones)
  hard    36 seeds of bench.py's --steps 20 workload that the round-2 solver struggled with (stage-4 fallbacks, most iterations)
  pipe    four 60-frame sequences of the kind the kinematic optimisation produces (tests/golden/pipe_inputs/)
  long    one 600-frame sequence, 10 degree tilt  (configs[4]) -- only with --long (takes tens of minutes on one core)

Reference iteration caps 7000/7000/7000/2500/2000/7000, tol 1e-3 (phys_optim.cpp:571-743) and the solver's default
options -- the same configuration bench.py measures.

The reference binary cannot be built here, so these vectors pin HIP-vs-oracle parity, NOT oracle-vs-IPOPT
parity.  Snapshots are stored as float64; ~8 cores x 5 minutes.

    python tests/golden/make_bench_parity_golden.py [--workers 8] [--long]
"""
import argparse
import multiprocessing as mp
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))

CAPS = [7000, 7000, 7000, 2500, 2000, 7000]
FLAT = [(s, 90, 0.0) for s in range(128)]
TILTED = [(200 + i, 90, 2.0 + 0.25 * i) for i in range(32)]
LONG = [(0, 600, 10.0)]
# the sequences of bench.py's 2 560 (--steps 20) that were hard for the round-2 solver: its 20 stage-4 fallbacks (six of them ended by
# the stall guard) and the 24 with the most iterations (gpurun_out/r03a/seq_stats.npz, tests/tools/gpu_seq_stats.py)
HARD = [(s, 90, 0.0) for s in (139, 162, 197, 284, 302, 436, 659, 731, 827, 842, 983, 1051, 1231, 1288, 1384, 1416, 1449, 1554, 1682, 1688, 1779, 1823,
                               1887, 1907, 1945, 2010, 2105, 2133, 2194, 2242, 2256, 2281, 2395, 2459, 2510, 2556)]


# sequences of another family: what the kinematic optimisation hands to the physics stage for synthetic standing-up clips (the pipeline
# benchmark's videos 16, 8, 17, 4: tests/golden/pipe_inputs/video_0XX/ = their phys_optim_in_combined/ directories, 60 frames) -- seeds >= 100000
PIPE = [(100000 + v, 60, 0.0) for v in (16, 8, 17, 4)]


def case_key(seed, F, tilt):
    return 's%d_F%d_t%03d' % (seed, F, int(round(tilt * 100)))


def make_case(seed, F, tilt):
    import chd_amd  # noqa: F401
    if seed >= 100000:
        from chd_amd import io_formats as iof
        return iof.read_inputs(os.path.join(HERE, 'pipe_inputs', 'video_%03d' % (seed - 100000)), F)
    from chd_amd.synth import make_walk
    return make_walk(seed=seed, F=F, randomize=True, tilt_deg=tilt)


def _work(case):
    from common import oracle_run
    from oracle.oracle import lib
    lib().orc_set_stall_window(int(os.environ.get('CHD_FIXTURE_STALL_WINDOW', '0')))      # (study runs only: the committed fixture is made with the guard off)
    seed, F, tilt = case
    t0 = time.time()
    stats, snaps = oracle_run(make_case(seed, F, tilt), CAPS)
    return case, stats, snaps, time.time() - t0


if __name__ == '__main__':
    ap = argparse.ArgumentParser()
    ap.add_argument('--workers', type=int, default=8)
    ap.add_argument('--long', action='store_true')
    ap.add_argument('--missing', action='store_true', help='solve only the cases the existing fixture lacks')
    ap.add_argument('--out', default=os.path.join(HERE, 'bench_parity_golden.npz'))
    args = ap.parse_args()
    from oracle import oracle
    oracle.build()
    cases = FLAT + TILTED + HARD + PIPE + (LONG if args.long else [])
    out = {}
    if (args.long or args.missing) and os.path.exists(args.out):          # keep what a previous run produced: only missing cases are solved
        old = np.load(args.out)
        out = {k: old[k] for k in old.files}
        cases = [c for c in cases if case_key(*c) + '_status' not in out]
    t0 = time.time()
    with mp.get_context('spawn').Pool(args.workers) as pool:
        for case, stats, snaps, dt in pool.imap_unordered(_work, cases):
            key = case_key(*case)
            out[key + '_status'] = np.array([s[0] for s in stats], dtype=np.int32)
            out[key + '_iters'] = np.array([s[1] for s in stats], dtype=np.int32)
            out[key + '_obj'] = np.array([s[2] for s in stats])
            for k, sn in enumerate(snaps):
                for name in ('base_lin', 'base_ang_deg', 'ee_pos', 'ee_force'):
                    out['%s_snap%d_%s' % (key, k, name)] = np.asarray(sn[name], dtype=np.float64)
                out['%s_snap%d_contact' % (key, k)] = np.asarray(sn['contact'], dtype=np.uint8)
            print('%s  %5.1f s  %s' % (key, dt, [(s[0], s[1]) for s in stats]), flush=True)
    np.savez_compressed(args.out, **out)
    print('wrote %s: %d cases, %.0f s wall, %.1f MB' % (args.out, len(FLAT + TILTED + HARD + PIPE) + (1 if args.long else 0), time.time() - t0,
                                                      os.path.getsize(args.out) / 1e6))

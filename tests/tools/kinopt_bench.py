"""Kinematic optimisation on one MI355X: a batch of synthetic clips on the combined 28-joint skeleton through the whole
`KinematicOptimizer.optimize` (IK initialisation on libchd_ik.so, two least-squares solves on libchd_kinopt.so, floor fit on the
host), with the device time of the two solves and -- on clip 0, bounded to a few frames' worth of time -- the oracle's CPU time.

    python tests/tools/kinopt_bench.py [clips=256] [frames=100] [oracle_frames=12]

The skeleton template comes from the committed fixture (tests/golden/kinopt_golden.npz); nothing reads /root/reference."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import chd_amd  # noqa: E402,F401
from chd_amd import kinematic_optimizer as kopt  # noqa: E402
from chd_amd import skeleton_io as sio  # noqa: E402

G = np.load(os.path.join(ROOT, 'tests', 'golden', 'kinopt_golden.npz'))
OFFSETS, PARENTS = G['c0_skel_offsets'], G['c0_skel_parents']


def make_clip(seed, F):
    from chd_amd.synth import make_kin_clip
    return make_kin_clip(seed, F, OFFSETS, PARENTS)


if __name__ == '__main__':
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    F = int(sys.argv[2]) if len(sys.argv) > 2 else 100
    FO = int(sys.argv[3]) if len(sys.argv) > 3 else 12
    clips = [make_clip(s, F) for s in range(B)]
    opt = kopt.KinematicOptimizer(device=0)
    if os.environ.get('KIN_LDS_DOUBLES'):
        opt.kin.cfg.reserved[1] = int(os.environ['KIN_LDS_DOUBLES'])
    if os.environ.get('KIN_FRAMES_PER_WORKGROUP'):
        opt.kin.cfg.reserved[2] = int(os.environ['KIN_FRAMES_PER_WORKGROUP'])
    opt.optimize([make_clip(10_000, 8)])                      # warm-up (module load)
    kin_ms = []
    real_solve = opt.kin.solve

    def timed(problems):
        r = real_solve(problems)
        kin_ms.append(opt.kin.last_kernel_ms())
        return r

    opt.kin.solve = timed
    t0 = time.perf_counter(); res = opt.optimize(clips, chunk=int(os.environ.get('KIN_CHUNK', '256')), workers=int(os.environ.get('KIN_WORKERS', '2'))); t1 = time.perf_counter()
    chunk_marks = opt.timings.get('chunks', [])[-max(1, (B + 255) // 256 * 2):] if B > 128 else opt.timings.get('chunks', [])[-1:]
    ik_ms, ik_frames = opt.ik.last_kernel_ms()
    its = np.array([[s['lsmr_iterations'] for s in r['stages']] for r in res])
    nfev = np.array([[s['nfev'] for s in r['stages']] for r in res])
    n, m = 87 * F, 507 * F - 423
    # algorithmic bytes per LSMR iteration ("kinematic optimisation"): u (m) read + written, v / h / hbar / x (n each) read + written, the
    # linearisation (420 F doubles) read twice (J v, J^T u); per clip.  Since round 5 they move through LDS, not HBM.
    alg_bytes = 8.0 * (2 * m + 8 * n + 2 * 420 * F) * its.sum()
    out = dict(clips=B, frames=F, unknowns=n, rows=m, wall_s=t1 - t0, clips_per_s=B / (t1 - t0), ik_kernel_ms=ik_ms, lsq_kernel_ms=kin_ms,
               lsmr_iterations_per_clip=float(its.sum(axis=1).mean()), nfev_per_stage=nfev.mean(axis=0).tolist(),
               status_counts={str(k): int(v) for k, v in zip(*np.unique([s['status'] for r in res for s in r['stages']], return_counts=True))},
               algorithmic_GBps=alg_bytes / (sum(kin_ms) * 1e-3) / 1e9,
               lsq_time_share={'jv': float(np.mean([s['jv_fraction'] for r in res for s in r['stages']])), 'jtu': float(np.mean([s['jtu_fraction'] for r in res for s in r['stages']]))},
               chunk_timelines=[[(n, round(t - t0, 3)) for n, t in m] for m in chunk_marks],
               relabelled_contacts_per_clip=float(np.mean([np.abs(r['velConstraints'] - c['velConstraints']).sum() for r, c in zip(res, clips)])))
    # CPU: the oracle (dense restatement of the reference, with SciPy's sparse products) on a short clip, scaled per frame
    if FO > 0:
        from oracle import kinopt_oracle as ko      # checker / CPU baseline only
        c = make_clip(0, FO)
        t2 = time.perf_counter()
        ko.optimize_trajectory(c['poses2D'], c['joint_conf_2d'], c['poses3D'], c['root_pos'], c['joint_angles'], c['offsets'], c['parents'], (c['ppx'], c['ppy']), c['camFocal'], c['velConstraints'])
        out['oracle_s_for_%d_frames_one_core' % FO] = time.perf_counter() - t2
    print(json.dumps(out))

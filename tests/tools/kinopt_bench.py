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
    """Same recipe as tests/golden/make_kinopt_golden.py: smooth random joint angles with quiet legs, a swing-knee bend, a root
    drifting in front of the camera, noisy 3D joints / 2D projections / confidences, alternating contacts with one spurious label."""
    rng = np.random.default_rng(seed)
    nj = 28
    t = np.arange(F)[:, None, None] / 30.0
    amp = rng.uniform(0.05, 0.35, size=(1, nj, 3)); ph = rng.uniform(0, 2 * np.pi, size=(1, nj, 3)); fr = rng.uniform(0.5, 2.0, size=(1, nj, 3))
    amp[:, 1:13] = rng.uniform(0.01, 0.04, size=(1, 12, 3))
    eul = amp * np.sin(2 * np.pi * fr * t + ph)
    eul[:, 0] += np.array([0.1, 0.4, 0.05])
    half = F // 2
    swing = np.sin(np.pi * np.clip((np.arange(F) - half) / max(F - half - 1, 1), 0, 1)) ** 2
    eul[:, 2, 0] += 0.9 * swing; eul[:, 8, 0] += 0.9 * swing[::-1]
    rot = sio.quat_from_euler(eul, order='xyz', world=True)
    offsets = OFFSETS * rng.uniform(0.9, 1.15)
    root = np.array([20.0, 40.0, 320.0]) + np.arange(F)[:, None] * np.array([1.5, 0.05, -0.8]) * (10.0 / F) + rng.normal(size=(F, 3)) * 0.3
    pos = np.repeat(offsets[None], F, axis=0); pos[:, 0] = root
    gp = sio.positions_global(sio.Motion(rot, pos, np.tile([1.0, 0, 0, 0], (nj, 1)), offsets, PARENTS))
    gabs = gp[:, kopt.BACKWARD_MAPPING]
    p3 = gabs - root[:, None] + rng.normal(size=gabs.shape) * 1.5
    p3[:, kopt.ROOT_IDX] = 0.0
    focal = np.array([2000.0, 2000.0])
    p2 = np.stack([focal[0] * gabs[..., 0] / gabs[..., 2] + 960.0, focal[1] * gabs[..., 1] / gabs[..., 2] + 540.0], axis=2) + rng.normal(size=(F, nj, 2)) * 3.0
    conf = rng.uniform(0.3, 1.0, size=(F, nj)); conf[rng.uniform(size=(F, nj)) < 0.05] = 0.0
    p2[:, 25:] = 0.0; conf[:, 25:] = 0.0
    ang = 2.0 * np.arccos(np.clip(rot[..., 0], -1, 1))
    ax = rot[..., 1:] / np.maximum(np.linalg.norm(rot[..., 1:], axis=-1, keepdims=True), 1e-12)
    vel = np.zeros((F, nj))
    vel[:half + 1, 19] = 1; vel[:half + 1, 20] = 1; vel[:half, 21] = 1
    vel[half:, 22] = 1; vel[half:, 23] = 1; vel[half + 1:, 24] = 1
    vel[half + (F - half) // 2, 21] = 1
    return dict(poses2D=p2, joint_conf_2d=conf, poses3D=p3, root_pos=root + rng.normal(size=root.shape), joint_angles=-(ax * ang[..., None]) + rng.normal(size=(F, nj, 3)) * 0.03,
                offsets=OFFSETS, parents=PARENTS, ppx=960.0, ppy=540.0, camFocal=focal, velConstraints=vel)


if __name__ == '__main__':
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    F = int(sys.argv[2]) if len(sys.argv) > 2 else 100
    FO = int(sys.argv[3]) if len(sys.argv) > 3 else 12
    clips = [make_clip(s, F) for s in range(B)]
    opt = kopt.KinematicOptimizer(device=0)
    if os.environ.get('KIN_THREADS'):
        opt.kin.cfg.reserved[0] = int(os.environ['KIN_THREADS'])
    opt.optimize([make_clip(10_000, 8)])                      # warm-up (module load)
    kin_ms = []
    real_solve = opt.kin.solve

    def timed(problems):
        r = real_solve(problems)
        kin_ms.append(opt.kin.last_kernel_ms())
        return r

    opt.kin.solve = timed
    t0 = time.perf_counter(); res = opt.optimize(clips); t1 = time.perf_counter()
    ik_ms, ik_frames = opt.ik.last_kernel_ms()
    its = np.array([[s['lsmr_iterations'] for s in r['stages']] for r in res])
    nfev = np.array([[s['nfev'] for s in r['stages']] for r in res])
    n, m = 87 * F, 507 * F - 423
    # algorithmic bytes per LSMR iteration ("kinematic optimisation"): u (m) read + written, v / h / hbar / x (n each) read + written, the
    # linearisation (420 F doubles) read twice (J v, J^T u); per clip
    alg_bytes = 8.0 * (2 * m + 8 * n + 2 * 420 * F) * its.sum()
    out = dict(clips=B, frames=F, unknowns=n, rows=m, wall_s=t1 - t0, clips_per_s=B / (t1 - t0), ik_kernel_ms=ik_ms, lsq_kernel_ms=kin_ms,
               lsmr_iterations_per_clip=float(its.sum(axis=1).mean()), nfev_per_stage=nfev.mean(axis=0).tolist(),
               status_counts={str(k): int(v) for k, v in zip(*np.unique([s['status'] for r in res for s in r['stages']], return_counts=True))},
               algorithmic_GBps=alg_bytes / (sum(kin_ms) * 1e-3) / 1e9, us_per_lsmr_iteration_per_workgroup=1e3 * sum(kin_ms) / max(1.0, its.sum() / min(B, 256 * 2)),
               relabelled_contacts_per_clip=float(np.mean([np.abs(r['velConstraints'] - c['velConstraints']).sum() for r, c in zip(res, clips)])))
    # CPU: the oracle (dense restatement of the reference, with SciPy's sparse products) on a short clip, scaled per frame
    if FO > 0:
        from oracle import kinopt_oracle as ko      # checker / CPU baseline only
        c = make_clip(0, FO)
        t2 = time.perf_counter()
        ko.optimize_trajectory(c['poses2D'], c['joint_conf_2d'], c['poses3D'], c['root_pos'], c['joint_angles'], c['offsets'], c['parents'], (c['ppx'], c['ppy']), c['camFocal'], c['velConstraints'])
        out['oracle_s_for_%d_frames_one_core' % FO] = time.perf_counter() - t2
    print(json.dumps(out))

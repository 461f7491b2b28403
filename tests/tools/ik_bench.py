"""IK back-projection on one MI355X: batch of synthetic 33-joint skeletons (13 targets, 90 frames each) through
libchd_ik.so, checked against the numpy oracle on the first video and timed against it.

    python tests/tools/ik_bench.py [videos=128] [frames=90]

Measurements: profiles/r02a_round_start/ik_bench.log (first MI355X run), profiles/r02k_final/ik_bench.log (end of round 2)."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import chd_amd  # noqa: E402,F401
from chd_amd.ik_backproject import IkBackProject  # noqa: E402
from oracle import ik_oracle as ik  # noqa: E402  (checker / CPU baseline only)

PARENTS = [-1, 0, 1, 2, 3, 4, 0, 6, 7, 8, 9, 0, 11, 12, 13, 14, 13, 16, 17, 18, 19, 20, 13, 22, 23, 24, 25, 26, 3, 8, 15, 21, 27]
TARGETS = [0, 11, 12, 13, 14, 15, 16, 22, 19, 25, 4, 9, 28]


def make_video(seed, F):
    rng = np.random.default_rng(seed)
    J = len(PARENTS)
    offsets = rng.normal(size=(J, 3)) * 10.0
    offsets[0] = 0.0
    eul = np.cumsum(rng.normal(size=(F, J, 3)) * 0.02, axis=0) + rng.normal(size=(1, J, 3)) * 0.3      # smooth motion
    pos = np.repeat(offsets[None], F, axis=0).copy()
    pos[:, 0] = np.cumsum(rng.normal(size=(F, 3)) * 0.5, axis=0)
    rot = ik.quat_from_euler_xyz_world(eul)
    gp = ik.positions_global(rot, pos, np.array(PARENTS))
    tg = np.stack([gp[:, j] + rng.normal(size=(F, 3)) * 2.0 for j in TARGETS], axis=0)
    return dict(parents=np.array(PARENTS), target_joints=np.array(TARGETS), targets=tg, rot=rot, pos=pos)


if __name__ == '__main__':
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
    F = int(sys.argv[2]) if len(sys.argv) > 2 else 90
    vids = [make_video(s, F) for s in range(B)]
    solver = IkBackProject(device=0)
    solver.solve(vids[:2])                                   # warm-up (module load, allocations)
    t0 = time.perf_counter(); outs = solver.solve(vids); t1 = time.perf_counter()
    t2 = time.perf_counter(); ro, po = ik.ik_ck(vids[0]['rot'], vids[0]['pos'], vids[0]['parents'], vids[0]['target_joints'], vids[0]['targets']); t3 = time.perf_counter()
    gp_gpu = ik.positions_global(outs[0][0], outs[0][1], vids[0]['parents'])
    gp_cpu = ik.positions_global(ro, po, vids[0]['parents'])
    err = np.abs(gp_gpu - gp_cpu).max()
    ms, frames = solver.last_kernel_ms()
    J, T, its = len(PARENTS), len(TARGETS), int(solver.cfg.iterations)
    # algorithmic HBM bytes per (frame, iteration): state of the frame and its two neighbours in (3 x 7J doubles), state out
    # (7J), targets (3T) -- DESIGN.md "Next row"; flops: J J^T blocks + LDL^T (R^3/3) + J^T y, counted as fused multiply-adds x 2
    R = 3 * T
    bytes_alg = 8.0 * (4 * 7 * J + 3 * T) * frames * its
    flops = 2.0 * (T * (T + 1) / 2 * 6 * 3 * 27 + R ** 3 / 3.0 + 6 * J * T * 9) * frames * its
    print('kernel time %.2f ms for %d launches of %d workgroups: %.1f us per launch, %.2f us per frame-step per CU slot; '
          'algorithmic %.1f GB/s (of ~8000), ~%.2f TFLOP/s fp64 (vector peak 78.6)'
          % (ms, its, frames, 1e3 * ms / its, 1e3 * ms / its / max(1.0, frames / 256.0), bytes_alg / (ms * 1e-3) / 1e9, flops / (ms * 1e-3) / 1e12))
    print('videos %d frames %d: GPU %.3f s (%.1f videos/s, host buffers in/out included); oracle %.2f s per video on one core; '
          'max |global joint position difference| on video 0: %.2e' % (B, F, t1 - t0, B / (t1 - t0), t3 - t2, err))
    assert err < 1e-6

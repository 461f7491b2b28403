"""Host side of the physics stage with the reference's command line.

Reference: ``scripts/run_phys_mocap.py`` — for every video directory it runs (3) kinematic optimisation,
(4) retargeting, writes ``phys_optim_in_<char>/`` and then starts ``./phys_optim`` once per video
(:159-174).  This driver keeps the flags that concern the physics stage (:13-31, :33-44) and replaces the
per-video child process by ONE batched call into ``libchd_phys.so`` over all directories (sharded over the
GPUs of the node when launched with torch.distributed.run).  The upstream stages (kinematic optimisation,
retargeting, ``towr_utils.prepare_input``) and the IK back-projection are outside this path:
the directories must already contain ``phys_optim_in_<character>/``.
"""
import argparse
import os
import sys

from . import sharding
from .phys_optim import PhysOptim, default_config


def parse_args(argv):
    p = argparse.ArgumentParser()
    p.add_argument('--data', required=True, help='root directory with one sub-directory per video (run_phys_mocap.py:14)')
    p.add_argument('--character', default='ybot', help='run_phys_mocap.py:16')
    p.add_argument('--nframes', type=int, default=None, help='frames per video; default: number of OpenPose JSON files (:97) '
                                                              'or the line count implied by motion_info.txt')
    # PhysOptimParsms (:33-44) / gflags of phys_optim (phys_optim.cpp:27-31)
    p.add_argument('--w-com-lin', type=float, default=0.4)
    p.add_argument('--w-com-ang', type=float, default=1.7)
    p.add_argument('--w-ee', type=float, default=0.3)
    p.add_argument('--w-smooth', type=float, default=0.1)
    p.add_argument('--w-dur', type=float, default=0.1)
    p.add_argument('--batch', type=int, default=128, help='sequences per kernel launch')
    return p.parse_args(argv)


def count_frames(video_dir, in_dir):
    op = os.path.join(video_dir, 'openpose_result')
    if os.path.isdir(op):
        n = len([f for f in os.listdir(op) if f.endswith('.json')])     # run_phys_mocap.py:97
        if n > 0:
            return n
    with open(os.path.join(in_dir, 'motion_info.txt')) as f:
        return (len(f.read().split()) - 1) // 18


def main(argv=None):
    a = parse_args(sys.argv[1:] if argv is None else argv)
    vids = sorted(d for d in os.listdir(a.data) if os.path.isdir(os.path.join(a.data, d)) and not d.startswith('.'))
    jobs = []
    for v in vids:
        vd = os.path.join(a.data, v)
        ind = os.path.join(vd, 'phys_optim_in_' + a.character)
        if not os.path.isdir(ind):
            print('[run_phys_mocap] %s: no %s, skipping' % (v, os.path.basename(ind)))
            continue
        outd = os.path.join(vd, 'phys_optim_out_' + a.character)
        os.makedirs(outd, exist_ok=True)                                   # run_phys_mocap.py:156-158
        jobs.append((ind, outd, a.nframes or count_frames(vd, ind)))
    rank, world, local = sharding.rank_world()
    mine = sharding.my_shard([j[2] for j in jobs])
    cfg = default_config(w_com_lin=a.w_com_lin, w_com_ang=a.w_com_ang, w_ee=a.w_ee, w_smooth=a.w_smooth, w_dur=a.w_dur)
    solver = PhysOptim(device=local, config=cfg)
    bad = 0
    for s in range(0, len(mine), a.batch):
        part = [jobs[i] for i in mine[s:s + a.batch]]
        st = solver.solve_dirs([p[0] for p in part], [p[1] for p in part], [p[2] for p in part])
        bad += sum(1 for x in st if x != 0)
    solver.close()
    print('[run_phys_mocap] rank %d/%d: %d sequences, %d I/O failures' % (rank, world, len(mine), bad))
    return 0 if bad == 0 else 1


if __name__ == '__main__':
    sys.exit(main())

"""Host side of the physics stage with the reference's command line.

Reference: ``scripts/run_phys_mocap.py`` — for every video directory it runs (3) kinematic optimisation,
(4) retargeting, writes ``phys_optim_in_<char>/`` and then starts ``./phys_optim`` once per video
(:159-174).  This driver keeps the flags that concern the physics stage (:13-31, :33-44) and replaces the
per-video child process by ONE batched call into ``libchd_phys.so`` over all directories (sharded over the
GPUs of the node when launched with torch.distributed.run).  Retargeting is outside this path.  The stages
either side of the solver are optional here:

* ``--kinematic`` runs the kinematic optimisation in front (chd_amd.run_kinematic_optimizer: all videos in one batched solve) and hands
  its ``final_test.bvh`` on as ``combined_out.bvh`` (run_phys_mocap.py:103-131 for ``--character combined``; other characters need the
  reference's re-targeting step, which is outside this path);
* ``--prepare`` writes ``phys_optim_in_<character>/`` from ``kinematic_results/{<character>_out.bvh, floor_out.txt,
  foot_contacts.npy}`` (the ``towr_utils.py --anim ... --out ...`` child process of :137-150; `prepare_input.py`);
  without it the directories must already contain ``phys_optim_in_<character>/``;
* ``--out-bvh`` back-projects the three solution files onto the skeleton and writes
  ``<video>_<character>_{no_dynamics,dynamics,durations}.bvh`` (the ``towr_utils.py --viz --out-bvh`` child process of
  :180-201 without its plots / video; `apply_results.py`, one batched IK launch sequence per solution kind).

``--prepare`` and ``--out-bvh`` need the character's joint / segment tables as a JSON file (``--character-json``, fields of
``apply_results.Character``); none are baked in.
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
    p.add_argument('--batch', type=int, default=4096, help='most sequences handed to the library in one call (one persistent launch drains them all)')
    p.add_argument('--kinematic', action='store_true', help='run the kinematic optimisation first (run_phys_mocap.py:103-131): every video directory with openpose_result/, '
                   'tracked_results.json and foot_contacts.npy gets kinematic_results/; all videos in one batched solve')
    p.add_argument('--skel-path', default='skeleton_fitting/combined_body_25.bvh', help='template of the combined skeleton for --kinematic (run_phys_mocap.py:106)')
    p.add_argument('--prepare', action='store_true', help='write phys_optim_in_<character>/ from kinematic_results/ first')
    p.add_argument('--prepare-device', action='store_true', help='with --prepare: the BVH files of all videos read by the native parser and their per-frame numerics in ONE launch of the '
                   'HIP kernel of libchd_prepare.so (prepare_input.prepare_sequences_device) instead of NumPy video by video')
    p.add_argument('--out-bvh', action='store_true', help='back-project the solutions onto the skeleton and write BVH files')
    p.add_argument('--character-json', default=None, help='joint / segment tables of the character (apply_results.Character)')
    p.add_argument('--fps', type=float, default=30.0, help='frame rate of the animation (the reference reads it from the video, :90-91)')
    return p.parse_args(argv)


def count_frames(video_dir, in_dir):
    op = os.path.join(video_dir, 'openpose_result')
    if os.path.isdir(op):
        n = len([f for f in os.listdir(op) if f.endswith('.json')])     # run_phys_mocap.py:97
        if n > 0:
            return n
    if in_dir is None:
        return None                                                         # --prepare without OpenPose results: the whole BVH
    with open(os.path.join(in_dir, 'motion_info.txt')) as f:
        return (len(f.read().split()) - 1) // 18


LAST_TIMINGS = {}          # seconds per stage of the last main() call: kinematic / prepare / physics / back_projection (pipeline benchmark)


def main(argv=None):
    a = parse_args(sys.argv[1:] if argv is None else argv)
    vids = sorted(d for d in os.listdir(a.data) if os.path.isdir(os.path.join(a.data, d)) and not d.startswith('.'))
    jobs = []
    character = None
    if a.prepare or a.out_bvh:
        if not a.character_json:
            raise SystemExit('--prepare / --out-bvh need --character-json')
        from .apply_results import Character
        character = Character.from_json(a.character_json)
    import time as _time
    LAST_TIMINGS.clear()
    _t = _time.perf_counter()
    if a.kinematic:
        # run_phys_mocap.py:103-131: kinematic optimisation per video, then "re-targeting to character if needed".  Re-targeting
        # (skeleton_fitting/combined_to_mixamo.py) is outside this path: only the combined skeleton itself goes straight on (:129-131)
        if a.character != 'combined':
            raise SystemExit("--kinematic produces the combined skeleton's final_test.bvh; any other --character needs the reference's re-targeting step in between")
        if int(os.environ.get('WORLD_SIZE', '1')) > 1:
            raise SystemExit('--kinematic writes kinematic_results/: run it once as a single process (or python -m chd_amd.run_kinematic_optimizer under torch.distributed.run)')
        import shutil
        from . import run_kinematic_optimizer as rko
        kvids = [v for v in vids if all(os.path.exists(os.path.join(a.data, v, q)) for q in ('openpose_result', 'tracked_results.json', 'foot_contacts.npy'))]
        kdirs = [os.path.join(a.data, v) for v in kvids]
        kouts = [os.path.join(d, 'kinematic_results') for d in kdirs]
        res = rko.optimize_videos(kdirs, kouts, a.skel_path, 0, [a.nframes or count_frames(d, None) for d in kdirs])
        kin_failed = []
        for v, o, r in zip(kvids, kouts, res):
            if r.get('error'):
                print('[run_phys_mocap] %s: kinematic optimisation failed -- %s' % (v, r['error']))
                kin_failed.append(v)
            else:
                shutil.copyfile(os.path.join(o, 'final_test.bvh'), os.path.join(o, a.character + '_out.bvh'))
    LAST_TIMINGS['kinematic'] = _time.perf_counter() - _t; _t = _time.perf_counter()
    prep_batch = []
    for v in vids:
        vd = os.path.join(a.data, v)
        ind = os.path.join(vd, 'phys_optim_in_' + a.character)
        if a.prepare and not os.path.exists(os.path.join(vd, 'kinematic_results', a.character + '_out.bvh')):
            print('[run_phys_mocap] %s: no kinematic_results/%s_out.bvh, skipping' % (v, a.character))
            continue
        if a.prepare and int(os.environ.get('WORLD_SIZE', '1')) > 1:
            raise SystemExit('--prepare writes the input directories: run it once as a single process, then launch the ranks')
        if a.prepare and not a.prepare_device:
            from .prepare_input import prepare_input
            kin = os.path.join(vd, 'kinematic_results')
            n = a.nframes or count_frames(vd, None)
            prepare_input(os.path.join(kin, a.character + '_out.bvh'), os.path.join(kin, 'floor_out.txt'), os.path.join(kin, 'foot_contacts.npy'),
                          ind, character, start_idx=0, end_idx=n, dt=1.0 / a.fps)
        elif a.prepare:
            prep_batch.append((vd, ind, a.nframes or count_frames(vd, None)))
            continue
        if not os.path.isdir(ind):
            print('[run_phys_mocap] %s: no %s, skipping' % (v, os.path.basename(ind)))
            continue
        outd = os.path.join(vd, 'phys_optim_out_' + a.character)
        os.makedirs(outd, exist_ok=True)                                   # run_phys_mocap.py:156-158
        jobs.append((ind, outd, a.nframes or count_frames(vd, ind)))
    if prep_batch:                                   # --prepare-device: every video of the run in one batch of tensor operations
        import numpy as np
        from . import io_formats as iof
        from . import prepare_input as pi
        from . import skeleton_io as sk
        kins = [os.path.join(vd, 'kinematic_results') for vd, _, _ in prep_batch]
        from . import prepare_capi
        motions = [m for m, _, _ in prepare_capi.load_bvh_batch([os.path.join(k, a.character + '_out.bvh') for k in kins])]      # native reader, all files on the host's cores
        seqs = pi.prepare_sequences_device(motions, [pi.read_floor(os.path.join(k, 'floor_out.txt')) for k in kins],
                                           [np.load(os.path.join(k, 'foot_contacts.npy')) for k in kins], character,
                                           starts=[0] * len(kins), ends=[n for _, _, n in prep_batch], dt=1.0 / a.fps, device='cuda:%d' % sharding.rank_world()[2])
        for (vd, ind, n), seq in zip(prep_batch, seqs):
            iof.write_inputs(seq, ind)
            outd = os.path.join(vd, 'phys_optim_out_' + a.character)
            os.makedirs(outd, exist_ok=True)
            jobs.append((ind, outd, seq.F))
    LAST_TIMINGS['prepare'] = _time.perf_counter() - _t; _t = _time.perf_counter()
    rank, world, local = sharding.rank_world()
    mine = sharding.my_shard([j[2] for j in jobs])
    cfg = default_config(w_com_lin=a.w_com_lin, w_com_ang=a.w_com_ang, w_ee=a.w_ee, w_smooth=a.w_smooth, w_dur=a.w_dur)
    solver = PhysOptim(device=local, config=cfg)
    bad = len(kin_failed) if a.kinematic else 0          # videos the kinematic optimisation dropped count as failures of the run
    for s in range(0, len(mine), a.batch):
        part = [jobs[i] for i in mine[s:s + a.batch]]
        st = solver.solve_dirs([p[0] for p in part], [p[1] for p in part], [p[2] for p in part])
        bad += sum(1 for x in st if x != 0)
    solver.close()
    LAST_TIMINGS['physics'] = _time.perf_counter() - _t; _t = _time.perf_counter()
    if a.out_bvh:
        from . import apply_results as ar
        from .ik_backproject import IkBackProject
        ik = IkBackProject(device=local)
        parsed = {}                                                          # the input animations, read once for the three kinds
        for kind in ('no_dynamics', 'dynamics', 'durations'):               # run_phys_mocap.py:183-186
            part = [jobs[i] for i in mine if os.path.exists(os.path.join(jobs[i][1], 'sol_out_%s.txt' % kind))]
            if not part:
                continue
            vdirs = [os.path.dirname(p[1]) for p in part]
            ar.apply_results_batch([os.path.join(p[1], 'sol_out_%s.txt' % kind) for p in part],
                                   [os.path.join(vd, 'kinematic_results', a.character + '_out.bvh') for vd in vdirs],
                                   [os.path.join(p[1], '%s_%s_%s.bvh' % (os.path.basename(vd), a.character, kind)) for p, vd in zip(part, vdirs)],
                                   character, ik, starts=[0] * len(part), ends=[p[2] for p in part], animations=parsed)
    LAST_TIMINGS['back_projection'] = _time.perf_counter() - _t
    print('[run_phys_mocap] rank %d/%d: %d sequences, %d failed (unreadable inputs, rejected at set-up or unwritable outputs: each loses only itself)' % (rank, world, len(mine), bad))
    return 0 if bad == 0 else 1


if __name__ == '__main__':
    sys.exit(main())

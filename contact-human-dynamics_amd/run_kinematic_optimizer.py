"""``python -m chd_amd.run_kinematic_optimizer`` -- the reference's ``optimize/kinematic_optimizer.py`` (src/optimize/
kinematic_optimizer.py:30-224, started per video by scripts/run_phys_mocap.py:103-115) for one video with the reference's own
flags, or for every video directory under ``--data`` in ONE batched solve (IK initialisation + two least-squares launches for
all videos, chd_amd.kinematic_optimizer).

Per video directory it reads ``openpose_result/*.json`` (2D keypoints + confidences), ``tracked_results.json`` (monocular total
capture: root translation, body-25 and SMPL joints, SMPL joint angles), ``foot_contacts.npy`` (contact network output) and, with
``--gt-floor``, ``floor_gt.txt``; it writes ``<output>/foot_contacts.npy`` (relabelled, F x 4), ``floor_out.txt`` and
``final_test.bvh`` -- what ``towr_utils.prepare_input`` / the re-targeting step read next.  Visualisation flags are accepted
and ignored (no display here).  With ``torch.distributed`` (``python -m torch.distributed.run --nproc-per-node N ...``) the
videos are sharded over the ranks, one GPU each; there is no collective on the data path."""
import argparse
import os
import sys

import numpy as np

from . import kinematic_optimizer as kopt
from . import skeleton_io as sio
from . import totalcap_io as tc
from .contact_net import load_keypoint_dir

MTC_FOCAL_LENGTH = (2000.0, 2000.0)          # kinematic_optimizer.py:27-28: the intrinsics monocular total capture assumes
MTC_PP = (1920 / 2, 1080 / 2)


def load_clip(video_dir, skeleton, start=0, end=100, use_gt_floor=False, keypoints=None, totalcap=None):
    """kinematic_optimizer.py:43-153 up to the call of optimize_trajectory -> the clip dict KinematicOptimizer.optimize takes.
    `keypoints` / `totalcap`: the two JSON inputs already read (optimize_videos reads them for all videos at once through the native readers)."""
    openpose_dir = os.path.join(video_dir, 'openpose_result')
    totalcap_path = os.path.join(video_dir, 'tracked_results.json')
    contacts_path = os.path.join(video_dir, 'foot_contacts.npy')
    for p, what in ((openpose_dir, 'openpose results'), (totalcap_path, 'total capture results'), (contacts_path, 'foot contact labels')):
        if not os.path.exists(p):
            raise FileNotFoundError('Could not find %s in %s' % (what, video_dir))
    kp = keypoints if keypoints is not None else load_keypoint_dir(openpose_dir)         # F x 25 x 3
    res = totalcap if totalcap is not None else tc.load_totalcap_results(totalcap_path)
    body25_root, body25_3d = tc.normalize_root_pos(res.root_trans, res.joint3d)
    _, smpl_3d = tc.normalize_root_pos(res.root_trans, res.smpl_joint3d, root_idx=tc.SMPL_ROOT_IDX)
    poses3d = tc.create_combined_model(body25_3d, smpl_3d)[start:end]
    F = poses3d.shape[0]
    poses2d = np.concatenate([kp[start:end, :, :2], np.zeros((F, 3, 2))], axis=1)       # :93-96: the spine joints have no 2D detection
    conf = np.concatenate([kp[start:end, :, 2], np.zeros((F, 3))], axis=1)
    fc = np.load(contacts_path)[start:end]                                                # [l_heel, l_toe, r_heel, r_toe]
    vel = np.zeros((F, poses3d.shape[1]))
    vel[:, 19] = fc[:, 1]; vel[:, 20] = fc[:, 1]; vel[:, 21] = fc[:, 0]                  # :111-117
    vel[:, 22] = fc[:, 3]; vel[:, 23] = fc[:, 3]; vel[:, 24] = fc[:, 2]
    clip = dict(poses2D=poses2d, joint_conf_2d=conf, poses3D=poses3d, root_pos=body25_root[start:end].copy(),
                joint_angles=tc.combined_angles_from_smpl(res.smpl_joint_angles[start:end]), offsets=skeleton.offsets, parents=skeleton.parents,
                ppx=MTC_PP[0], ppy=MTC_PP[1], camFocal=np.array(MTC_FOCAL_LENGTH), velConstraints=vel)
    if use_gt_floor:                                                                      # :121-131
        with open(os.path.join(video_dir, 'floor_gt.txt')) as fh:
            clip['plane_normal'] = np.array([float(x) for x in fh.readline().split(' ')])
            clip['plane_point'] = np.array([float(x) for x in fh.readline().split('\n')[0].split(' ')]) * 100.0      # to cm
    return clip


def optimize_videos(video_dirs, out_dirs, skel_path, start=0, ends=None, use_gt_floor=False, optimizer=None, device=0):
    """All videos in one batched solve.  Returns the per-video results of KinematicOptimizer.optimize."""
    skeleton, names, _ = sio.load_bvh(skel_path)
    ends = ends if ends is not None else [100] * len(video_dirs)
    for d in video_dirs:                                                                   # (the reference's messages, before anything is read)
        for p, what in ((os.path.join(d, 'openpose_result'), 'openpose results'), (os.path.join(d, 'tracked_results.json'), 'total capture results')):
            if not os.path.exists(p):
                raise FileNotFoundError('Could not find %s in %s' % (what, d))
    from . import prepare_capi                                                             # both JSON inputs of ALL videos on the host's cores (libchd_prepare.so)
    kps = prepare_capi.load_keypoint_dirs([os.path.join(d, 'openpose_result') for d in video_dirs])
    tcs = prepare_capi.load_totalcap_batch([os.path.join(d, 'tracked_results.json') for d in video_dirs])
    clips = [load_clip(d, skeleton, start, e, use_gt_floor, kp, res) for d, e, kp, res in zip(video_dirs, ends, kps, tcs)]
    opt = optimizer if optimizer is not None else kopt.KinematicOptimizer(device=device, parents=skeleton.parents)
    results = opt.optimize(clips)
    for out, r in zip(out_dirs, results):
        if not r.get('error'):
            kopt.save_results(out, r, names)
    return results


def main(argv=None):
    ap = argparse.ArgumentParser(description='kinematic optimisation (reference: src/optimize/kinematic_optimizer.py)')
    ap.add_argument('--input_path', default=None, help='path of the source video: its directory holds openpose_result/, tracked_results.json, foot_contacts.npy')
    ap.add_argument('--output_path', default=None)
    ap.add_argument('--data', default=None, help='batch mode: every sub-directory with the three inputs is a video; outputs go to <video>/kinematic_results')
    ap.add_argument('--skel_path', default='skeleton_fitting/combined_body_25.bvh')
    ap.add_argument('--start', type=int, default=0)
    ap.add_argument('--end', type=int, default=100, help='single video: last frame; batch mode: ignored (the number of OpenPose files, as run_phys_mocap.py:97 passes it)')
    ap.add_argument('--gt-floor', dest='use_gt_floor', action='store_true')
    ap.add_argument('--visualize', action='store_true'); ap.add_argument('--viz-only', dest='viz_only', action='store_true')
    ap.add_argument('--character', default='ybot')
    ap.add_argument('--device', type=int, default=None)
    args = ap.parse_args(argv)
    rank, world = int(os.environ.get('RANK', 0)), int(os.environ.get('WORLD_SIZE', 1))
    device = args.device if args.device is not None else int(os.environ.get('LOCAL_RANK', 0))
    if args.data:
        names = sorted(d for d in os.listdir(args.data) if all(os.path.exists(os.path.join(args.data, d, p)) for p in ('openpose_result', 'tracked_results.json', 'foot_contacts.npy')))
        dirs = [os.path.join(args.data, d) for d in names][rank::world]
        outs = [os.path.join(d, 'kinematic_results') for d in dirs]
        ends = [len([f for f in os.listdir(os.path.join(d, 'openpose_result')) if f.endswith('.json')]) for d in dirs]
    elif args.input_path and args.output_path:
        dirs = [os.path.dirname(args.input_path)] if rank == 0 else []
        outs = [args.output_path] if rank == 0 else []
        ends = [args.end]
    else:
        ap.error('give --data, or --input_path and --output_path')
    n_failed = 0
    if dirs:
        res = optimize_videos(dirs, outs, args.skel_path, args.start, ends, args.use_gt_floor, device=device)
        for d, r in zip(dirs, res):
            if r.get('error'):
                print('%s: FAILED -- %s' % (d, r['error']), flush=True)
                n_failed += 1
                continue
            print('%s: %d frames, cost %.4f (stage 1) / %.4f (with the floor), floor normal %s' % (d, r['pose3d'].shape[0], r['stages'][0]['cost'], r['stages'][1]['cost'],
                                                                                             np.round(r['plane_normal'], 4)), flush=True)
    print('Finished kinematic optimization!' if n_failed == 0 else 'Finished kinematic optimization: %d video(s) FAILED (no kinematic_results written for them)' % n_failed)
    return 1 if n_failed else 0          # a sharded / batched run sees dropped videos in the exit code


if __name__ == '__main__':
    sys.exit(main())

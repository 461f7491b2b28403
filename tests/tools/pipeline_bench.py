"""End to end on one GPU, the chain of scripts/run_phys_mocap.py:97-201 for the reference's `combined` character:

    openpose_result/*.json + tracked_results.json
      --run_detect_contacts (contact MLP, device ops)--> foot_contacts.npy
      --run_phys_mocap --kinematic (libchd_ik.so + libchd_kinopt.so)--> kinematic_results/{final_test.bvh, floor_out.txt, foot_contacts.npy}
      --prepare --prepare-device (BVH parse + batched tensor operations)--> phys_optim_in_combined/
      --libchd_phys.so--> sol_out_*.txt --out-bvh (libchd_ik.so)--> <video>_combined_*.bvh

on synthetic video directories (SUBSTITUTED inputs, SURVEY 8d config 4: no OpenPose / MTC / network weights offline -- the contact
network runs with seeded random weights, its labels are timed and then replaced by the clip's own alternating schedule so that the
later stages see a plausible gait).  Reports videos/s and the seconds of every stage: where the next round should go.

    python tests/tools/pipeline_bench.py [--videos 32] [--frames 60]
"""
import argparse
import json
import os
import shutil
import sys
import tempfile
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
TESTS = os.path.dirname(HERE)
ROOT = os.path.dirname(TESTS)
sys.path.insert(0, ROOT); sys.path.insert(0, TESTS)


def run(n_videos=32, frames=60, keep=None):
    import torch
    import chd_amd  # noqa: F401
    from chd_amd import contact_net as cn
    from chd_amd import run_detect_contacts, run_phys_mocap
    from chd_amd.synth import make_kin_clip
    from test_config4_gpu import COMBINED
    from test_kinopt_driver import write_skeleton, write_video_dir
    g = np.load(os.path.join(TESTS, 'golden', 'kinopt_golden.npz'))
    tmp = keep or tempfile.mkdtemp(prefix='chd_pipe_')
    root = os.path.join(tmp, 'data'); os.makedirs(root)
    rng = np.random.default_rng(0)
    t0 = time.perf_counter()
    contacts = {}
    for i in range(n_videos):
        v = 'video_%03d' % i
        write_video_dir(os.path.join(root, v), make_kin_clip(i, frames, g['c0_skel_offsets'], g['c0_skel_parents'], upright=True), rng)
        contacts[v] = np.load(os.path.join(root, v, 'foot_contacts.npy'))
    write_skeleton(os.path.join(tmp, 'skel.bvh'))
    cj = os.path.join(tmp, 'combined.json'); json.dump(COMBINED, open(cj, 'w'))
    torch.manual_seed(0)
    weights = os.path.join(tmp, 'w.pth')
    torch.save(cn.randomize_batchnorm_stats(cn.OpenPoseModel(), seed=0).state_dict(), weights)
    t_make = time.perf_counter() - t0
    stages = {}
    if torch.cuda.is_available():                  # (context creation and the GEMM libraries' first use are the process's, not the stage's: 0.2-1.3 s of what this stage showed when it
        from chd_amd.contact_net import synthetic_keypoints, detect_contacts_device      #  was the first GPU work of the process; inside bench.py the contact-net metric has run before)
        warm = cn.OpenPoseModel()
        detect_contacts_device([synthetic_keypoints(s, F=frames) for s in range(2)], warm, torch.device('cuda'))
        torch.cuda.synchronize()
    t = time.perf_counter()
    rc0 = run_detect_contacts.main(['--data', root, '--weights', weights, '--device-ops'])
    torch.cuda.synchronize()
    stages['contact_detection'] = time.perf_counter() - t
    for v, c in contacts.items():                      # (random-weight labels -> the clip's own schedule, untimed)
        np.save(os.path.join(root, v, 'foot_contacts.npy'), c)
    t = time.perf_counter()
    rc1 = run_phys_mocap.main(['--data', root, '--character', 'combined', '--kinematic', '--skel-path', os.path.join(tmp, 'skel.bvh'), '--prepare', '--prepare-device',
                               '--out-bvh', '--character-json', cj])
    total_phys_mocap = time.perf_counter() - t
    lt = dict(run_phys_mocap.LAST_TIMINGS)
    stages['kinematic_optimisation'] = lt.get('kinematic'); stages['prepare_input_incl_bvh_parse'] = lt.get('prepare')
    stages['physics_incl_file_io'] = lt.get('physics'); stages['ik_back_projection_incl_bvh_write'] = lt.get('back_projection')
    n_bvh = sum(1 for v in contacts for k in ('no_dynamics', 'dynamics', 'durations')
                if os.path.exists(os.path.join(root, v, 'phys_optim_out_combined', '%s_combined_%s.bvh' % (v, k))))
    total = stages['contact_detection'] + total_phys_mocap
    if keep is None:
        shutil.rmtree(tmp, ignore_errors=True)
    return {'videos': n_videos, 'frames': frames, 'videos_per_s': n_videos / total, 'seconds_total': total, 'seconds_per_stage': stages,
            'bvh_files_written': n_bvh, 'driver_return_codes': [rc0, rc1], 'seconds_writing_the_synthetic_inputs_untimed': t_make,
            'note': 'one process, one GPU, file to file in the reference\'s directory layout; inputs substituted (synthetic OpenPose / total-capture JSON, random-weight contact '
                    'network whose labels are replaced by the clip\'s schedule after being timed); re-targeting (combined_to_mixamo.py) is outside this path'}


if __name__ == '__main__':
    ap = argparse.ArgumentParser()
    ap.add_argument('--videos', type=int, default=32)
    ap.add_argument('--frames', type=int, default=60)
    ap.add_argument('--keep', default=None, help='keep the generated directory tree here (a fresh directory)')
    a = ap.parse_args()
    print(json.dumps(run(a.videos, a.frames, keep=a.keep)))

"""BASELINE configs[3] end to end on one GPU, SUBSTITUTED: the real videos need OpenPose / MTC and their
weights, none of which exist offline, so the chain starts from synthetic OpenPose-like JSON files and a seeded
random-weight contact network:

    openpose_result/*.json --run_detect_contacts (ROCm, device ops)--> foot_contacts.npy
        --run_phys_mocap --prepare--> phys_optim_in_<char>/ --libchd_phys.so--> sol_out_*.txt
        --run_phys_mocap --out-bvh (libchd_ik.so)--> <video>_<char>_*.bvh

with the reference's directory layout (scripts/run_detect_contacts.py:52-58, scripts/run_phys_mocap.py:97-201).  Every
stage's numerics is pinned elsewhere; this checks that the stages accept each other's output on the device, that the
labels equal the CPU labels bit for bit, and that both drivers run unmodified on a directory tree."""
import json
import os
import sys

import numpy as np
import pytest

import chd_amd  # noqa: F401

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def test_openpose_json_to_bvh_on_one_gpu(tmp_path):
    import torch
    sys.path.insert(0, os.path.join(HERE, 'golden'))
    from make_apply_golden import CHARACTER
    from chd_amd import contact_net as cn
    from chd_amd import io_formats as iof
    from chd_amd import run_detect_contacts, run_phys_mocap
    from chd_amd import skeleton_io as sk
    g = np.load(os.path.join(HERE, 'golden', 'apply_golden.npz'))
    F = 14                                                            # the fixture clip's length
    root = tmp_path / 'data'
    vids = ['dance_a', 'dance_b', 'dance_c']
    for k, v in enumerate(vids):
        op = root / v / 'openpose_result'; kin = root / v / 'kinematic_results'
        os.makedirs(op); os.makedirs(kin)
        kp = cn.synthetic_keypoints(k, F=F)
        for i in range(F):                                            # OpenPose's per-frame JSON (openpose_utils.py:48-76)
            json.dump({'version': 1.3, 'people': [{'pose_keypoints_2d': [float(x) for x in kp[i].reshape(-1)]}]},
                      open(op / ('%s_%012d_keypoints.json' % (v, i)), 'w'))
        open(kin / 'synth_out.bvh', 'wb').write(g['bvh_text'].tobytes())
        open(kin / 'floor_out.txt', 'wb').write(g['prep_floor_text'].tobytes())
    torch.manual_seed(0)
    model = cn.randomize_batchnorm_stats(cn.OpenPoseModel(), seed=0)
    weights = str(tmp_path / 'op_only_weights.pth')
    torch.save(model.state_dict(), weights)
    # ---- contact detection on the GPU (run_detect_contacts.py:52-58)
    assert run_detect_contacts.main(['--data', str(root), '--weights', weights, '--device-ops']) == 0
    cpu_labels, _ = cn.detect_contacts([cn.load_keypoint_dir(str(root / v / 'openpose_result')) for v in vids], model.eval(), torch.device('cpu'))
    for v, lab in zip(vids, cpu_labels):
        got = np.load(str(root / v / 'foot_contacts.npy'))
        assert got.shape == (F, 4) and np.array_equal(got, lab)       # bit-exact labels, GPU device ops vs CPU NumPy path
        # the physics stage reads kinematic_results/foot_contacts.npy (run_phys_mocap.py:146: the kinematic optimisation's refined
        # contacts, a stage outside this path -- here the network's own)
        np.save(str(root / v / 'kinematic_results' / 'foot_contacts.npy'), got)
    # ---- prepare_input -> physics -> back-projection (run_phys_mocap.py:137-201)
    cj = str(tmp_path / 'character.json')
    json.dump(CHARACTER, open(cj, 'w'))
    rc = run_phys_mocap.main(['--data', str(root), '--character', 'synth', '--prepare', '--out-bvh', '--character-json', cj])
    assert rc == 0
    for v in vids:
        out = root / v / 'phys_optim_out_synth'
        files = sorted(os.listdir(out))
        assert {'sol_out_no_dynamics.txt', 'sol_out_dynamics.txt', 'sol_out_durations.txt', 'success_log.txt'} <= set(files)
        sol = iof.load_results(str(out / 'sol_out_no_dynamics.txt'))
        assert sol.num_frames == F and np.isfinite(sol.base_lin).all() and np.isfinite(sol.ee_pos).all()
        bvh = str(out / ('%s_synth_no_dynamics.bvh' % v))
        assert os.path.exists(bvh)
        m, names, _ = sk.load_bvh(bvh)
        assert m.n_frames == F and m.n_joints == 20 and names[0] == 'Hips'
        log = open(str(out / 'success_log.txt')).read().split()
        assert log[0] == 'dynamics' and log[1] in '01' and log[2] == 'durations' and log[3] in '01'


# the reference's tables of its `combined` character (src/utils/character_info_utils.py:143-160, :181, :255-283): the skeleton the kinematic
# optimisation works on, which run_phys_mocap.py:129-131 hands to the physics stage without re-targeting
COMBINED = dict(toe_inds=[5, 11], ankle_inds=[3, 9], upper_body=[0] + list(range(13, 28)), heel_inds=[4, 10], left_leg_chain=[1, 2, 3, 5], hip_inds=[1, 7], mass=73.0,
                seg_to_joints={'head': [17], 'upper_trunk': [15, 16], 'mid_trunk': [14, 15], 'lower_trunk': [13, 14], 'left_upper_arm': [22, 23], 'left_forearm': [23, 24],
                               'left_hand': [24], 'left_thigh': [1, 2], 'left_shank': [2, 3], 'left_foot': [3, 4, 5, 6], 'right_upper_arm': [25, 26], 'right_forearm': [26, 27],
                               'right_hand': [27], 'right_thigh': [7, 8], 'right_shank': [8, 9], 'right_foot': [9, 10, 11, 12]},
                seg_to_mass_perc={'head': 6.94, 'upper_trunk': 15.96, 'mid_trunk': 16.33, 'lower_trunk': 11.17, 'left_upper_arm': 2.71, 'left_forearm': 1.62, 'left_hand': 0.61,
                                  'left_thigh': 14.16, 'left_shank': 4.33, 'left_foot': 1.37, 'right_upper_arm': 2.71, 'right_forearm': 1.62, 'right_hand': 0.61,
                                  'right_thigh': 14.16, 'right_shank': 4.33, 'right_foot': 1.37})


def test_kinematic_optimisation_to_bvh_on_one_gpu(tmp_path):
    """The chain of scripts/run_phys_mocap.py:97-201 for `--character combined`, every stage on the GPU libraries:

        openpose_result/*.json + tracked_results.json + foot_contacts.npy
            --kinematic (libchd_ik.so, libchd_kinopt.so)--> kinematic_results/{final_test.bvh = combined_out.bvh, floor_out.txt, foot_contacts.npy}
            --prepare --prepare-device (batched tensor operations)--> phys_optim_in_combined/ --libchd_phys.so--> sol_out_*.txt --out-bvh (libchd_ik.so)--> <video>_combined_*.bvh

    on synthetic clips of a standing-up person (y down, as monocular total capture delivers them)."""
    from chd_amd import io_formats as iof
    from chd_amd import run_phys_mocap
    from chd_amd import skeleton_io as sk
    from chd_amd.synth import make_kin_clip
    sys.path.insert(0, HERE)
    from test_kinopt_driver import write_skeleton, write_video_dir
    g = np.load(os.path.join(HERE, 'golden', 'kinopt_golden.npz'))
    rng = np.random.default_rng(8)
    root = tmp_path / 'data'
    frames = {'walk_a': 24, 'walk_b': 30}
    for i, (v, F) in enumerate(frames.items()):
        write_video_dir(str(root / v), make_kin_clip(i, F, g['c0_skel_offsets'], g['c0_skel_parents'], upright=True), rng)
    write_skeleton(str(tmp_path / 'skel.bvh'))
    cj = str(tmp_path / 'combined.json')
    json.dump(COMBINED, open(cj, 'w'))
    rc = run_phys_mocap.main(['--data', str(root), '--character', 'combined', '--kinematic', '--skel-path', str(tmp_path / 'skel.bvh'), '--prepare', '--prepare-device', '--out-bvh', '--character-json', cj])
    assert rc == 0
    for v, F in frames.items():
        kin = root / v / 'kinematic_results'
        assert open(str(kin / 'combined_out.bvh')).read() == open(str(kin / 'final_test.bvh')).read()
        assert np.load(str(kin / 'foot_contacts.npy')).shape == (F, 4)
        out = root / v / 'phys_optim_out_combined'
        sol = iof.load_results(str(out / 'sol_out_no_dynamics.txt'))
        assert sol.num_frames == F and np.isfinite(sol.base_lin).all() and np.isfinite(sol.ee_pos).all() and sol.ee_pos.shape[0] == 4
        log = open(str(out / 'success_log.txt')).read().split()
        assert log[0] == 'dynamics' and log[1] in '01' and log[2] == 'durations' and log[3] in '01'
        m, names, _ = sk.load_bvh(str(out / ('%s_combined_no_dynamics.bvh' % v)))
        assert m.n_frames == F and m.n_joints == 28 and names[0] == 'J00'       # the heels are joints of this skeleton: nothing appended, nothing removed


def test_prepare_input_batched_on_gpu(tmp_path):
    """prepare_input's per-frame numerics for a batch of clips of different lengths on the MI355X -- ONE launch of the HIP kernel of libchd_prepare.so over
    the frames of all clips (the default of prepare_sequences_device on a cuda device) -- against the NumPy mirror (which tests/test_prepare_input.py pins to
    the files the reference's own prepare_input wrote); then the tensor-operation form as a third opinion, and a batch of 4 000 frames for the kernel's rate."""
    sys.path.insert(0, HERE); sys.path.insert(0, os.path.join(HERE, 'golden'))
    from make_apply_golden import CHARACTER
    from chd_amd import apply_results as ar
    from chd_amd import prepare_capi as pc
    from chd_amd import prepare_input as pi
    from test_prepare_input import ARRAYS, _batch_inputs, check_device_batch
    gold = np.load(os.path.join(HERE, 'golden', 'apply_golden.npz'))
    character = ar.Character(**CHARACTER)
    check_device_batch(gold, tmp_path, character, 'cuda:0')
    assert pc.last_kernel_ms() > 0.0                                   # the library's kernel ran (not a fallback)
    clips, floors, fcs, starts, ends = _batch_inputs(gold, tmp_path)
    a = pi.prepare_sequences_device(clips, floors, fcs, character, starts, ends, device='cuda:0')
    b = pi.prepare_sequences_device(clips, floors, fcs, character, starts, ends, device='cuda:0', backend='torch')
    for x, y in zip(a, b):
        for k in ARRAYS:
            assert np.allclose(getattr(x, k), getattr(y, k), rtol=1e-11, atol=1e-13), k
    many = [clips[0]] * 300                                            # 300 clips x 14 frames in one launch
    got = pi.prepare_sequences_device(many, [floors[0]] * 300, [fcs[0]] * 300, character, device='cuda:0')
    assert all(np.array_equal(g.com, got[0].com) for g in got)
    print('prepare kernel: %d frames in %.3f ms' % (300 * clips[0].n_frames, pc.last_kernel_ms()))

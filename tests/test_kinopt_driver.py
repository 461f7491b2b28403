"""The kinematic optimisation's driver and ingest (chd_amd.run_kinematic_optimizer, chd_amd.totalcap_io): the reference's
`optimize/kinematic_optimizer.py` command line and the monocular-total-capture reader, pinned to what the reference's own
`totalcap_utils` functions return (tests/golden/make_totalcap_golden.py) and run end to end on synthetic video directories with
the host emulation of the two GPU solvers.  The GPU twin of the end-to-end part is in tests/test_kinopt_gpu.py."""
import json
import os
import sys

import numpy as np
import pytest

import chd_amd  # noqa: F401
from chd_amd import kinematic_optimizer as kopt
from chd_amd import run_kinematic_optimizer as drv
from chd_amd import skeleton_io as sio
from chd_amd import totalcap_io as tc
from chd_amd.synth import make_kin_clip

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from test_kinopt_emu import EmuIk, EmuKin      # noqa: E402
G = np.load(os.path.join(HERE, 'golden', 'kinopt_golden.npz'))


def test_totalcap_ingest_matches_the_reference(tmp_path):
    g = np.load(os.path.join(HERE, 'golden', 'totalcap_golden.npz'))
    path = tmp_path / 'tracked_results.json'
    path.write_bytes(g['json_text'].tobytes())
    res = tc.load_totalcap_results(str(path))
    for name in ('root_trans', 'joint3d', 'smpl_joint3d', 'smpl_joint_angles'):
        assert np.array_equal(getattr(res, name), g[name]), name
    assert res.body_coeffs.shape == (6, 30) and res.face_coeffs.shape == (6, 200)
    root, b3d = tc.normalize_root_pos(res.root_trans, res.joint3d)
    _, s3d = tc.normalize_root_pos(res.root_trans, res.smpl_joint3d, root_idx=tc.SMPL_ROOT_IDX)
    assert np.array_equal(root, g['body25_root_pos']) and np.array_equal(b3d, g['body25_3d']) and np.array_equal(s3d, g['smpl_3d'])
    assert np.array_equal(tc.create_combined_model(b3d, s3d), g['poses3D'])
    assert np.array_equal(tc.combined_angles_from_smpl(res.smpl_joint_angles), g['init_combined_joint_rot'])


def write_skeleton(path):
    m = sio.Motion(np.tile([1.0, 0, 0, 0], (1, 28, 1)), G['c0_skel_offsets'][None].copy(), np.tile([1.0, 0, 0, 0], (28, 1)), G['c0_skel_offsets'].copy(), G['c0_skel_parents'].copy())
    sio.save_bvh(path, m, ['J%02d' % j for j in range(28)])


def write_video_dir(d, clip, rng):
    """The three inputs of a video directory such that the driver's ingest reproduces `clip` (angles only where an SMPL joint exists)."""
    F = clip['poses2D'].shape[0]
    os.makedirs(os.path.join(d, 'openpose_result'))
    for f in range(F):
        kp = np.concatenate([clip['poses2D'][f, :25], clip['joint_conf_2d'][f, :25, None]], axis=1)
        with open(os.path.join(d, 'openpose_result', 'v_%012d_keypoints.json' % f), 'w') as fh:
            json.dump({'people': [{'pose_keypoints_2d': kp.reshape(-1).tolist()}]}, fh)
    frames = []
    for f in range(F):
        delta = rng.normal(size=3) * 5.0; eps = rng.normal(size=3) * 5.0
        spos = rng.normal(size=(22, 3)) * 30.0
        spos[0] = eps; spos[[3, 6, 9]] = clip['poses3D'][f, 25:28] + eps
        srot = rng.normal(size=(22, 3)) * 0.2
        for cj, sj in enumerate(tc.COMBINED_SKEL_TO_SMPL):
            if sj >= 0:
                srot[sj] = clip['joint_angles'][f, cj]
        frames.append({'trans': dict(zip('xyz', (clip['root_pos'][f] - delta).tolist())),
                       'joints': [{'pos': dict(zip('xyz', (clip['poses3D'][f, j] + delta).tolist()))} for j in range(25)],
                       'SMPLJoints': [{'pos': dict(zip('xyz', spos[j].tolist())), 'rot': dict(zip('xyz', srot[j].tolist()))} for j in range(22)],
                       'bodyCoeffs': [0.0] * 30, 'faceCoeffs': [0.0] * 200})
    with open(os.path.join(d, 'tracked_results.json'), 'w') as fh:
        json.dump({'totalcapResults': frames}, fh)
    v = clip['velConstraints']
    np.save(os.path.join(d, 'foot_contacts.npy'), np.stack([v[:, 21], v[:, 19], v[:, 24], v[:, 22]], axis=1).astype(int))


@pytest.fixture()
def data_root(tmp_path):
    rng = np.random.default_rng(4)
    clips = {}
    for name, seed, F in (('walk_a', 0, 9), ('walk_b', 1, 12)):
        clips[name] = make_kin_clip(seed, F, G['c0_skel_offsets'], G['c0_skel_parents'])
        write_video_dir(str(tmp_path / name), clips[name], rng)
    os.makedirs(str(tmp_path / 'not_a_video'))
    write_skeleton(str(tmp_path / 'skel.bvh'))
    return tmp_path, clips


def test_ingest_reproduces_the_clip(data_root):
    root, clips = data_root
    skel, _, _ = sio.load_bvh(str(root / 'skel.bvh'))
    for name, cl in clips.items():
        got = drv.load_clip(str(root / name), skel, 0, cl['poses2D'].shape[0])
        assert np.allclose(got['poses3D'], cl['poses3D'], atol=1e-10) and np.allclose(got['root_pos'], cl['root_pos'], atol=1e-10)
        assert np.allclose(got['poses2D'], cl['poses2D']) and np.allclose(got['joint_conf_2d'], cl['joint_conf_2d'])
        assert np.array_equal(got['velConstraints'], cl['velConstraints'])
        has = tc.COMBINED_SKEL_TO_SMPL >= 0
        assert np.allclose(got['joint_angles'][:, has], cl['joint_angles'][:, has]) and not got['joint_angles'][:, ~has].any()
        assert np.allclose(got['offsets'], G['c0_skel_offsets'], atol=1e-6) and np.array_equal(got['parents'], G['c0_skel_parents'])
    got = drv.load_clip(str(root / 'walk_b'), skel, 2, 9)                       # --start / --end clip every input alike
    assert got['poses3D'].shape[0] == 7 and np.allclose(got['root_pos'], clips['walk_b']['root_pos'][2:9], atol=1e-10)
    with pytest.raises(FileNotFoundError):
        drv.load_clip(str(root / 'not_a_video'), skel)


def test_batch_driver_writes_what_the_physics_stage_reads(data_root):
    root, clips = data_root
    opt = kopt.KinematicOptimizer(ik=EmuIk(), kin=EmuKin())
    dirs = [str(root / 'walk_a'), str(root / 'walk_b')]
    outs = [os.path.join(d, 'kinematic_results') for d in dirs]
    res = drv.optimize_videos(dirs, outs, str(root / 'skel.bvh'), 0, [9, 12], optimizer=opt)
    skel, _, _ = sio.load_bvh(str(root / 'skel.bvh'))
    direct = opt.optimize([drv.load_clip(d, skel, 0, e) for d, e in zip(dirs, (9, 12))])
    for out, r, ref in zip(outs, res, direct):
        assert np.array_equal(r['pose3d'], ref['pose3d'])                         # batch composition does not matter
        fc = np.load(os.path.join(out, 'foot_contacts.npy'))
        assert fc.shape == (r['pose3d'].shape[0], 4) and np.array_equal(fc, kopt.refined_contacts(r['velConstraints']))
        n, p = [np.array([float(v) for v in line.split(' ')]) for line in open(os.path.join(out, 'floor_out.txt')).read().split('\n')]
        assert np.allclose(n, r['plane_normal']) and np.allclose(p, r['plane_point']) and abs(np.linalg.norm(n) - 1) < 1e-12
        m, names, _ = sio.load_bvh(os.path.join(out, 'final_test.bvh'))
        assert names[0] == 'J00' and m.n_frames == r['pose3d'].shape[0]
        assert np.abs(sio.positions_global(m)[:, kopt.BACKWARD_MAPPING] - r['pose3d']).max() < 1e-3       # '%f' digits of the file


def test_chunked_two_thread_pipeline_gives_the_same_results():
    clips = [make_kin_clip(s, 8, G['c0_skel_offsets'], G['c0_skel_parents']) for s in range(5)]
    opt = kopt.KinematicOptimizer(ik=EmuIk(), kin=EmuKin(max_nfev=6, lsmr_maxiter=10))
    a = opt.optimize(clips)
    b = opt.optimize(clips, chunk=2, workers=2)
    assert len(b) == 5
    for x, y in zip(a, b):
        assert np.array_equal(x['pose3d'], y['pose3d']) and np.array_equal(x['velConstraints'], y['velConstraints']) and np.array_equal(x['plane_normal'], y['plane_normal'])


def test_a_clip_without_contacts_loses_only_itself(data_root):
    """Two contact labels cannot pin a plane: scikit-learn raises inside the reference; here the clip is marked, the other one is solved."""
    root, clips = data_root
    d = str(root / 'walk_a')
    fc = np.zeros_like(np.load(os.path.join(d, 'foot_contacts.npy'))); fc[0, 1] = 1
    np.save(os.path.join(d, 'foot_contacts.npy'), fc)
    opt = kopt.KinematicOptimizer(ik=EmuIk(), kin=EmuKin(max_nfev=4, lsmr_maxiter=5))
    dirs = [d, str(root / 'walk_b')]
    outs = [os.path.join(x, 'kinematic_results') for x in dirs]
    res = drv.optimize_videos(dirs, outs, str(root / 'skel.bvh'), 0, [9, 12], optimizer=opt)
    assert 'contact labels' in res[0]['error'] and res[1]['error'] is None
    assert not os.path.exists(outs[0]) and os.path.exists(os.path.join(outs[1], 'floor_out.txt'))
    with pytest.raises(ValueError):
        kopt.save_results(outs[0], res[0], None)


def test_command_line_flags_of_the_reference():
    """scripts/run_phys_mocap.py:103-115 passes --input_path --skel_path --output_path --end --character [--gt-floor] [--visualize]."""
    with pytest.raises(SystemExit):
        drv.main([])
    with pytest.raises(FileNotFoundError):
        drv.main(['--input_path', '/nonexistent/video/video.mp4', '--skel_path', os.path.join(HERE, 'nonexistent.bvh'), '--output_path', '/tmp/x', '--end', '30',
                  '--character', 'ybot', '--gt-floor', '--visualize'])


# ---- shard determinism of the batch driver: `--data` over five video directories as one process and as two ranks (RANK / WORLD_SIZE, as
#      torch.distributed.run sets them; the path has no collective).  This container has no GPU: the two library-backed solvers are replaced
#      by the host emulation of the same kernel sources; directory handling, ingest, sharding and file output are the driver's own code.
def _shard_worker(rank, world, root, skel):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    sys.path.insert(0, HERE)
    import chd_amd  # noqa: F401
    from chd_amd import kinematic_optimizer as k
    from chd_amd import run_kinematic_optimizer as d
    from test_kinopt_emu import EmuIk, EmuKin
    k.IkBackProject = lambda device, cfg: EmuIk()
    k.KinSolver = lambda device, parents=None: EmuKin(max_nfev=5, lsmr_maxiter=8)
    sys.exit(d.main(['--data', root, '--skel_path', skel]))


def test_two_ranks_write_the_same_files_as_one(tmp_path):
    import multiprocessing as mp
    import shutil
    rng = np.random.default_rng(9)
    one = tmp_path / 'one'
    for i, F in enumerate((8, 11, 9, 12, 10)):
        write_video_dir(str(one / ('clip_%d' % i)), make_kin_clip(20 + i, F, G['c0_skel_offsets'], G['c0_skel_parents']), rng)
    write_skeleton(str(tmp_path / 'skel.bvh'))
    two = tmp_path / 'two'
    shutil.copytree(str(one), str(two))
    ctx = mp.get_context('spawn')
    for root, world in ((one, 1), (two, 2)):
        procs = [ctx.Process(target=_shard_worker, args=(r, world, str(root), str(tmp_path / 'skel.bvh'))) for r in range(world)]
        for p in procs:
            p.start()
        for p in procs:
            p.join(timeout=300)
            assert p.exitcode == 0
    for i in range(5):
        for name in ('foot_contacts.npy', 'floor_out.txt', 'final_test.bvh'):
            a = open(str(one / ('clip_%d' % i) / 'kinematic_results' / name), 'rb').read()
            b = open(str(two / ('clip_%d' % i) / 'kinematic_results' / name), 'rb').read()
            assert a == b and len(a) > 0, (i, name)

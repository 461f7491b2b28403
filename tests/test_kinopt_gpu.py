"""GPU parity of the kinematic optimisation through its C ABIs: libchd_kinopt.so (the two least-squares
solves) and libchd_ik.so (the IK initialisation) on an MI355X -- against the oracle, the host emulation of the same kernel source
and the vectors the reference's own functions produced (tests/golden/kinopt_golden.npz).  Tolerances: see tests/test_kinopt_emu.py."""
import os
import sys

import numpy as np
import pytest

import chd_amd  # noqa: F401
from chd_amd import kinematic_optimizer as kopt

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, 'host_emu'))
sys.path.insert(0, HERE)
from test_kinopt_emu import clip_of, plane_gap_at_feet, problem, rel      # noqa: E402
GOLD = os.path.join(HERE, 'golden', 'kinopt_golden.npz')


@pytest.fixture(scope='module')
def gold():
    torch = pytest.importorskip('torch')
    if not torch.cuda.is_available():
        pytest.skip('needs an MI355X')
    return np.load(GOLD)


def test_bounded_solve_in_lockstep_with_oracle_and_emulation(gold):
    from oracle import kinopt_oracle as ko
    import kin_emu
    for ci, li, cut in [(0, 0, 3), (2, 1, 8)]:
        p, q = problem(gold, ci, li)
        k = 'c%d_' % ci
        P = ko.Problem(p['offsets'], gold[k + 'skel_parents'], p['pose3d'], p['root_trans'], p['pose2d_n'], p['proj_w'], p['data_w'], p['contact'], p['floor_n'], p['floor_p'],
                       kopt.STAGE_WEIGHTS[li])
        x, cost, nfev, njev, status = ko.trf_lsmr(P.fun, P.jac, p['x0'], lsmr_maxiter=cut)
        e = kin_emu.solve([p], kin_emu.default_config(lsmr_maxiter=cut))[0]
        r = kopt.KinSolver(device=0, lsmr_maxiter=cut).solve([p])[0]
        assert (r['nfev'], r['njev'], r['status']) == (nfev, njev, status) == (e['nfev'], e['njev'], e['status'])
        assert rel(r['x'], x) < 1e-7 and rel(r['x'], e['x']) < 1e-7 and abs(r['cost'] - cost) < 1e-6 * cost


def test_slices_in_device_memory_on_gpu(gold):
    """1 100 doubles of LDS hold no frame: the clip is cut into two-frame slices (8 workgroups for 16 frames) whose arrays stay in device memory -- the
    same code through generic pointers -- against the default (one workgroup, everything in LDS) and the emulation."""
    import kin_emu
    p, q = problem(gold, 1, 1)
    small = kopt.KinSolver(device=0, lsmr_maxiter=3); small.cfg.reserved[1] = 1100
    a = kopt.KinSolver(device=0, lsmr_maxiter=3).solve([p])[0]
    b = small.solve([p, p])[1]
    ecfg = kin_emu.default_config(lsmr_maxiter=3); ecfg.reserved[1] = 1100
    e = kin_emu.solve([p], ecfg)[0]
    assert (a['nfev'], a['status']) == (b['nfev'], b['status']) == (e['nfev'], e['status'])
    assert rel(b['x'], a['x']) < 1e-8 and rel(b['x'], e['x']) < 1e-8


@pytest.mark.parametrize('frames_cap', [2, 3, 5])
def test_clusters_of_workgroups_on_gpu(gold, frames_cap):
    """A clip split over G workgroups that exchange halos and partial sums through device memory (round 5), forced on the fixture's 12- and 16-frame clips by
    `reserved[2]` (frames per workgroup): G = 6 / 8, 4 / 6, 3 / 4.  Sixty clips of mixed length in one call: two launches (one per cluster size), more clips
    than resident clusters for the larger size (the queue), every copy of a clip bit for bit the same, and each equal to the emulation of the same split
    to rounding (bounded solves: the comparison is of the arithmetic, not of where a sensitive solve ends)."""
    import kin_emu
    ps = [problem(gold, ci, li)[0] for ci, li in [(1, 1), (2, 0), (0, 0)]]
    solver = kopt.KinSolver(device=0, lsmr_maxiter=4); solver.cfg.reserved[2] = frames_cap
    res = solver.solve(ps * 20)
    ecfg = kin_emu.default_config(lsmr_maxiter=4); ecfg.reserved[2] = frames_cap
    emu = kin_emu.solve(ps, ecfg)
    for k, r in enumerate(res):
        e, first = emu[k % 3], res[k % 3]
        assert np.array_equal(r['x'], first['x']) and r['cost'] == first['cost']
        assert (r['nfev'], r['njev'], r['status']) == (e['nfev'], e['njev'], e['status'])
        assert rel(r['x'], e['x']) < 1e-8 and abs(r['cost'] - e['cost']) < 1e-7 * e['cost']      # (device and host sin / cos differ in the last bit)


def test_a_clip_too_long_for_sixteen_slices_in_lds(gold):
    """250 frames: sixteen workgroups with slices of 15-16 frames, more than the LDS block holds (13) -- the slices' arrays stay in device memory and the same templates run
    through generic pointers (items in two rounds per phase: 16 x 32 lanes against 512 threads).  A bounded solve against the emulation of the same split."""
    import kin_emu
    from chd_amd.synth import make_kin_clip
    F = 250
    cl = make_kin_clip(11, F, gold['c0_skel_offsets'], gold['c0_skel_parents'])
    p2n, pw, dw = kopt.prepare_weights(cl['poses2D'], cl['joint_conf_2d'], (cl['ppx'], cl['ppy']), cl['camFocal'])
    rng = np.random.default_rng(5)
    p = dict(offsets=cl['offsets'], pose3d=cl['poses3D'], root_trans=cl['root_pos'], pose2d_n=p2n, proj_w=pw, data_w=dw, contact=np.array(cl['velConstraints']),
             floor_n=np.array([0.0, 1.0, 0.0]), floor_p=np.zeros(3), weights=kopt.STAGE_WEIGHTS[1],
             x0=np.concatenate([cl['root_pos'], rng.normal(size=(F, 84)) * 0.2], axis=1).reshape(-1))
    cfg = kin_emu.default_config(lsmr_maxiter=3, max_nfev=4)
    assert kin_emu.cluster_size(cfg, F) == 16
    e = kin_emu.solve([p], cfg)[0]
    solver = kopt.KinSolver(device=0, lsmr_maxiter=3); solver.cfg.max_nfev = 4
    r = solver.solve([p, p])
    assert np.array_equal(r[0]['x'], r[1]['x'])
    assert (r[0]['nfev'], r[0]['njev'], r[0]['status'], r[0]['lsmr_iterations']) == (e['nfev'], e['njev'], e['status'], e['lsmr_iterations'])
    assert rel(r[0]['x'], e['x']) < 1e-8 and abs(r[0]['cost'] - e['cost']) < 1e-7 * e['cost']


def test_clusters_under_concurrent_calls_and_uploads(gold):
    """Two host threads call the library at once, as `KinematicOptimizer.optimize` does: their launches take turns, but one thread's uploads run beside the other's
    launch.  That is how round 5 found a flag overtaking its payload (a workgroup-scope release fence emits no `s_waitcnt` for device memory here: the halo
    stores of other wavefronts were still in flight when the flag went out; the ranks of a cluster then disagreed and waited for each other).  Every result of
    eight interleaved calls must be bit for bit the single-threaded one."""
    from concurrent.futures import ThreadPoolExecutor
    ps = [problem(gold, ci, li)[0] for ci, li in [(1, 1), (2, 0), (0, 0)]] * 12
    def make():
        k = kopt.KinSolver(device=0, lsmr_maxiter=40); k.cfg.reserved[2] = 3
        return k
    ref = make().solve(ps)
    solvers = [make(), make()]
    with ThreadPoolExecutor(max_workers=2) as ex:
        outs = list(ex.map(lambda i: solvers[i % 2].solve(ps), range(8)))
    for out in outs:
        for r, a in zip(out, ref):
            assert np.array_equal(r['x'], a['x']) and r['cost'] == a['cost'] and r['lsmr_iterations'] == a['lsmr_iterations']


def test_a_missing_workgroup_is_an_error_not_a_hang(gold):
    """The members of a cluster wait on each other: a launch that is not fully resident would spin for ever.  The wait is bounded (5 s; 0.25 s under the
    test hook `reserved[3] = 0x7e57`, which makes the last workgroup of the first cluster return at once): the launch winds down, the call reports it, and
    the next call on the same device works."""
    p = problem(gold, 1, 1)[0]
    bad = kopt.KinSolver(device=0, lsmr_maxiter=3); bad.cfg.reserved[2] = 4; bad.cfg.reserved[3] = 0x7e57; bad.cfg.reserved[0] = 1      # (no retry: the error itself)
    with pytest.raises(RuntimeError, match='waited too long'):
        bad.solve([p, p, p])
    # default behaviour (round 6, advisor): the batch is solved once more with one workgroup per clip instead of failing -- same results to rounding as a
    # call that asked for one workgroup per clip in the first place
    retry = kopt.KinSolver(device=0, lsmr_maxiter=3); retry.cfg.reserved[2] = 4; retry.cfg.reserved[3] = 0x7e57
    rr = retry.solve([p, p, p])
    assert retry.last_call_retried()
    one = kopt.KinSolver(device=0, lsmr_maxiter=3); one.cfg.reserved[2] = int(np.asarray(p['pose3d']).shape[0])      # one workgroup per clip, asked for
    r1 = one.solve([p])[0]
    assert not one.last_call_retried()
    for r in rr:
        assert np.array_equal(r['x'], r1['x']) and r['cost'] == r1['cost']
    good = kopt.KinSolver(device=0, lsmr_maxiter=3); good.cfg.reserved[2] = 4
    r = good.solve([p])[0]
    assert r['nfev'] >= 1 and np.isfinite(r['cost'])


def test_every_solve_of_the_fixture_matches_the_reference(gold):
    ps = [problem(gold, ci, li) for ci in range(3) for li in range(2)]
    solver = kopt.KinSolver(device=0)
    res = solver.solve([p for p, _ in ps] * 3)                 # 18 workgroups; the three copies must agree bit for bit
    for i, ((p, q), r) in enumerate(zip(ps, res)):
        assert rel(r['x'], gold[q + 'x']) < 5e-4
        assert abs(r['cost'] - float(gold[q + 'cost'])) < 1e-2 * float(gold[q + 'cost'])       # (the cost is steep: projection residuals carry a weight of 1000)
        assert r['status'] == int(gold[q + 'status']) and abs(r['nfev'] - int(gold[q + 'nfev'])) <= 1
        print('%s  x rel %.2e  cost %.3f (reference %.3f)  nfev %d (%d)  LSMR iterations %d' % (q, rel(r['x'], gold[q + 'x']), r['cost'], float(gold[q + 'cost']), r['nfev'], int(gold[q + 'nfev']), r['lsmr_iterations']))
        for rep in (1, 2):
            assert np.array_equal(res[i + rep * len(ps)]['x'], r['x'])
    assert solver.last_kernel_ms() > 0


def test_whole_optimisation_on_gpu(gold, tmp_path):
    g = gold
    res = kopt.KinematicOptimizer(device=0).optimize([clip_of(g, ci) for ci in range(3)])
    for ci, r in enumerate(res):
        k = 'c%d_' % ci
        s = np.sign((r['ik_rot'] * g[k + 'ik_rot']).sum(-1, keepdims=True))
        assert rel(r['ik_rot'] * s, g[k + 'ik_rot']) < 1e-10
        assert np.array_equal(r['velConstraints'], g[k + 'out_vel'])
        assert plane_gap_at_feet(g, k, r) < 0.1 and np.abs(r['plane_normal'] - g[k + 'out_floor_n']).max() < 2e-2      # (see tests/test_kinopt_emu.py)
        assert rel(r['pose3d'], g[k + 'out_pose3d']) < 2e-3 and rel(r['proj2d'], g[k + 'out_proj2d']) < 2e-3
        print('clip %d  floor gap at the feet %.1e cm  normal %.1e  pose3d %.1e  proj2d %.1e' % (ci, plane_gap_at_feet(g, k, r), np.abs(r['plane_normal'] - g[k + 'out_floor_n']).max(), rel(r['pose3d'], g[k + 'out_pose3d']), rel(r['proj2d'], g[k + 'out_proj2d'])))
        kopt.save_results(str(tmp_path / ('clip%d' % ci)), r, ['j%d' % j for j in range(28)])
        assert np.array_equal(np.load(str(tmp_path / ('clip%d' % ci) / 'foot_contacts.npy')), kopt.refined_contacts(g[k + 'out_vel']))


def test_batch_driver_on_gpu(gold, tmp_path):
    """`python -m chd_amd.run_kinematic_optimizer --data <root>`: OpenPose JSON + tracked_results.json + foot_contacts.npy of every video
    directory in, kinematic_results/{foot_contacts.npy, floor_out.txt, final_test.bvh} out, all videos in one batched solve."""
    from chd_amd import run_kinematic_optimizer as drv
    from chd_amd import skeleton_io as sio
    from chd_amd.synth import make_kin_clip
    from test_kinopt_driver import write_skeleton, write_video_dir
    rng = np.random.default_rng(4)
    frames = {'walk_a': 9, 'walk_b': 12, 'walk_c': 30}
    for i, (name, F) in enumerate(frames.items()):
        write_video_dir(str(tmp_path / name), make_kin_clip(i, F, gold['c0_skel_offsets'], gold['c0_skel_parents']), rng)
    write_skeleton(str(tmp_path / 'skel.bvh'))
    assert drv.main(['--data', str(tmp_path), '--skel_path', str(tmp_path / 'skel.bvh')]) == 0
    for name, F in frames.items():
        out = tmp_path / name / 'kinematic_results'
        fc = np.load(str(out / 'foot_contacts.npy'))
        assert fc.shape == (F, 4) and set(np.unique(fc)) <= {0, 1}
        n = np.array([float(v) for v in open(str(out / 'floor_out.txt')).readline().split(' ')])
        assert abs(np.linalg.norm(n) - 1) < 1e-12 and n[1] < -0.9              # y is down in the camera frame: the floor's normal points up
        m, _, _ = sio.load_bvh(str(out / 'final_test.bvh'))
        assert m.n_frames == F and m.n_joints == 28


@pytest.mark.parametrize('fixture,lds_doubles', [('kinopt_golden_long.npz', 0), ('kinopt_golden_long.npz', 9216), ('kinopt_golden_100.npz', 0)])
def test_realistic_clip_lengths_match_the_reference(fixture, lds_doubles):
    """Clips of 40 and 60 frames (tests/golden/kinopt_golden_long.npz: the reference's own `optimize_trajectory` run on them,
    make_kinopt_golden.py --long) through the kernel: every least-squares solve within 5e-4 of the reference's solution, the relabelled contacts exact
    (VERDICT r02 item 5).  Twice: at the DEFAULT frame tiles (144 KB of LDS: 71 frames for J v, 54 for J^T u -- the 40-frame clip crosses no tile boundary,
    the 60-frame one only that of J^T u) and at 72 KB tiles (34 / 27 frames: both clips cross both boundaries, as the 100-frame clips of the bench do at the
    default) -- VERDICT r03 "weak" 3.  And one 100-frame clip (kinopt_golden_100.npz, make_kinopt_golden.py --long 100 --out=...: the bench's clip length), which
    crosses both boundaries of the default tiles."""
    path = os.path.join(HERE, 'golden', fixture)
    if not os.path.exists(path):
        pytest.skip(fixture + ' not generated')
    torch = pytest.importorskip('torch')
    if not torch.cuda.is_available():
        pytest.skip('needs an MI355X')
    g = np.load(path)
    n = int(g['n_cases'])
    ps = [problem(g, ci, li) for ci in range(n) for li in range(2)]
    solver = kopt.KinSolver(device=0)
    if lds_doubles:
        solver.cfg.reserved[1] = lds_doubles
    res = solver.solve([p for p, _ in ps])
    for (p, q), r in zip(ps, res):
        err = rel(r['x'], g[q + 'x'])
        print('%s  frames %d  x rel %.2e  cost %.3f (reference %.3f)  nfev %d (%d)  LSMR iterations %d' % (q, p['pose3d'].shape[0], err, r['cost'], float(g[q + 'cost']), r['nfev'],
                                                                                                 int(g[q + 'nfev']), r['lsmr_iterations']))
        assert err < 5e-4, (q, err)
        assert abs(r['cost'] - float(g[q + 'cost'])) < 1e-2 * float(g[q + 'cost'])
    if lds_doubles:
        return
    whole = kopt.KinematicOptimizer(device=0).optimize([clip_of(g, ci) for ci in range(n)])
    for ci, r in enumerate(whole):
        k = 'c%d_' % ci
        assert np.array_equal(r['velConstraints'], g[k + 'out_vel'])
        assert rel(r['pose3d'], g[k + 'out_pose3d']) < 2e-3
        print('clip %d (%d frames): pose3d %.1e vs the reference' % (ci, r['pose3d'].shape[0], rel(r['pose3d'], g[k + 'out_pose3d'])))

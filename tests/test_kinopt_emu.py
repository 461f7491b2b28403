"""The kinematic-optimisation kernel source (csrc/chd_kinopt_kernels.hpp) through its host emulation, and the host pipeline
(chd_amd.kinematic_optimizer) driven by the emulated IK and least-squares solvers -- against the oracle and against the vectors
the REFERENCE's own functions produced (tests/golden/kinopt_golden.npz).  The GPU twin is tests/test_kinopt_gpu.py.

Tolerances.  The reference's quaternion constructor divides a unit axis by (1 + 1e-10) (Quaternions.py:394-399); the kernel uses
exact rotation matrices, so residuals and Jacobian products agree with the reference to ~1e-9, not 1e-15.  LSMR amplifies any
difference -- 1e-10 after 5 iterations, 1e-4 after 25, 1e-2 after 100 (SciPy's own sparse and dense products differ that
much) -- and the reference runs it to its limit of 87 F iterations, where the variants have come back to within ~1e-3.  So
 * with LSMR cut to 3 (8) iterations per trust-region iteration the whole 50-evaluation solve must take the oracle's path (same
   evaluation counts, same termination) and end within 1e-8 of it: the algorithm is the same;
 * with SciPy's settings the solutions must match the reference's to 5e-4 (the oracle under a 1e-10 perturbation of its input,
   or SciPy with dense instead of sparse products, moves by up to 1e-4: tests/test_kinopt_oracle.py).
What it took to get there: the accuracy of the norms.  With one running sum per norm (the emulation's single thread, before it
modelled the workgroup's 512 partial sums + butterfly) the Golub-Kahan vectors lose orthogonality faster, LSMR is 5 % away from
the others' iterate at its iteration limit and the final solutions differ by 1e-3..3e-3 -- the product's sums are the
workgroup's tree sums, and the emulation reproduces them lane for lane."""
import os
import sys

import numpy as np
import pytest

import chd_amd  # noqa: F401
from chd_amd import kinematic_optimizer as kopt
from chd_amd import skeleton_io as sio

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, 'host_emu'))
GOLD = os.path.join(HERE, 'golden', 'kinopt_golden.npz')


def rel(a, b):
    return np.linalg.norm(np.asarray(a) - np.asarray(b)) / max(np.linalg.norm(b), 1e-300)


@pytest.fixture(scope='module')
def gold():
    return np.load(GOLD)


def problem(g, ci, li):
    k = 'c%d_' % ci; q = '%slsq%d_' % (k, li)
    return dict(offsets=g[k + 'fit_offsets'], pose3d=g[k + 'poses3D'], root_trans=g[k + 'root_pos'], pose2d_n=g[q + 'pose2d_n'], proj_w=g[q + 'proj_w'],
                data_w=g[q + 'data_w'], contact=g[q + 'vel'], floor_n=g[q + 'floor_n'], floor_p=g[q + 'floor_p'], weights=kopt.STAGE_WEIGHTS[li], x0=g[q + 'x0']), q


def clip_of(g, ci):
    k = 'c%d_' % ci
    cl = dict(poses2D=g[k + 'poses2D'], joint_conf_2d=g[k + 'conf'], poses3D=g[k + 'poses3D'], root_pos=g[k + 'root_pos'], joint_angles=g[k + 'joint_angles'],
              offsets=g[k + 'skel_offsets'], parents=g[k + 'skel_parents'], ppx=g[k + 'pp'][0], ppy=g[k + 'pp'][1], camFocal=g[k + 'focal'], velConstraints=g[k + 'vel'])
    if int(g[k + 'given_floor']):
        cl['plane_normal'] = g[k + 'floor_in_n']; cl['plane_point'] = g[k + 'floor_in_p']
    return cl


def plane_gap_at_feet(g, k, r):
    """|height difference| of the fitted and the reference floor at the centroid of the reference's contact feet."""
    feet = g[k + 'out_pose3d'][:, 19:25][g[k + 'out_vel'][:, 19:25] == 1]
    c = feet.mean(axis=0)
    ya = r['plane_point'][1] - (r['plane_normal'][0] * (c[0] - r['plane_point'][0]) + r['plane_normal'][2] * (c[2] - r['plane_point'][2])) / r['plane_normal'][1]
    n, p = g[k + 'out_floor_n'], g[k + 'out_floor_p']
    yb = p[1] - (n[0] * (c[0] - p[0]) + n[2] * (c[2] - p[2])) / n[1]
    return abs(ya - yb)


class EmuIk:
    def solve(self, seqs):
        import ik_emu
        from chd_amd.ik_capi import ChdIkConfig
        return ik_emu.solve(seqs, ChdIkConfig.default(iterations=200, translate=0, damping=7.0, smoothness=0.0))


class EmuKin:
    def __init__(self, **kw):
        self.kw = kw

    def solve(self, problems):
        import kin_emu
        return kin_emu.solve(problems, kin_emu.default_config(**self.kw))


@pytest.mark.parametrize('ci,li', [(0, 0), (0, 1), (1, 1), (2, 1)])
def test_residual_and_jacobian_products_match_the_reference(gold, ci, li):
    import kin_emu
    p, q = problem(gold, ci, li)
    assert rel(kin_emu.probe(p, 0)[0], gold[q + 'f0']) < 5e-9
    assert rel(kin_emu.probe(p, 1, gold[q + 'v'])[0], gold[q + 'Jv']) < 5e-9
    assert rel(kin_emu.probe(p, 2, gold[q + 'u'])[0], gold[q + 'JTu']) < 5e-9


def test_jacobian_products_are_adjoint(gold):
    import kin_emu
    p, q = problem(gold, 1, 1)
    rng = np.random.default_rng(5)
    F = p['pose3d'].shape[0]
    v = rng.normal(size=87 * F); u = rng.normal(size=507 * F - 423)
    a = kin_emu.probe(p, 1, v)[0].dot(u); b = kin_emu.probe(p, 2, u)[0].dot(v)
    assert abs(a - b) < 1e-12 * max(abs(a), abs(b))


def test_bounded_lsmr_solve_matches_the_oracle(gold):
    """Same algorithm: with LSMR cut to a few iterations per trust-region iteration (before its error amplification sets in) the
    whole solve -- regularisation, subspace, 2-D trust-region problem, radius updates, rejected steps, termination -- follows the
    oracle's path step for step."""
    from oracle import kinopt_oracle as ko
    import kin_emu
    for ci, li, cut, tol in [(0, 0, 3, 1e-8), (2, 1, 3, 1e-7), (2, 1, 8, 1e-8)]:      # the last one ends on xtol after 21 evaluations
        p, q = problem(gold, ci, li)
        k = 'c%d_' % ci
        P = ko.Problem(p['offsets'], gold[k + 'skel_parents'], p['pose3d'], p['root_trans'], p['pose2d_n'], p['proj_w'], p['data_w'], p['contact'], p['floor_n'], p['floor_p'],
                       kopt.STAGE_WEIGHTS[li])
        x, cost, nfev, njev, status = ko.trf_lsmr(P.fun, P.jac, p['x0'], lsmr_maxiter=cut)
        r = kin_emu.solve([p], kin_emu.default_config(lsmr_maxiter=cut))[0]
        assert (r['nfev'], r['njev'], r['status']) == (nfev, njev, status)
        assert rel(r['x'], x) < tol and rel(r['x'] - p['x0'], x - p['x0']) < 1e-5 and abs(r['cost'] - cost) < 1e-6 * cost


def test_the_reference_solve_is_this_sensitive(gold):
    """The oracle (which reproduces the reference's arithmetic to 1e-15 at the start point) on an input perturbed by 1e-10:
    the solution moves by 1e-5..1e-4 -- the floor under any implementation's distance to the reference."""
    from oracle import kinopt_oracle as ko
    p, q = problem(gold, 0, 0)
    P = ko.Problem(p['offsets'], gold['c0_skel_parents'], p['pose3d'], p['root_trans'], p['pose2d_n'], p['proj_w'], p['data_w'], p['contact'], p['floor_n'], p['floor_p'],
                   kopt.STAGE_WEIGHTS[0])
    rng = np.random.default_rng(0)
    xa = ko.trf_lsmr(P.fun, P.jac, p['x0'])[0]
    xb = ko.trf_lsmr(P.fun, P.jac, p['x0'] * (1 + 1e-10 * rng.normal(size=p['x0'].size)))[0]
    assert 2e-6 < rel(xb, xa) < 5e-4


def test_every_solve_of_the_fixture_matches_the_reference(gold):
    import kin_emu
    ps = [problem(gold, ci, li) for ci in range(3) for li in range(2)]
    for (p, q), r in zip(ps, kin_emu.solve([p for p, _ in ps])):
        assert rel(r['x'], gold[q + 'x']) < 5e-4
        assert abs(r['cost'] - float(gold[q + 'cost'])) < 5e-3 * float(gold[q + 'cost'])
        assert r['status'] == int(gold[q + 'status']) and abs(r['nfev'] - int(gold[q + 'nfev'])) <= 1


def test_whole_optimisation_and_its_files(gold, tmp_path):
    g = gold
    opt = kopt.KinematicOptimizer(ik=EmuIk(), kin=EmuKin())
    res = opt.optimize([clip_of(g, ci) for ci in range(3)])
    for ci, r in enumerate(res):
        k = 'c%d_' % ci
        s = np.sign((r['ik_rot'] * g[k + 'ik_rot']).sum(-1, keepdims=True))
        assert rel(r['ik_rot'] * s, g[k + 'ik_rot']) < 1e-10                     # IK initialisation
        assert np.array_equal(r['velConstraints'], g[k + 'out_vel'])             # relabelled contacts: exact
        # the floor: compared where it matters, under the feet (two nearly static feet pin the plane's height there, not its tilt:
        # 1 mm of foot position turns the normal by 0.5 degrees and moves the plane under the camera, 3 m away, by centimetres)
        assert plane_gap_at_feet(g, k, r) < 0.1                                  # centimetres
        assert np.abs(r['plane_normal'] - g[k + 'out_floor_n']).max() < 2e-2
        assert rel(r['pose3d'], g[k + 'out_pose3d']) < 2e-3 and rel(r['proj2d'], g[k + 'out_proj2d']) < 2e-3       # two chained solves
        assert rel(r['motion'].positions, g[k + 'out_pos']) < 2e-3
        out = str(tmp_path / ('clip%d' % ci))
        kopt.save_results(out, r, ['j%d' % j for j in range(28)])
        fc = np.load(os.path.join(out, 'foot_contacts.npy'))
        assert fc.shape == (r['pose3d'].shape[0], 4) and set(np.unique(fc)) <= {0, 1}
        ref_fc = kopt.refined_contacts(g[k + 'out_vel'])
        assert np.array_equal(fc, ref_fc)
        lines = open(os.path.join(out, 'floor_out.txt')).read().split('\n')      # parsed as towr_utils / kinematic_optimizer do
        assert np.allclose([float(v) for v in lines[0].split(' ')], r['plane_normal']) and np.allclose([float(v) for v in lines[1].split(' ')], r['plane_point'])
        m, names, _ = sio.load_bvh(os.path.join(out, 'final_test.bvh'))
        assert m.n_frames == r['pose3d'].shape[0] and m.n_joints == 28
        assert rel(sio.positions_global(m), sio.positions_global(r['motion'])) < 1e-5       # '%f' precision of the file
    # the relabelling fired on the two clips without a given floor
    assert int(np.abs(res[0]['velConstraints'] - g['c0_vel']).sum()) == 1 and int(np.abs(res[1]['velConstraints'] - g['c1_vel']).sum()) == 1


@pytest.mark.parametrize('frames_cap,lds_doubles', [(2, 0), (3, 0), (5, 0), (7, 0), (5, 700), (0, 9000)])
def test_a_clip_split_over_a_cluster_of_workgroups(gold, frames_cap, lds_doubles):
    """A clip is solved by G workgroups that each own a run of frames, its LSMR state in the owner's LDS (round 5).  The fixture's clips (12 and 16
    frames) are one and two workgroups by default; `reserved[2]` (frames per workgroup) forces 2 .. 8, with slices that do not divide the clip evenly, and
    `reserved[1]` (LDS doubles) decides what stays in LDS: 700 = the received halos only, every slice in device memory; 9 000 = slices of 5 frames, all of
    LSMR's vectors in LDS.  The building blocks against the reference's vectors whatever the split; bounded solves against the one-workgroup solve (the
    partial sums of a norm are added in a different order: rounding level)."""
    import kin_emu
    for ci, li in [(1, 1), (2, 0)]:
        p, q = problem(gold, ci, li)
        cfg = kin_emu.default_config()
        cfg.reserved[1] = lds_doubles; cfg.reserved[2] = frames_cap
        F = p['pose3d'].shape[0]
        G = kin_emu.cluster_size(cfg, F)
        assert G == (min(8, F // 2, -(-F // frames_cap)) if frames_cap else -(-F // 5)) and G >= 2
        assert rel(kin_emu.probe(p, 0, cfg=cfg)[0], gold[q + 'f0']) < 5e-9
        assert rel(kin_emu.probe(p, 1, gold[q + 'v'], cfg=cfg)[0], gold[q + 'Jv']) < 5e-9
        assert rel(kin_emu.probe(p, 2, gold[q + 'u'], cfg=cfg)[0], gold[q + 'JTu']) < 5e-9
        one = kin_emu.default_config(lsmr_maxiter=3); one.reserved[2] = 1000
        many = kin_emu.default_config(lsmr_maxiter=3); many.reserved[1] = lds_doubles; many.reserved[2] = frames_cap
        assert kin_emu.cluster_size(one, F) == 1
        a = kin_emu.solve([p], one)[0]; b = kin_emu.solve([p], many)[0]
        assert (a['nfev'], a['status']) == (b['nfev'], b['status']) and rel(b['x'], a['x']) < 1e-9


def test_a_skeleton_that_is_not_in_depth_first_order(gold):
    """The walk over a joint's descendants takes the contiguous range j + 1 .. j + n when the joints are in depth-first order (the reference's skeleton is), with the four
    idle lanes of a frame helping on the longest walks; any other numbering (parents[j] < j is all the ABI asks for) falls back to every later joint as a candidate under the
    descendant mask.  Products and residual against the oracle's Jacobian for such a tree, one workgroup and clusters of four."""
    from oracle import kinopt_oracle as ko
    import kin_emu
    p, q = problem(gold, 1, 1)
    par = np.array([-1, 0, 0, 1, 2, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 17, 18, 19, 20, 21, 22, 23])      # siblings interleaved with each other's children
    P = ko.Problem(p['offsets'], par, p['pose3d'], p['root_trans'], p['pose2d_n'], p['proj_w'], p['data_w'], p['contact'], p['floor_n'], p['floor_p'], kopt.STAGE_WEIGHTS[1])
    J = P.jac(p['x0']); J = J.toarray() if hasattr(J, 'toarray') else np.asarray(J)
    rng = np.random.default_rng(1)
    v, u = rng.normal(size=J.shape[1]), rng.normal(size=J.shape[0])
    for cap in (0, 3):
        cfg = kin_emu.default_config()
        for j in range(28):
            cfg.parents[j] = int(par[j])
        cfg.reserved[2] = cap
        assert rel(kin_emu.probe(p, 0, cfg=cfg)[0], P.fun(p['x0'])) < 5e-9
        assert rel(kin_emu.probe(p, 1, v, cfg=cfg)[0], J @ v) < 5e-9 and rel(kin_emu.probe(p, 2, u, cfg=cfg)[0], J.T @ u) < 5e-9


def test_batched_huber_fits_find_the_regressors_minimum():
    """The floor fits of a batch (`huber_fit_batch`: block descent on all clips' problems at once) against the per-clip solve that mirrors
    HuberRegressor (`huber_fit`: SciPy's L-BFGS-B, gtol 1e-5) and against scikit-learn itself: never a higher objective, the same outlier
    labels, coefficients within the regressor's own stopping error; ragged sizes and an outlier-heavy problem included."""
    from sklearn.linear_model import HuberRegressor
    rng = np.random.default_rng(4)
    Xs, ys = [], []
    for b in range(40):
        n = int(rng.integers(3, 300))
        X = rng.normal(size=(n, 2)) * np.array([30.0, 50.0]) + np.array([10.0, 200.0])
        y = X @ (rng.normal(size=2) * 0.05) + rng.normal() * 20 - 90 + rng.normal(size=n) * rng.uniform(0.2, 3.0)
        k = int(rng.integers(0, max(1, n // 4)))
        y[:k] += rng.normal(size=k) * 40
        Xs.append(X); ys.append(y)
    for eps in (1.5, 2.2):
        new = kopt.huber_fit_batch(Xs, ys, eps)
        for X, y, q in zip(Xs, ys, new):
            r = kopt.huber_fit(X, y, eps)
            f_new = kopt._huber_objective(np.r_[q[0], q[1], q[2]], X, y, eps, 1e-4)[0]
            f_ref = kopt._huber_objective(np.r_[r[0], r[1], r[2]], X, y, eps, 1e-4)[0]
            assert f_new <= f_ref + 1e-9 * abs(f_ref)
            if len(y) >= 20:            # (tiny problems are flat enough for L-BFGS-B to stop visibly early)
                assert np.abs(q[0] - r[0]).max() < 1e-4 and abs(q[1] - r[1]) < 2e-2 and abs(q[2] - r[2]) < 2e-3 * r[2]
                assert np.array_equal(q[3], r[3])
    X, y = Xs[5], ys[5]
    h = HuberRegressor(epsilon=1.5).fit(X, y)
    q = kopt.huber_fit_batch([X], [y], 1.5)[0]
    assert np.abs(q[0] - h.coef_).max() < 1e-4 and abs(q[1] - h.intercept_) < 2e-2 and np.array_equal(q[3], h.outliers_)
    # the two-fit floor of a batch = the per-clip one
    fp = [np.c_[X[:, 0], y_, X[:, 1]] for X, y_ in zip(Xs[:6], ys[:6])] + [None]
    for a, b in zip(kopt.fit_floor_batch(fp), [kopt.fit_floor(f) if f is not None else None for f in fp]):
        if a is None:
            assert b is None
        else:
            assert np.abs(a[0] - b[0]).max() < 1e-5 and np.abs(a[1] - b[1]).max() < 2e-2 and np.array_equal(a[2], b[2])

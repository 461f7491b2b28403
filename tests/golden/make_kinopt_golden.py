"""Generates tests/golden/kinopt_golden.npz: inputs, intermediates and outputs of the REFERENCE's kinematic optimisation
(`optimize_trajectory`, src/optimize/optimize_trajectory.py:522-834 -- SURVEY 8(f) rank 3, the producer of `floor_out.txt`,
the refined `foot_contacts.npy` and `final_test.bvh` that the physics stage consumes) on small synthetic clips, run with the
reference's own functions, SciPy's `least_squares(method='trf', tr_solver='lsmr')` and scikit-learn's `HuberRegressor`.

What is recorded per case: the inputs; the fitted skeleton offsets (`update_skeleton`, :485-520); the IK initialisation
(`JacobianInverseKinematicsCK(translate=False, iterations=200, smoothness=0, damping=7)`, :611-617); for each of the two
`least_squares` calls (:660, :779) the start point, the residual vector and J v / J^T u products of the reference's residual /
Jacobian functions at the start point (:237-483; random v, u), and the solution; the fitted floor and relabelled contacts
(:713-767); the final outputs (:791-834).

The reference library needs three stand-ins under NumPy 2 / this image: `numpy.core.umath_tests` (two functions), `np.float`
(removed alias) and empty modules for `cv2` / `openpose_utils` / `totalcap_utils` (imported by optimize_trajectory.py:29-30 but
not used on this path).  Run in the build container only (it reads /root/reference); tests use the committed fixture.

    python tests/golden/make_kinopt_golden.py                      # the three short clips (10 / 16 / 12 frames)
    python tests/golden/make_kinopt_golden.py --long 40 60         # kinopt_golden_long.npz: clips of 40 and 60 frames (tens of minutes: the reference's dense Jacobians)
    python tests/golden/make_kinopt_golden.py --long 100 --out=kinopt_golden_100.npz      # one clip of the bench's length (crosses the kernel's default frame tiles)
"""
import contextlib
import io
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = '/root/reference/src'


def load_reference():
    shim = types.ModuleType('numpy.core.umath_tests')
    shim.matrix_multiply = np.matmul
    shim.inner1d = lambda a, b: np.einsum('...i,...i->...', a, b)
    sys.modules['numpy.core.umath_tests'] = shim
    np.float = float
    for name in ('cv2', 'openpose_utils', 'totalcap_utils'):
        sys.modules[name] = types.ModuleType(name)
    sys.path.insert(0, os.path.join(REF, 'skeleton_fitting', 'ik'))
    sys.path.insert(0, os.path.join(REF, 'optimize'))
    import optimize_trajectory as ot
    import BVH
    import Animation
    from Quaternions import Quaternions
    return ot, BVH, Animation, Quaternions


def synth_case(ot, BVH, Animation, Quaternions, seed, F, noise3d=1.5, noise2d=3.0):
    """A walking-like clip on the combined 28-joint skeleton: smooth random joint angles, a root drifting in front of the
    camera (y down, z forward, centimetres -- the monocular-total-capture frame the reference works in), noisy 3D joints,
    noisy 2D projections with confidences, alternating foot contacts."""
    rng = np.random.default_rng(seed)
    skel, names, _ = BVH.load(os.path.join(REF, 'skeleton_fitting', 'combined_body_25.bvh'))
    nj = skel.shape[1]
    t = np.arange(F)[:, None, None] / 30.0
    amp = rng.uniform(0.05, 0.35, size=(1, nj, 3)); ph = rng.uniform(0, 2 * np.pi, size=(1, nj, 3)); fr = rng.uniform(0.5, 2.0, size=(1, nj, 3))
    amp[:, 1:13] = rng.uniform(0.01, 0.04, size=(1, 12, 3))               # quiet legs: planted feet stay (nearly) planted ...
    eul = amp * np.sin(2 * np.pi * fr * t + ph)
    eul[:, 0] += np.array([0.1, 0.4, 0.05])
    half = F // 2
    swing = np.sin(np.pi * np.clip((np.arange(F) - half) / max(F - half - 1, 1), 0, 1)) ** 2
    eul[:, 2, 0] += 0.9 * swing                                            # ... and the swing leg bends its knee (left: second half,
    eul[:, 8, 0] += 0.9 * swing[::-1]                                      #     right: first half): a label there is an outlier of the floor fit
    rot = Quaternions.from_euler(eul, order='xyz', world=True)
    scale = rng.uniform(0.9, 1.15)
    offsets = skel.offsets * scale
    pos = np.repeat(offsets[None], F, axis=0)
    root = np.array([20.0, 40.0, 320.0]) + np.arange(F)[:, None] * np.array([1.5, 0.05, -0.8]) + rng.normal(size=(F, 3)) * 0.3
    pos[:, 0] = root
    anim = Animation.Animation(rot, pos, skel.orients, offsets, skel.parents)
    gp = Animation.positions_global(anim)                                  # F x 28 x 3, skeleton order
    fwd, bwd = ot.FORWARD_MAPPING, ot.BACKWARD_MAPPING
    p3 = np.zeros_like(gp)
    for j in range(nj):
        p3[:, j] = gp[:, bwd[j]] - root                                    # data order (body-25 + 3 spine), root relative
    p3 += rng.normal(size=p3.shape) * noise3d
    p3[:, ot.ROOT_IDX] = 0.0
    root_in = root + rng.normal(size=root.shape) * 1.0
    focal = np.array([2000.0, 2000.0]); ppx, ppy = 960.0, 540.0
    gabs = np.zeros_like(gp)
    for j in range(nj):
        gabs[:, j] = gp[:, bwd[j]]
    p2 = np.stack([focal[0] * gabs[:, :, 0] / gabs[:, :, 2] + ppx, focal[1] * gabs[:, :, 1] / gabs[:, :, 2] + ppy], axis=2)
    p2 += rng.normal(size=p2.shape) * noise2d
    conf = rng.uniform(0.3, 1.0, size=(F, nj))
    conf[rng.uniform(size=(F, nj)) < 0.05] = 0.0                           # missed detections
    p2[:, 25:] = 0.0; conf[:, 25:] = 0.0                                   # kinematic_optimizer.py:93-96 padding of the spine joints
    # joint angles in the convention of :589-594 (angle-axis with the axis negated)
    q = rot.qs
    ang = 2.0 * np.arccos(np.clip(q[..., 0], -1, 1))
    ax = q[..., 1:] / np.maximum(np.linalg.norm(q[..., 1:], axis=-1, keepdims=True), 1e-12)
    jangles = -(ax * ang[..., None]) + rng.normal(size=(F, nj, 3)) * 0.03
    # contacts (kinematic_optimizer.py:111-117 columns): left foot planted in the first half, right foot in the second
    vel = np.zeros((F, nj))
    vel[:half + 1, 19] = 1; vel[:half + 1, 20] = 1; vel[:half, 21] = 1
    vel[half:, 22] = 1; vel[half:, 23] = 1; vel[half + 1:, 24] = 1
    vel[half + (F - half) // 2, 21] = 1                                    # one spurious label: the left heel in mid swing
    return dict(poses2D=p2, conf=conf, poses3D=p3, root_pos=root_in, joint_angles=jangles, vel=vel, focal=focal, pp=np.array([ppx, ppy]),
                skel_offsets=skel.offsets.copy(), skel_parents=skel.parents.copy(), names=names), skel


if __name__ == '__main__':
    ot, BVH, Animation, Quaternions = load_reference()
    import scipy
    import sklearn
    cases = [dict(seed=1, F=10, floor=None), dict(seed=2, F=16, floor=None),
             dict(seed=3, F=12, floor='refit')]
    out_name = 'kinopt_golden.npz'
    if len(sys.argv) > 1 and sys.argv[1] == '--long':          # clips of a realistic length: kinopt_golden_long.npz, frames from the command line
        frames_ = [a for a in sys.argv[2:] if not a.startswith('--out=')] or ['40', '60']
        cases = [dict(seed=10 + i + (0 if len(frames_) > 1 else int(frames_[0])), F=int(f), floor=None) for i, f in enumerate(frames_)]
        out_name = ([a[6:] for a in sys.argv[2:] if a.startswith('--out=')] or ['kinopt_golden_long.npz'])[0]
    out = {'n_cases': np.array(len(cases)), 'scipy_version': np.array(scipy.__version__), 'sklearn_version': np.array(sklearn.__version__),
           'forward_mapping': np.array([ot.FORWARD_MAPPING[j] for j in range(28)]), 'backward_mapping': np.array([ot.BACKWARD_MAPPING[j] for j in range(28)])}
    real_lsq = ot.least_squares
    real_update = ot.update_skeleton
    real_ik = ot.JacobianInverseKinematicsCK
    def run_case(ci, data, skel, fl):
        rec = {'lsq': []}
        rng = np.random.default_rng(100 + ci)

        def lsq(fun, x0, **kw):
            args = kw['args']
            f0 = fun(x0, *args)
            J0 = kw['jac'](x0, *args).tocsr()
            v = rng.normal(size=x0.size); u = rng.normal(size=f0.size)
            sol = real_lsq(fun, x0, **kw)
            rec['lsq'].append(dict(x0=x0.copy(), f0=f0, v=v, u=u, Jv=J0 @ v, JTu=J0.T @ u, x=sol.x.copy(), cost=sol.cost, nfev=sol.nfev, njev=sol.njev,
                                   status=sol.status, optimality=sol.optimality, proj_w=args[6].copy(), data_w=args[7].copy(), pose2d_n=args[3].copy(),
                                   floor_n=np.asarray(args[4], dtype=float).copy(), floor_p=np.asarray(args[5], dtype=float).copy(), vel=np.asarray(args[11]).copy()))
            return sol

        def upd(skel_ref, targets, names=None):
            s = real_update(skel_ref, targets, names)
            rec['fit_offsets'] = s.offsets.copy()
            return s

        class IK(real_ik):
            def __call__(self, *a, **kw):
                rec['ik_rot0'] = self.animation.rotations.qs.copy()
                r = real_ik.__call__(self, *a, **kw)
                rec['ik_rot'] = self.animation.rotations.qs.copy(); rec['ik_pos'] = self.animation.positions.copy()
                return r

        ot.least_squares = lsq; ot.update_skeleton = upd; ot.JacobianInverseKinematicsCK = IK
        tmp = os.path.join('/tmp', 'kinopt_case%d' % ci)
        os.makedirs(tmp, exist_ok=True)
        with contextlib.redirect_stdout(io.StringIO()):
            res = ot.optimize_trajectory(
                data['poses2D'].copy(), data['conf'].copy(), data['poses3D'].copy(), data['root_pos'].copy(), data['joint_angles'].copy(), skel, data['names'],
                data['pp'][0], data['pp'][1], data['focal'], data['vel'].copy(), save_dir=tmp,
                plane_normal=None if fl is None else fl[0].copy(), plane_point=None if fl is None else fl[1].copy())
        return rec, res, tmp

    for ci, cs in enumerate(cases):
        data, skel = synth_case(ot, BVH, Animation, Quaternions, cs['seed'], cs['F'])
        fl = cs['floor']
        if fl == 'refit':                # a GIVEN floor (kinematic_optimizer.py --gt-floor) consistent with the motion: the clip's own fitted floor, rounded
            _, res0, _ = run_case(ci, data, skel, None)
            fl = (np.round(np.asarray(res0[3], dtype=float), 3), np.round(np.asarray(res0[4], dtype=float), 1))
        rec, (anim, new3d, proj2d, pn, pp, velc), tmp = run_case(ci, data, skel, fl)
        k = 'c%d_' % ci
        for name in ('poses2D', 'conf', 'poses3D', 'root_pos', 'joint_angles', 'vel', 'focal', 'pp', 'skel_offsets', 'skel_parents'):
            out[k + name] = np.asarray(data[name])
        out[k + 'given_floor'] = np.array(0 if fl is None else 1)
        if fl is not None:
            out[k + 'floor_in_n'] = fl[0]; out[k + 'floor_in_p'] = fl[1]
        out[k + 'fit_offsets'] = rec['fit_offsets']
        out[k + 'ik_rot0'] = rec['ik_rot0']; out[k + 'ik_rot'] = rec['ik_rot']; out[k + 'ik_pos'] = rec['ik_pos']
        for li, r in enumerate(rec['lsq']):
            for name, val in r.items():
                out['%slsq%d_%s' % (k, li, name)] = np.asarray(val)
        out[k + 'out_rot'] = anim.rotations.qs.copy(); out[k + 'out_pos'] = anim.positions.copy()
        out[k + 'out_pose3d'] = new3d; out[k + 'out_proj2d'] = proj2d
        out[k + 'out_floor_n'] = np.asarray(pn, dtype=float); out[k + 'out_floor_p'] = np.asarray(pp, dtype=float)
        out[k + 'out_vel'] = np.asarray(velc)
        with open(os.path.join(tmp, 'final_test.bvh')) as fh:
            out[k + 'out_bvh'] = np.array(fh.read())
        print('case %d: F %d  lsq nfev %s status %s cost %s  floor %s  contacts changed %d' % (
            ci, cs['F'], [int(r['nfev']) for r in rec['lsq']], [int(r['status']) for r in rec['lsq']], ['%.4f' % r['cost'] for r in rec['lsq']],
            np.round(out[k + 'out_floor_n'], 4), int(np.abs(out[k + 'out_vel'] - data['vel']).sum())), flush=True)
    np.savez_compressed(os.path.join(HERE, out_name), **out)
    print('wrote %s: %.2f MB' % (out_name, os.path.getsize(os.path.join(HERE, out_name)) / 1e6))

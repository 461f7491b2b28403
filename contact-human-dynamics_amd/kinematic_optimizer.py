"""Host side of the kinematic optimisation: the reference's ``optimize_trajectory``
(src/optimize/optimize_trajectory.py:522-834) for a LIST of clips, with the two expensive steps on the GPU:

* the IK initialisation (:605-617, ``JacobianInverseKinematicsCK(translate=False, iterations=200, smoothness=0, damping=7)``)
  through ``libchd_ik.so`` (include/chd_ik.h) -- all clips in one call;
* the two ``least_squares`` solves (:660-670, :779-789) through ``libchd_kinopt.so`` (include/chd_kinopt.h) -- a cluster of
  workgroups per clip (each owns a run of frames, LSMR's state in LDS), all clips of a stage in one persistent launch.  The reference assembles a dense Jacobian in Python loops (3.5 GB and minutes
  per evaluation for 100 frames); the kernel never forms it.

Host code (NumPy, negligible cost): bone lengths (``update_skeleton``, :485-520), weights and normalised 2D targets (:556-572),
the Huber floor fit and contact relabelling (:713-767; the same optimisation problem scikit-learn's ``HuberRegressor`` solves,
with SciPy's L-BFGS-B like it), the outputs (:791-834) and the three files ``kinematic_optimizer.py:184-219`` leaves for the
physics stage: ``foot_contacts.npy``, ``floor_out.txt``, ``final_test.bvh``.

There is no CPU path for the two GPU steps: without the libraries / a HIP device the constructor or the first call raises.
The monocular-total-capture ingest in front of this (``totalcap_utils``) is not part of this row.

Reproducibility: the reference's solves stop on SciPy's ``xtol`` after rejected steps of an inexact Jacobian, with LSMR at its
iteration limit -- rounding-level differences (even SciPy's own, sparse vs dense Jacobian storage) move the result by 1e-4..1e-3;
tests/test_kinopt_*.py state the tolerances.
"""
import ctypes as C
import os
import subprocess

import numpy as np

from . import skeleton_io as sio
from .ik_backproject import IkBackProject
from .ik_capi import ChdIkConfig
from .kinopt_capi import ChdKinConfig, NJ, NV, problems_to_c, results_of

_HERE = os.path.dirname(os.path.abspath(__file__))
_CSRC = os.path.join(_HERE, 'csrc')
LIB_PATH = os.environ.get('CHD_KINOPT_LIB') or os.path.join(_CSRC, 'libchd_kinopt.so')      # (override: kernel experiments with variant builds)
SOURCES = ['chd_kinopt.hip', 'chd_kinopt_kernels.hpp', 'chd_kinopt_host.hpp']
EXPORTS = ['chd_kin_version', 'chd_kin_config_default', 'chd_kin_solve_batch', 'chd_kin_last_error', 'chd_kin_last_kernel_ms', 'chd_kin_last_call_retried']

# ---- SkeletonDefinitions.py:64-137 (combined skeleton: body-25 + three spine joints) --------------------------------------------
ROOT_IDX = 8                                            # COMBINED_ROOT_IDX (data order)
FEET_IDX = np.array([4, 5, 6, 10, 11, 12])             # COMBINED_FEET_IDX (skeleton order)
SPINE_JOINTS = (13, 14, 15)                             # COMBINED_SKEL_SPINE_JOINTS
FORWARD_MAPPING = np.array([8, 12, 13, 14, 21, 19, 20, 9, 10, 11, 24, 22, 23, 25, 26, 27, 1, 0, 16, 18, 15, 17, 5, 6, 7, 2, 3, 4])
BACKWARD_MAPPING = np.argsort(FORWARD_MAPPING)
PROJ_WEIGHTS = np.array([0.1, 0.1, 0.3, 0.1, 0.1, 0.3, 0.1, 0.1, 0.1, 1.0, 0.1, 0.1, 1.0] + [0.1] * 12 + [0.0] * 3)
DATA_WEIGHTS = np.array([2.5] + [1.0] * 14 + [2.5] * 4 + [1.0] * 6 + [0.0] * 3)
STAGE_WEIGHTS = ((1000.0, 0.1, 0.5, 0.3, 10.0, 0.0),    # :630-635  projection, velocity smoothness, acceleration smoothness, data, contact velocity, floor
                 (1000.0, 0.1, 0.5, 0.3, 10.0, 10.0))   # :773-778


def build_library(force=False, verbose=False):
    """hipcc --offload-arch=gfx950 (cross-compiles without a GPU); in-tree so that the .so travels with the repo."""
    srcs = [os.path.join(_CSRC, s) for s in SOURCES] + [os.path.join(_HERE, '..', 'include', 'chd_kinopt.h')]
    if not force and os.path.exists(LIB_PATH) and all(os.path.getmtime(s) <= os.path.getmtime(LIB_PATH) for s in srcs):
        return LIB_PATH
    # -ffp-contract=off: no fused multiply-adds, so that the device rounds like the host emulation of the same source (tests/host_emu) --
    # the solve amplifies rounding differences (LSMR at its iteration limit), and the kernel is bound by the latency of its dependent steps: with contraction the
    # least-squares kernels of 256 clips took 1.80 s against 1.87 (round 5, within the run-to-run spread; the GPU tests pass either way at their tolerances)
    cmd = ['hipcc', '--offload-arch=gfx950', '-O3', '-ffp-contract=off', '-std=c++17', '-fPIC', '-shared', os.path.join(_CSRC, 'chd_kinopt.hip'), '-o', LIB_PATH]
    if verbose:
        print(' '.join(cmd))
    subprocess.check_call(cmd)
    return LIB_PATH


def load_library():
    path = os.environ.get('CHD_KINOPT_LIB', LIB_PATH)          # (another build of the same C ABI: kernel experiments)
    if not os.path.exists(path):
        raise RuntimeError('libchd_kinopt.so is not built (run __graft_entry__.build()); the kinematic optimisation has no CPU path')
    lib = C.CDLL(path)
    lib.chd_kin_version.restype = C.c_char_p
    lib.chd_kin_last_error.restype = C.c_char_p
    lib.chd_kin_last_kernel_ms.restype = C.c_double
    return lib


class KinSolver:
    """``least_squares`` of optimize_trajectory.py:660 / :779 for a batch of problems on one GPU (include/chd_kinopt.h)."""

    def __init__(self, device=0, parents=None, **overrides):
        self.lib = load_library()
        self.device = device
        self.cfg = ChdKinConfig()
        self.lib.chd_kin_config_default(C.byref(self.cfg))
        if parents is not None:
            self.cfg.parents = (C.c_int * NJ)(*[int(p) for p in parents])
        for k, v in overrides.items():
            setattr(self.cfg, k, v)

    def solve(self, problems):
        arr, keep, xs = problems_to_c(problems)
        if self.lib.chd_kin_solve_batch(C.byref(self.cfg), self.device, len(problems), arr) != 0:
            raise RuntimeError('chd_kin_solve_batch: ' + self.lib.chd_kin_last_error().decode())
        return results_of(arr, xs)

    def last_kernel_ms(self):
        return float(self.lib.chd_kin_last_kernel_ms())

    def last_call_retried(self):
        """True when the last solve() on this thread found its launch not fully resident and fell back to one workgroup per clip (include/chd_kinopt.h)"""
        return bool(self.lib.chd_kin_last_call_retried())


# ---- host steps ------------------------------------------------------------------------------------------------------------------
def update_skeleton(offsets, parents, targets):
    """:485-520.  Bone length = median over the frames of the target bone (the three spine bones: a third of root -> Spine2,
    'to avoid crunched spine from SMPL'); bone directions from the template; root offset zero."""
    par = np.asarray(parents).astype(int).copy(); par[0] = 0
    d = targets - targets[:, par]                                                  # every joint's bone at once (one median call instead of 27)
    lengths = np.median(np.sqrt(np.sum(d * d, axis=2)), axis=0)
    ds = targets[:, SPINE_JOINTS[2]] - targets[:, 0]
    lengths[list(SPINE_JOINTS)] = np.median(np.sqrt(np.sum(ds * ds, axis=1)) / 3.0)
    lengths[0] = 0.0
    out = np.array(offsets, dtype=np.float64)
    out[1:] = out[1:] / np.linalg.norm(out[1:], axis=1, keepdims=True) * lengths[1:, None]
    out[0] = 0.0
    return out


def prepare_weights(poses2d, conf, cam_center, focal):
    """:556-572: weights of the projection and data terms, 2D targets with focal length and camera centre removed."""
    F = conf.shape[0]
    proj_w = np.zeros((F, NJ)); data_w = np.zeros((F, NJ))
    p2 = np.array(poses2d, dtype=np.float64)
    proj_w[:, :25] = conf[:, :25] * PROJ_WEIGHTS[:25]
    data_w[:, :25] = (1.0 + conf[:, :25]) * DATA_WEIGHTS[:25]
    data_w[:, 25:] = (1.0 + 0.4) * DATA_WEIGHTS[25:]
    p2[:, :25, 0] = (poses2d[:, :25, 0] - cam_center[0]) / focal[0]
    p2[:, :25, 1] = (poses2d[:, :25, 1] - cam_center[1]) / focal[1]
    return p2, proj_w, data_w


def _huber_objective(w, X, y, epsilon, alpha):
    """n sigma + sum_i sigma H_eps((y_i - x_i w - c) / sigma) + alpha |w|^2 and its gradient in (w, c, sigma): the problem
    HuberRegressor(epsilon, alpha=1e-4) poses (Owen 2007), unit sample weights."""
    p = X.shape[1]
    sigma, c0, coef = w[-1], w[-2], w[:p]
    r = y - X.dot(coef) - c0
    out = np.abs(r) > epsilon * sigma
    rin = r[~out]
    sgn = np.where(r[out] < 0, -1.0, 1.0)
    n_out = int(out.sum())
    grad = np.empty(p + 2)
    grad[:p] = -2.0 / sigma * X[~out].T.dot(rin) - 2.0 * epsilon * X[out].T.dot(sgn) + 2.0 * alpha * coef
    grad[-2] = -2.0 * rin.sum() / sigma - 2.0 * epsilon * sgn.sum()
    grad[-1] = len(y) - n_out * epsilon ** 2 - rin.dot(rin) / sigma ** 2
    loss = len(y) * sigma + rin.dot(rin) / sigma + 2.0 * epsilon * np.abs(r[out]).sum() - sigma * n_out * epsilon ** 2 + alpha * coef.dot(coef)
    return loss, grad


def huber_fit(X, y, epsilon, alpha=1e-4, max_iter=100, tol=1e-5):
    """linear_model.HuberRegressor(epsilon).fit(X, y) of :719-720 / :743-744 -> coef, intercept, scale, outlier mask."""
    from scipy import optimize
    p = X.shape[1]
    w0 = np.zeros(p + 2); w0[-1] = 1.0
    bounds = [(None, None)] * (p + 1) + [(np.finfo(np.float64).eps * 10, None)]
    res = optimize.minimize(_huber_objective, w0, method='L-BFGS-B', jac=True, args=(X, y, epsilon, alpha), bounds=bounds,
                            options={'maxiter': max_iter, 'gtol': tol, 'iprint': -1})
    w = res.x
    return w[:p], w[-2], w[-1], np.abs(y - X.dot(w[:p]) - w[-2]) > w[-1] * epsilon


LAST_HUBER_BATCH = {}          # statistics of the last huber_fit_batch call (problems, not converged within max_iter, solved by huber_fit instead)


def huber_fit_batch(Xs, ys, epsilon, alpha=1e-4, max_iter=2000, gtol=1e-7):
    """`huber_fit` for many small problems at once (one per clip: the floor fits of a batch), without SciPy: the same convex objective
    (`_huber_objective`) minimised by block descent -- iteratively reweighted least squares in (coef, intercept) for the current
    scale, the scale from its stationarity condition for the current residuals -- on arrays of all problems together, until the
    gradient of every problem is below `gtol` per sample (HuberRegressor's L-BFGS-B stops at 1e-5: the result is the same minimiser,
    located more tightly: a documented deviation from the reference's stopping point).  A few hundred fits of a few hundred points take
    ~50 ms instead of 3 ms each.  A problem that does not converge within `max_iter`, or has fewer than 20 points, is handed to `huber_fit`.
    Returns a list of (coef, intercept, scale, outlier mask)."""
    B = len(Xs)
    if B == 0:
        return []
    p = Xs[0].shape[1]
    n = np.array([len(y) for y in ys])
    nmax = int(n.max())
    X = np.zeros((B, nmax, p)); Y = np.zeros((B, nmax)); M = np.zeros((B, nmax), dtype=bool)
    for b, (x, y) in enumerate(zip(Xs, ys)):
        X[b, :len(y)] = x; Y[b, :len(y)] = y; M[b, :len(y)] = True
    Mf = M.astype(np.float64)
    w = np.zeros((B, p)); c = np.zeros(B); sigma = np.ones(B)
    smin = np.finfo(np.float64).eps * 10
    active = np.ones(B, dtype=bool)

    def residual():
        return (Y - np.einsum('bnp,bp->bn', X, w) - c[:, None]) * Mf

    for _ in range(max_iter):
        r = residual()
        out = (np.abs(r) > epsilon * sigma[:, None]) & M
        inl = M & ~out
        # gradient of the objective (as _huber_objective), to decide who is done
        sgn = np.where(r < 0, -1.0, 1.0) * out
        rin = r * inl
        gw = -2.0 / sigma[:, None] * np.einsum('bnp,bn->bp', X, rin) - 2.0 * epsilon * np.einsum('bnp,bn->bp', X, sgn) + 2.0 * alpha * w
        gc = -2.0 * rin.sum(1) / sigma - 2.0 * epsilon * sgn.sum(1)
        n_out = out.sum(1)
        gs = n - n_out * epsilon ** 2 - (rin * rin).sum(1) / sigma ** 2
        gs = np.where((sigma <= smin) & (gs > 0), 0.0, gs)                      # (at the lower bound of the scale)
        gmax = np.maximum(np.abs(gw).max(1), np.maximum(np.abs(gc), np.abs(gs)))
        active = gmax > gtol * np.maximum(1, n)
        if not active.any():
            break
        # (coef, intercept): weighted least squares with the weights of the current residuals (1 / sigma inside, eps / |r| outside)
        om = np.where(inl, 1.0 / sigma[:, None], epsilon / np.maximum(np.abs(r), 1e-300)) * Mf
        Xa = np.concatenate([X, np.ones((B, nmax, 1))], axis=2)
        A = np.einsum('bni,bn,bnj->bij', Xa, om, Xa)
        A[:, np.arange(p), np.arange(p)] += alpha
        rhs = np.einsum('bni,bn->bi', Xa, om * Y)
        try:
            sol = np.linalg.solve(A, rhs[..., None])[..., 0]
        except np.linalg.LinAlgError:             # a degenerate problem (e.g. identical points): it alone gets a least-squares solution
            sol = np.stack([np.linalg.lstsq(A[b], rhs[b], rcond=None)[0] for b in range(B)])
        w = np.where(active[:, None], sol[:, :p], w); c = np.where(active, sol[:, p], c)
        # scale: n - n_out eps^2 = sum_inliers r^2 / sigma^2 (the inlier set depends on sigma: a few fixed-point steps)
        r = residual()
        r2 = r * r
        for _k in range(6):
            out = (np.abs(r) > epsilon * sigma[:, None]) & M
            den = n - out.sum(1) * epsilon ** 2
            sin = (r2 * (M & ~out)).sum(1)
            snew = np.sqrt(np.maximum(sin, 0.0) / np.maximum(den, 1e-300))
            snew = np.where(den > 0, np.maximum(snew, smin), 2.0 * sigma)      # (so many outliers that the condition has no root here: the scale is too small)
            sigma = np.where(active, snew, sigma)
    r = residual()
    res = []
    n_fallback = 0
    for b in range(B):
        # Problems the block descent did not finish (iteration limit) and small ones (fewer than 20 points: a borderline outlier label there can depend on
        # WHERE inside its tolerance a minimiser stopped) are solved exactly as the reference does -- SciPy L-BFGS-B with HuberRegressor's limits (`huber_fit`).
        if active[b] or n[b] < 20:
            res.append(huber_fit(Xs[b], ys[b], epsilon, alpha)); n_fallback += 1
        else:
            res.append((w[b].copy(), float(c[b]), float(sigma[b]), (np.abs(r[b, :n[b]]) > sigma[b] * epsilon)))
    LAST_HUBER_BATCH.update(problems=B, not_converged=int(active.sum()), solved_like_the_reference=n_fallback)
    return res


def fit_floor_batch(feet_positions):
    """`fit_floor` for a list of clips (None entries are skipped): two batched Huber fits instead of two SciPy solves per clip."""
    idx = [i for i, fp in enumerate(feet_positions) if fp is not None]
    Xs = [feet_positions[i][:, [0, 2]] for i in idx]; ys = [feet_positions[i][:, 1] for i in idx]
    plane = huber_fit_batch(Xs, ys, 1.5)
    drop = huber_fit_batch(Xs, ys, 2.2)
    out = [None] * len(feet_positions)
    for k, i in enumerate(idx):
        coef, c0 = plane[k][0], plane[k][1]
        verts = np.array([[0.0, -1.0, 0.0], [0.0, -1.0, 100.0], [100.0, -1.0, 0.0]])
        verts[:, 1] = verts[:, [0, 2]].dot(coef) + c0
        normal = np.cross(verts[2] - verts[0], verts[1] - verts[2])
        normal /= np.linalg.norm(normal)
        out[i] = (normal, verts[0].copy(), drop[k][3])
    return out


def fit_floor(feet_pos):
    """:713-767.  y = a x + b z + c through the contact positions: the epsilon = 1.5 fit gives the plane (normal from three of
    its points, point = the plane under the origin), the epsilon = 2.2 fit marks the labels to drop."""
    X = feet_pos[:, [0, 2]]; y = feet_pos[:, 1]
    coef, c0, _, _ = huber_fit(X, y, 1.5)
    verts = np.array([[0.0, -1.0, 0.0], [0.0, -1.0, 100.0], [100.0, -1.0, 0.0]])
    verts[:, 1] = verts[:, [0, 2]].dot(coef) + c0
    normal = np.cross(verts[2] - verts[0], verts[1] - verts[2])
    normal /= np.linalg.norm(normal)
    return normal, verts[0].copy(), huber_fit(X, y, 2.2)[3]


def _motion(x, offsets, parents):
    """The animation of an unknown vector (:675-683): Euler angles -> rotations, root translation -> root position."""
    F = x.shape[0]
    rot = sio.quat_from_euler(x[:, 3:].reshape(F, NJ, 3), order='xyz', world=True)
    pos = np.repeat(offsets[None], F, axis=0)
    pos[:, 0] = x[:, :3]
    return sio.Motion(rot, pos, np.tile([1.0, 0.0, 0.0, 0.0], (NJ, 1)), offsets.copy(), np.asarray(parents).copy())


def _motions_batch(xs, offsets_list, parents):
    """`_motion` + its global joint positions for the unknown vectors of many clips with one skeleton hierarchy: the quaternion and
    forward-kinematics passes run once over the frames of all clips (per clip they were a quarter of the host time of a batch).
    Returns [(Motion, positions_global)] per clip."""
    if isinstance(parents, list):               # one hierarchy per clip: batch only if they are all the same
        if not all(np.array_equal(q, parents[0]) for q in parents):
            return [(lambda m: (m, sio.positions_global(m)))(_motion(x, o, q)) for x, o, q in zip(xs, offsets_list, parents)]
        parents = parents[0]
    Fs = [x.shape[0] for x in xs]
    allx = np.concatenate(xs, axis=0)
    rot = sio.quat_from_euler_xyz_world(allx[:, 3:].reshape(-1, NJ, 3))
    pos = np.concatenate([np.repeat(o[None], F, axis=0) for o, F in zip(offsets_list, Fs)], axis=0)
    pos[:, 0] = allx[:, :3]
    parents = np.asarray(parents)
    gp = sio.positions_global_fast(rot, pos, parents)
    out, a = [], 0
    for o, F in zip(offsets_list, Fs):
        out.append((sio.Motion(rot[a:a + F].copy(), pos[a:a + F].copy(), np.tile([1.0, 0.0, 0.0, 0.0], (NJ, 1)), o.copy(), parents.copy()), gp[a:a + F]))
        a += F
    return out


class KinematicOptimizer:
    """``optimize_trajectory`` for a list of clips.  `ik` / `kin` default to the HIP libraries on `device`; tests inject the host
    emulation of the same kernel sources."""

    def __init__(self, device=0, ik=None, kin=None, parents=None):
        self.parents = np.asarray(parents) if parents is not None else None
        self.ik = ik if ik is not None else IkBackProject(device, ChdIkConfig.default(iterations=200, translate=0, damping=7.0, smoothness=0.0))
        self.kin = kin if kin is not None else KinSolver(device, parents=parents)
        self.timings = {}

    def optimize(self, clips, chunk=256, workers=2):
        """clips: dicts with poses2D (F,28,2), joint_conf_2d (F,28), poses3D (F,28,3), root_pos (F,3), joint_angles (F,28,3),
        offsets (28,3), parents (28,), ppx, ppy, camFocal (2,), velConstraints (F,28) and optionally plane_normal / plane_point --
        the arguments of optimize_trajectory (:522-526).  Returns one dict per clip (see the end of `_optimize`).

        More than `chunk` clips are processed chunk by chunk on `workers` threads: while one thread waits in a library call (ctypes releases the
        interpreter lock) the other does the host steps of its chunk -- bone lengths, floor fits, forward kinematics -- which are a third of the time of a
        chunk.  The least-squares launches of the threads take turns on the device (a launch's workgroups wait on each other: chd_kinopt.hip lets one run at
        a time), each of them fills it: a 100-frame clip is a cluster of 8 workgroups, 32 clusters are resident and draw clips from the launch's queue.  More
        than two threads do not help (measured: every launch has a tail).  A clip's result does not depend on the chunk it is in."""
        self.timings['chunks'] = []                # (the marks of THIS call's chunks: a long-lived optimizer must not accumulate them)
        if workers >= 2 and 128 < len(clips) <= chunk:
            chunk = (len(clips) + 1) // 2          # two halves, so that one half's host steps run under the other half's kernels
        if len(clips) <= chunk or workers < 2:
            return self._optimize(clips)
        from concurrent.futures import ThreadPoolExecutor
        parts = [clips[i:i + chunk] for i in range(0, len(clips), chunk)]
        with ThreadPoolExecutor(max_workers=workers) as ex:
            return [r for part in ex.map(self._optimize, parts) for r in part]

    def _optimize(self, clips):
        import time as _time
        marks = [('start', _time.perf_counter())]               # wall-clock marks of this chunk's steps (self.timings: one list per chunk; monitoring only)
        prep = []
        for cl in clips:
            F = cl['poses2D'].shape[0]
            if cl['poses2D'].shape[1] != cl['poses3D'].shape[1] or cl['poses3D'].shape[1] != NJ:
                raise ValueError('2D and 3D data must have the %d joints of the combined skeleton' % NJ)       # :538-542
            parents = np.asarray(cl['parents'])
            if self.parents is not None and not np.array_equal(parents, self.parents):
                raise ValueError('all clips of a KinematicOptimizer share one skeleton hierarchy')
            targets = cl['poses3D'][:, FORWARD_MAPPING] + cl['root_pos'][:, None]                              # :546-549
            offs = update_skeleton(cl['offsets'], parents, targets)
            p2n, pw, dw = prepare_weights(cl['poses2D'], cl['joint_conf_2d'], (cl['ppx'], cl['ppy']), cl['camFocal'])
            ang = np.linalg.norm(cl['joint_angles'], axis=2)                                                     # :589-594
            rot0 = sio.quat_from_angle_axis(ang, -(cl['joint_angles'] / (ang + 1e-10)[..., None]))
            pos = np.repeat(offs[None], F, axis=0); pos[:, 0] = cl['root_pos']
            tj = np.array([j for j in range(NJ) if j not in SPINE_JOINTS], dtype=np.int32)                       # :605-609
            given = cl.get('plane_normal') is not None and cl.get('plane_point') is not None
            prep.append(dict(F=F, parents=parents, offs=offs, p2n=p2n, pw=pw, dw=dw, vel=np.array(cl['velConstraints']), given=given,
                             floor_n=np.asarray(cl['plane_normal'], dtype=np.float64) if given else np.zeros(3),
                             floor_p=np.asarray(cl['plane_point'], dtype=np.float64) if given else np.zeros(3),
                             ik=dict(parents=parents, target_joints=tj, targets=np.swapaxes(targets[:, tj], 0, 1), rot=rot0, pos=pos)))
        marks.append(('prepared', _time.perf_counter()))
        iks = self.ik.solve([p['ik'] for p in prep])                                                             # :611-617
        marks.append(('ik', _time.perf_counter()))
        for p, (rot, pos) in zip(prep, iks):
            p['ik_rot'] = rot
            p['x'] = np.concatenate([pos[:, 0], sio.quat_to_euler_xyz(rot).reshape(p['F'], -1)], axis=1).reshape(-1)      # :638-640

        def problems(stage):
            return [dict(offsets=p['offs'], pose3d=cl['poses3D'], root_trans=cl['root_pos'], pose2d_n=p['p2n'], proj_w=p['pw'], data_w=p['dw'],
                         contact=p['vel'], floor_n=p['floor_n'], floor_p=p['floor_p'], weights=STAGE_WEIGHTS[stage], x0=p['x']) for p, cl in zip(prep, clips)]

        r1 = self.kin.solve(problems(0))                                                                         # :660-670
        marks.append(('solve_1', _time.perf_counter()))
        feet_contact = FORWARD_MAPPING[FEET_IDX]
        to_fit = []
        mg = _motions_batch([r['x'].reshape(p['F'], NV) for p, r in zip(prep, r1)], [p['offs'] for p in prep], [p['parents'] for p in prep]) if prep else []
        for p, r, (_, gp) in zip(prep, r1, mg):                                                                 # :693-709
            p['x'] = r['x']
            fv = p['vel'][:, feet_contact]
            feet_pos = gp[:, FEET_IDX][fv == 1]
            p['error'] = None
            if not p['given'] and feet_pos.shape[0] < 3:
                # (HuberRegressor.fit raises on an empty array and the reference dies with it; here the clip alone is marked and
                #  finishes without a floor term -- a bad video does not take the batch with it)
                p['error'] = 'fewer than 3 contact labels: no floor can be fitted'
            to_fit.append(feet_pos if (not p['given'] and p['error'] is None) else None)
        for p, fit in zip(prep, fit_floor_batch(to_fit)):                                                        # :713-767, all clips of the batch at once
            if fit is not None:
                p['floor_n'], p['floor_p'], outl = fit
                fv = p['vel'][:, feet_contact].copy()
                fv[fv == 1] = np.where(outl, 0, 1)                                                               # :755-767 (row-major walk = the reference's loops)
                p['vel'][:, feet_contact] = fv
        marks.append(('floor_fit', _time.perf_counter()))
        r2 = self.kin.solve(problems(1))                                                                         # :779-789
        marks.append(('solve_2', _time.perf_counter()))
        out = []
        mg = _motions_batch([b['x'].reshape(p['F'], NV) for p, b in zip(prep, r2)], [p['offs'] for p in prep], [p['parents'] for p in prep]) if prep else []
        for p, cl, a, b, (motion, gp) in zip(prep, clips, r1, r2, mg):
            new3d = gp[:, BACKWARD_MAPPING]                                                                      # :809-813
            proj = np.stack([cl['camFocal'][0] * new3d[..., 0] / new3d[..., 2] + cl['ppx'],
                             cl['camFocal'][1] * new3d[..., 1] / new3d[..., 2] + cl['ppy']], axis=-1)            # :816-830
            out.append(dict(motion=motion, pose3d=new3d, proj2d=proj, plane_normal=p['floor_n'], plane_point=p['floor_p'], velConstraints=p['vel'],
                            ik_rot=p['ik_rot'], stages=[{k: v for k, v in s.items()} for s in (a, b)], error=p['error']))
        marks.append(('outputs', _time.perf_counter()))
        self.timings.setdefault('chunks', []).append(marks)
        return out


def refined_contacts(vel):
    """kinematic_optimizer.py:184-204: F x 4 [l_heel, l_toe, r_heel, r_toe] from the relabelled body-25 feet columns 19..24."""
    f = np.asarray(vel)[:, 19:25]
    return np.stack([f[:, 2], np.logical_or(f[:, 0], f[:, 1]), f[:, 5], np.logical_or(f[:, 3], f[:, 4])], axis=1).astype(int)


def save_results(out_dir, result, names):
    """The three files the physics stage reads from `kinematic_results/` (kinematic_optimizer.py:204-219, optimize_trajectory.py:807)."""
    if result.get('error'):
        raise ValueError('no kinematic result to save: ' + result['error'])
    os.makedirs(out_dir, exist_ok=True)
    np.save(os.path.join(out_dir, 'foot_contacts'), refined_contacts(result['velConstraints']))
    n, p = result['plane_normal'], result['plane_point']
    with open(os.path.join(out_dir, 'floor_out.txt'), 'w') as fh:
        fh.write('%s %s %s\n%s %s %s' % tuple(str(float(v)) for v in (n[0], n[1], n[2], p[0], p[1], p[2])))
    sio.save_bvh(os.path.join(out_dir, 'final_test.bvh'), result['motion'], names)

"""CPU restatement of the reference's kinematic optimisation -- TEST INFRASTRUCTURE ONLY.

SURVEY 8(f) rank 3: `optimize_trajectory` (src/optimize/optimize_trajectory.py:522-834) refines the monocular-total-capture
estimate on the combined 28-joint skeleton: bone lengths from the data (`update_skeleton`, :485-520), a Jacobian-IK
initialisation (:611-617), two `scipy.optimize.least_squares(method='trf', tr_solver='lsmr', max_nfev=50)` solves of a
projection / smoothness / data / contact-velocity / floor objective (:660, :779) with a hand-written residual (:324-483) and
Jacobian (:51-322), a Huber floor fit with contact relabelling in between (:713-767).  This file restates all of it with plain
numpy, each function citing the lines it follows -- including the reference Jacobian's quirks, which decide the iterates.

Third-party algorithms the reference calls, restated here from their published form and checked against the installed
packages in tests/test_kinopt_oracle.py: SciPy 1.15.3 `least_squares` trust-region-reflective without bounds, 2-D subspace
variant (`scipy/optimize/_lsq/trf.py: trf_no_bounds`, Branch-Coleman-Li 1999) on top of LSMR (Fong & Saunders 2011,
`scipy/sparse/linalg/_isolve/lsmr.py`); scikit-learn 1.x `HuberRegressor` (Owen 2007; `sklearn/linear_model/_huber.py`).

Parity is PINNED: tests/golden/kinopt_golden.npz holds inputs, intermediates and outputs of the reference's own functions run
in the build container (tests/golden/make_kinopt_golden.py).

Only tests/ may import this module; the product path of this row is contact-human-dynamics_amd/kinematic_optimizer.py +
csrc/chd_kinopt* (HIP) and never touches this file.
"""
import numpy as np

from oracle import ik_oracle as ik

# ---- tables of SkeletonDefinitions.py:64-137 (combined skeleton: body-25 + three spine joints) -------------------------
ROOT_IDX = 8                                             # COMBINED_ROOT_IDX: MidHip in data (body-25) order
FEET_IDX = np.array([4, 5, 6, 10, 11, 12])              # skeleton order: L heel / big toe / small toe, R ...
SPINE = [13, 14, 15]
FORWARD = np.array([8, 12, 13, 14, 21, 19, 20, 9, 10, 11, 24, 22, 23, 25, 26, 27, 1, 0, 16, 18, 15, 17, 5, 6, 7, 2, 3, 4])      # skeleton joint -> data joint
BACKWARD = np.argsort(FORWARD)                           # data joint -> skeleton joint (mapping_body_25_to_combined_skel)
PROJ_W = np.array([0.1, 0.1, 0.3, 0.1, 0.1, 0.3, 0.1, 0.1, 0.1, 1.0, 0.1, 0.1, 1.0] + [0.1] * 12 + [0.0] * 3)
DATA_W = np.array([2.5] + [1.0] * 14 + [2.5] * 4 + [1.0] * 6 + [0.0] * 3)
SMOOTH_W = np.array([2.5, 2.5, 2.5, 1.5, 1.0, 2.5, 1.5, 1.0, 1.0, 2.5, 1.5, 1.0, 2.5, 1.5] + [1.0] * 11 + [1.5] * 3)
SMOOTH_VEL = np.array([1.0, 1.0, 2.0])                   # optimize_trajectory.py:43-45
SMOOTH_EULER = np.array([10.0, 10.0, 10.0])              # :46-48
NJ = 28
NV = 3 * (NJ + 1)                                        # unknowns per frame: root translation + 28 Euler triples


# ---- update_skeleton (:485-520) -----------------------------------------------------------------------------------------
def update_skeleton(offsets, parents, targets):
    """Bone lengths = median over frames of the target bone lengths (the three spine bones: a third of root -> Spine2);
    directions from the template offsets; root offset zeroed.  targets: (F, 28, 3) in skeleton order."""
    nj = len(parents)
    bones = np.zeros(nj)
    for j in range(1, nj):
        if j in SPINE:
            bones[j] = np.median(np.linalg.norm(targets[:, SPINE[2]] - targets[:, 0], axis=1) / 3.0)
        else:
            bones[j] = np.median(np.linalg.norm(targets[:, j] - targets[:, parents[j]], axis=1))
    out = np.array(offsets, dtype=np.float64)
    for j in range(1, nj):
        out[j] = out[j] / np.linalg.norm(out[j]) * bones[j]
    out[0] = 0.0
    return out


# ---- weights and normalised 2D targets (:556-572) -----------------------------------------------------------------------
def prepare_weights(poses2d, conf, pp, focal):
    F, J = conf.shape
    pw = np.zeros((F, J)); dw = np.zeros((F, J))
    p2 = np.array(poses2d, dtype=np.float64)
    pw[:, :25] = conf[:, :25] * PROJ_W[:25]
    dw[:, :25] = (1.0 + conf[:, :25]) * DATA_W[:25]
    dw[:, 25:] = (1.0 + 0.4) * DATA_W[25:]
    p2[:, :25, 0] = (poses2d[:, :25, 0] - pp[0]) / focal[0]
    p2[:, :25, 1] = (poses2d[:, :25, 1] - pp[1]) / focal[1]
    return p2, pw, dw


# ---- forward kinematics of an unknown vector (:344-359) --------------------------------------------------------------------
def fk(x, offsets, parents):
    """x: (F, 87) -> rot (F, 28, 4), y (F, 28, 3) in DATA order: the root's entry is its translation, every other joint is
    relative to the root (the skeleton's root offset is zero and the root position is written over afterwards)."""
    F = x.shape[0]
    rot = ik.quat_from_euler_xyz_world(x[:, 3:].reshape(F, NJ, 3))
    pos = np.repeat(np.asarray(offsets)[None], F, axis=0)
    gp = ik.positions_global(rot, pos, parents)
    gp[:, 0] = x[:, :3]
    return rot, pos, gp[:, BACKWARD]


class Problem:
    """Everything `fun_anim_for_projection` / `jac_anim_for_projection_sparse` take as `args` (:667-670)."""

    def __init__(self, offsets, parents, pose3d, root_trans, pose2d_n, proj_w, data_w, vel, floor_n, floor_p, weights):
        self.offsets = np.asarray(offsets, dtype=np.float64); self.parents = np.asarray(parents)
        self.pose3d = pose3d; self.root_trans = root_trans; self.pose2d_n = pose2d_n
        self.proj_w = proj_w; self.data_w = data_w; self.vel = np.asarray(vel)
        self.floor_n = np.asarray(floor_n, dtype=np.float64); self.floor_p = np.asarray(floor_p, dtype=np.float64)
        self.w = weights                                     # projWeight, smoothVel, smoothAcc, dataWeight, velWeight, floorWeight
        self.F = pose3d.shape[0]
        desc = ik.descendants_mask(self.parents)
        self.dsc = desc[:, 1:].repeat(3, axis=0).astype(int)                          # :280-283
        self.tdsc = (np.eye(NJ) + desc)[:, 1:].repeat(3, axis=0).astype(int)

    # -- residual (:324-483), vectorised; term order: projection, velocity smoothness, acceleration smoothness, data,
    #    contact velocity, floor, Euler-angle smoothness
    def fun(self, xflat):
        F = self.F
        pw_, sv, sa, dw_, vw, fw = self.w
        x = xflat.reshape(F, NV)
        _, _, y = fk(x, self.offsets, self.parents)
        yr = y[:, ROOT_IDX]
        A = y + yr[:, None]; A[:, ROOT_IDX] = yr                                      # absolute positions as the projection term forms them
        on = self.proj_w > 0
        px = A[..., 0] / A[..., 2]; py = A[..., 1] / A[..., 2]
        r1 = np.where(on[..., None], pw_ * self.proj_w[..., None] * (np.stack([px, py], axis=-1) - self.pose2d_n[..., :2]), 0.0)
        r2 = sv * SMOOTH_W[None, :, None] * SMOOTH_VEL * (y[:-1] - y[1:])
        r3 = sa * ((y[2:] - y[1:-1]) - (y[1:-1] - y[:-2]))
        tgt = np.array(self.pose3d); tgt[:, ROOT_IDX] = self.root_trans
        r4 = dw_ * (y - tgt) * self.data_w[..., None]
        Aj = y + yr[:, None]                                                          # root + joint (contacts never sit on the root)
        c = self.vel == 1
        r5 = np.where(c[:-1, :, None], vw * (Aj[:-1] - Aj[1:]), 0.0)
        r6 = np.where(c, fw * ((Aj - self.floor_p) @ self.floor_n), 0.0)
        r7 = sv * np.tile(SMOOTH_EULER, NJ + 1) * (x[:-1] - x[1:])
        return np.concatenate([r.reshape(-1) for r in (r1, r2, r3, r4, r5, r6, r7)])

    # -- dE/dP of jac_root_all_for_projection (:51-235) as a dense (rows, F, 28, 3) array in data order.  Kept quirk: the
    #    projection rows put their derivative w.r.t. the ROOT at data joint 0 (`varIndex + 0`, written for root_idx = 0) --
    #    assignments, not additions, so for joint 0 itself the two coincide.
    def jac_dp(self, y):
        F = self.F
        pw_, sv, sa, dw_, vw, fw = self.w
        yr = y[:, ROOT_IDX]
        A = y + yr[:, None]; A[:, ROOT_IDX] = yr
        n1 = F * NJ * 2; n2 = (F - 1) * NJ * 3; n3 = (F - 2) * NJ * 3; n4 = F * NJ * 3; n5 = (F - 1) * NJ * 3; n6 = F * NJ
        Jp = np.zeros((n1 + n2 + n3 + n4 + n5 + n6, F, NJ, 3))
        for f in range(F):
            for j in range(NJ):
                w = pw_ * self.proj_w[f, j]
                if not self.proj_w[f, j] > 0:
                    continue
                g = A[f, j, 2]
                for cc in range(2):
                    row = (f * NJ + j) * 2 + cc
                    num = A[f, j, cc]
                    for jj in (0, j):
                        Jp[row, f, jj, cc] = w / g
                        Jp[row, f, jj, 2] = -w * num / (g * g)
        o = n1
        for f in range(F - 1):
            for j in range(NJ):
                for cc in range(3):
                    s = sv * SMOOTH_W[j] * SMOOTH_VEL[cc]
                    Jp[o, f, j, cc] = s; Jp[o, f + 1, j, cc] = -s; o += 1
        for f in range(F - 2):
            for j in range(NJ):
                for cc in range(3):
                    Jp[o, f, j, cc] = sa; Jp[o, f + 1, j, cc] = -2 * sa; Jp[o, f + 2, j, cc] = sa; o += 1
        for f in range(F):
            for j in range(NJ):
                for cc in range(3):
                    Jp[o, f, j, cc] = dw_ * self.data_w[f, j]; o += 1
        for f in range(F - 1):
            for j in range(NJ):
                if self.vel[f, j] == 1:
                    for cc in range(3):
                        Jp[o + cc, f, ROOT_IDX, cc] = vw; Jp[o + cc, f + 1, ROOT_IDX, cc] = -vw
                        Jp[o + cc, f, j, cc] = vw; Jp[o + cc, f + 1, j, cc] = -vw
                o += 3
        for f in range(F):
            for j in range(NJ):
                if self.vel[f, j] == 1:
                    for cc in range(3):
                        Jp[o, f, ROOT_IDX, cc] = fw * self.floor_n[cc]
                        Jp[o, f, j, cc] = fw * self.floor_n[cc]
                o += 1
        return Jp

    # -- jac_anim_for_projection_sparse (:237-322), dense
    def jac(self, xflat):
        F = self.F
        sv = self.w[1]
        x = xflat.reshape(F, NV)
        rot, pos, y = fk(x, self.offsets, self.parents)
        Jp = self.jac_dp(y)
        gt = ik.transforms_global(rot, pos, self.parents)
        gp = gt[:, :, :3, 3] / gt[:, :, 3, 3, None]
        gr = ik.quat_from_matrix(gt)
        jdr = ik._jacobian(x[:, 3:], gp, gr, self.parents, np.arange(1, NJ), self.dsc, self.tdsc, False)         # (F, 81, 84)
        rows = Jp.shape[0]
        J = np.zeros((rows + (F - 1) * NV, F * NV))
        for f in range(F):
            jt2 = Jp[:, f, FORWARD, :].reshape(rows, NJ * 3)                          # skeleton order (:295-297)
            J[:rows, f * NV:f * NV + 3] = jt2[:, :3]
            J[:rows, f * NV + 3:(f + 1) * NV] = jt2[:, 3:] @ jdr[f]
        o = rows
        se = sv * np.tile(SMOOTH_EULER, NJ + 1)
        for f in range(F - 1):
            for k in range(NV):
                J[o, f * NV + k] = se[k]; J[o, (f + 1) * NV + k] = -se[k]; o += 1
        return J


# ---- LSMR (Fong & Saunders 2011; scipy lsmr with its default tolerances) ----------------------------------------------------
def _sym_ortho(a, b):
    """Stable Givens rotation of Choi's SymOrtho as SciPy's LSQR / LSMR use it."""
    if b == 0:
        return np.sign(a), 0.0, abs(a)
    if a == 0:
        return 0.0, np.sign(b), abs(b)
    if abs(b) > abs(a):
        tau = a / b
        s = np.sign(b) / np.sqrt(1 + tau * tau)
        return s * tau, s, b / s
    tau = b / a
    c = np.sign(a) / np.sqrt(1 + tau * tau)
    return c, c * tau, a / c


def lsmr(matvec, rmatvec, b, n, damp=0.0, atol=1e-6, btol=1e-6, conlim=1e8, maxiter=None):
    """min ||A x - b||^2 + damp^2 ||x||^2.  Returns (x, istop, itn)."""
    m = b.size
    if maxiter is None:
        maxiter = min(m, n)
    u = b.copy()
    normb = np.linalg.norm(b)
    x = np.zeros(n)
    beta = normb
    if beta > 0:
        u = (1 / beta) * u
        v = rmatvec(u)
        alpha = np.linalg.norm(v)
    else:
        v = np.zeros(n); alpha = 0.0
    if alpha > 0:
        v = (1 / alpha) * v
    itn = 0
    zetabar = alpha * beta; alphabar = alpha
    rho = rhobar = cbar = 1.0; sbar = 0.0
    h = v.copy(); hbar = np.zeros(n)
    betadd = beta; betad = 0.0; rhodold = 1.0; tautildeold = 0.0; thetatilde = 0.0; zeta = 0.0; d = 0.0
    normA2 = alpha * alpha; maxrbar = 0.0; minrbar = 1e100
    istop = 0
    ctol = 1.0 / conlim if conlim > 0 else 0.0
    if alpha * beta == 0:
        return x, istop, itn
    while itn < maxiter:
        itn += 1
        u = u * (-alpha) + matvec(v)
        beta = np.linalg.norm(u)
        if beta > 0:
            u = u * (1 / beta)
            v = v * (-beta) + rmatvec(u)
            alpha = np.linalg.norm(v)
            if alpha > 0:
                v = v * (1 / alpha)
        chat, shat, alphahat = _sym_ortho(alphabar, damp)
        rhoold = rho
        c, s, rho = _sym_ortho(alphahat, beta)
        thetanew = s * alpha
        alphabar = c * alpha
        rhobarold = rhobar; zetaold = zeta
        thetabar = sbar * rho
        rhotemp = cbar * rho
        cbar, sbar, rhobar = _sym_ortho(cbar * rho, thetanew)
        zeta = cbar * zetabar
        zetabar = -sbar * zetabar
        hbar = hbar * (-(thetabar * rho / (rhoold * rhobarold))) + h
        x = x + (zeta / (rho * rhobar)) * hbar
        h = h * (-(thetanew / rho)) + v
        betaacute = chat * betadd; betacheck = -shat * betadd
        betahat = c * betaacute; betadd = -s * betaacute
        thetatildeold = thetatilde
        ctildeold, stildeold, rhotildeold = _sym_ortho(rhodold, thetabar)
        thetatilde = stildeold * rhobar
        rhodold = ctildeold * rhobar
        betad = -stildeold * betad + ctildeold * betahat
        tautildeold = (zetaold - thetatildeold * tautildeold) / rhotildeold
        taud = (zeta - thetatilde * tautildeold) / rhodold
        d = d + betacheck * betacheck
        normr = np.sqrt(d + (betad - taud) ** 2 + betadd * betadd)
        normA2 = normA2 + beta * beta
        normA = np.sqrt(normA2)
        normA2 = normA2 + alpha * alpha
        maxrbar = max(maxrbar, rhobarold)
        if itn > 1:
            minrbar = min(minrbar, rhobarold)
        condA = max(maxrbar, rhotemp) / min(minrbar, rhotemp)
        normar = abs(zetabar)
        normx = np.linalg.norm(x)
        test1 = normr / normb
        test2 = normar / (normA * normr) if normA * normr != 0 else np.inf
        test3 = 1.0 / condA
        t1 = test1 / (1 + normA * normx / normb)
        rtol = btol + atol * normA * normx / normb
        if itn >= maxiter: istop = 7
        if 1 + test3 <= 1: istop = 6
        if 1 + test2 <= 1: istop = 5
        if 1 + t1 <= 1: istop = 4
        if test3 <= ctol: istop = 3
        if test2 <= atol: istop = 2
        if test1 <= rtol: istop = 1
        if istop > 0:
            break
    return x, istop, itn


# ---- trust-region-reflective without bounds, 2-D subspace (scipy trf_no_bounds, x_scale = 1, linear loss) -----------------
def solve_trust_region_2d(B, g, Delta):
    """min 1/2 p^T B p + g^T p, |p| <= Delta, in two dimensions: the Newton step if B is positive definite and the step
    inside, otherwise the best boundary point (roots of a quartic in the tangent half-angle parametrisation)."""
    det = B[0, 0] * B[1, 1] - B[0, 1] * B[0, 1]
    if B[0, 0] > 0 and det > 0:                                                       # Cholesky succeeds
        l00 = np.sqrt(B[0, 0]); l10 = B[0, 1] / l00
        d11 = B[1, 1] - l10 * l10
        if d11 > 0:
            l11 = np.sqrt(d11)
            z0 = -g[0] / l00; z1 = (-g[1] - l10 * z0) / l11
            p1 = z1 / l11; p0 = (z0 - l10 * p1) / l00
            if p0 * p0 + p1 * p1 <= Delta ** 2:
                return np.array([p0, p1]), True
    a = B[0, 0] * Delta ** 2; b = B[0, 1] * Delta ** 2; c = B[1, 1] * Delta ** 2
    d = g[0] * Delta; f = g[1] * Delta
    t = np.roots(np.array([-b + d, 2 * (a - c + f), 6 * b, 2 * (-a + c + f), -b - d]))
    t = np.real(t[np.isreal(t)])
    p = Delta * np.vstack((2 * t / (1 + t ** 2), (1 - t ** 2) / (1 + t ** 2)))
    value = 0.5 * np.sum(p * B.dot(p), axis=0) + np.dot(g, p)
    return p[:, np.argmin(value)], False


def trf_lsmr(fun, jac, x0, ftol=1e-8, xtol=1e-8, gtol=1e-12, max_nfev=50, trace=None, lsmr_maxiter=None):
    """least_squares(fun, x0, jac=jac, method='trf', tr_solver='lsmr', bounds=(-inf, inf), max_nfev=50, gtol=1e-12) as the
    reference calls it (:660-670, :779-789).  Returns (x, cost, nfev, njev, status)."""
    from scipy import sparse                                                          # (storage only: the Jacobian is ~2 % dense)
    x = x0.copy()
    f = fun(x); nfev = 1
    J = sparse.csr_matrix(jac(x)); njev = 1
    cost = 0.5 * np.dot(f, f)
    g = J.T.dot(f)
    Delta = np.linalg.norm(x0)
    if Delta == 0:
        Delta = 1.0
    status = None
    while True:
        g_norm = np.linalg.norm(g, ord=np.inf)
        if g_norm < gtol:
            status = 1
        if status is not None or nfev == max_nfev:
            break
        # regularisation: the Cauchy step's decrease inside the region sets the LSMR damping
        v = J.dot(-g)
        a = 0.5 * np.dot(v, v); b = -np.dot(g, g)
        to_tr = Delta / np.linalg.norm(g)
        ts = [0.0, to_tr]
        if a != 0:
            ext = -0.5 * b / a
            if 0 < ext < to_tr:
                ts.append(ext)
        ts = np.asarray(ts)
        ag_value = np.min(ts * (a * ts + b))
        reg_term = -ag_value / Delta ** 2
        gn, _, itn = lsmr(J.dot, J.T.dot, f, x.size, damp=reg_term ** 0.5, maxiter=lsmr_maxiter)      # (lsmr_maxiter: test knob; SciPy: min(m, n))
        if trace is not None:
            trace.append(('iter', cost, Delta, g_norm, reg_term ** 0.5, itn))
        S = np.vstack((g, gn)).T
        S, _ = np.linalg.qr(S, mode='reduced')
        JS = J.dot(S)
        B_S = np.dot(JS.T, JS)
        g_S = S.T.dot(g)
        actual_reduction = -1
        while actual_reduction <= 0 and nfev < max_nfev:
            p_S, _ = solve_trust_region_2d(B_S, g_S, Delta)
            step = S.dot(p_S)
            Js = J.dot(step)
            predicted_reduction = -(0.5 * np.dot(Js, Js) + np.dot(step, g))
            x_new = x + step
            f_new = fun(x_new); nfev += 1
            step_norm = np.linalg.norm(step)
            if not np.all(np.isfinite(f_new)):
                Delta = 0.25 * step_norm
                continue
            cost_new = 0.5 * np.dot(f_new, f_new)
            actual_reduction = cost - cost_new
            if trace is not None:
                trace.append(('try', p_S.copy(), predicted_reduction, actual_reduction, B_S.copy(), g_S.copy()))
            if predicted_reduction > 0:
                ratio = actual_reduction / predicted_reduction
            elif predicted_reduction == actual_reduction == 0:
                ratio = 1
            else:
                ratio = 0
            Delta_new = Delta
            if ratio < 0.25:
                Delta_new = 0.25 * step_norm
            elif ratio > 0.75 and step_norm > 0.95 * Delta:
                Delta_new = Delta * 2.0
            ftol_ok = actual_reduction < ftol * cost and ratio > 0.25
            xtol_ok = step_norm < xtol * (xtol + np.linalg.norm(x))
            status = 4 if (ftol_ok and xtol_ok) else 2 if ftol_ok else 3 if xtol_ok else None
            if status is not None:
                break
            Delta = Delta_new
        if actual_reduction > 0:
            x = x_new; f = f_new; cost = cost_new
            J = sparse.csr_matrix(jac(x)); njev += 1
            g = J.T.dot(f)
    return x, cost, nfev, njev, (0 if status is None else status)


# ---- Huber floor fit (sklearn HuberRegressor defaults: alpha 1e-4, max_iter 100, tol 1e-5, fit_intercept) --------------------
def huber_loss_and_gradient(w, X, y, epsilon, alpha):
    """sklearn/linear_model/_huber.py `_huber_loss_and_gradient`, unit sample weights:
    n sigma + sum_i sigma H_eps((y_i - x_i w - c) / sigma) + alpha |w|^2 over (w, c, sigma)."""
    n, p = X.shape
    sigma = w[-1]; c = w[-2]; coef = w[:p]
    r = y - X.dot(coef) - c
    ar = np.abs(r)
    out = ar > epsilon * sigma
    n_out = np.count_nonzero(out)
    grad = np.zeros(p + 2)
    loss_out = 2.0 * epsilon * np.sum(ar[out]) - sigma * n_out * epsilon ** 2
    rin = r[~out]
    loss_in = np.dot(rin, rin) / sigma
    grad[:p] = 2.0 / sigma * (-(X[~out].T.dot(rin)))
    sgn = np.where(r[out] < 0, -1.0, 1.0)
    grad[:p] -= 2.0 * epsilon * X[out].T.dot(sgn)
    grad[:p] += alpha * 2.0 * coef
    grad[-1] = n - n_out * epsilon ** 2 - np.dot(rin, rin) / sigma ** 2
    grad[-2] = -2.0 * np.sum(rin) / sigma - 2.0 * epsilon * np.sum(sgn)
    return n * sigma + loss_in + loss_out + alpha * np.dot(coef, coef), grad


def huber_fit(X, y, epsilon, alpha=1e-4, max_iter=100, tol=1e-5):
    """HuberRegressor.fit: L-BFGS-B from (0, ..., 0, sigma = 1), sigma bounded below by 10 eps(float64).
    Returns coef, intercept, sigma, outlier mask."""
    from scipy import optimize
    p = X.shape[1]
    w0 = np.zeros(p + 2); w0[-1] = 1.0
    bounds = np.tile([-np.inf, np.inf], (p + 2, 1)); bounds[-1][0] = np.finfo(np.float64).eps * 10
    res = optimize.minimize(huber_loss_and_gradient, w0, method='L-BFGS-B', jac=True, args=(X, y, epsilon, alpha),
                            options={'maxiter': max_iter, 'gtol': tol, 'iprint': -1}, bounds=bounds)
    w = res.x
    r = np.abs(y - X.dot(w[:p]) - w[-2])
    return w[:p], w[-2], w[-1], r > w[-1] * epsilon


def fit_floor(feet_pos):
    """:713-767: plane through three points of the epsilon = 1.5 fit; outliers of the epsilon = 2.2 fit lose their label."""
    X = feet_pos[:, [0, 2]]; yv = feet_pos[:, 1]
    coef, c0, _, _ = huber_fit(X, yv, 1.5)
    verts = np.array([[0.0, -1.0, 0.0], [0.0, -1.0, 100.0], [100.0, -1.0, 0.0]])
    for i in range(3):
        verts[i, 1] = verts[i, [0, 2]].dot(coef) + c0
    nrm = np.cross(verts[2] - verts[0], verts[1] - verts[2])
    nrm /= np.linalg.norm(nrm)
    _, _, _, outl = huber_fit(X, yv, 2.2)
    return nrm, verts[0], outl


# ---- optimize_trajectory (:522-834) --------------------------------------------------------------------------------------------
def optimize_trajectory(poses2d, conf, poses3d, root_pos, joint_angles, offsets, parents, pp, focal, vel, plane_normal=None, plane_point=None,
                        ik_iterations=200, lsq=trf_lsmr):
    F = poses2d.shape[0]
    given_floor = plane_normal is not None and plane_point is not None
    targets = poses3d[:, FORWARD] + root_pos[:, None]                                                 # :546-549 (skeleton order, absolute)
    offs = update_skeleton(offsets, parents, targets)
    p2n, pw, dw = prepare_weights(poses2d, conf, pp, focal)
    # IK initialisation from the SMPL angles (:581-620)
    ang = np.linalg.norm(joint_angles, axis=2)
    axis = -(joint_angles / (ang + 1e-10)[..., None])
    rot0 = ik.quat_from_angle_axis(ang, axis)
    pos = np.repeat(offs[None], F, axis=0); pos[:, 0] = root_pos
    tj = np.array([j for j in range(NJ) if j not in SPINE])
    rot, pos = ik.ik_ck(rot0, pos, parents, tj, np.swapaxes(targets[:, tj], 0, 1), iterations=ik_iterations, damping=7.0, smoothness=0.0, translate=False)
    x = np.concatenate([pos[:, 0], ik.quat_to_euler_xyz(rot).reshape(F, -1)], axis=1).reshape(-1)
    vel = np.array(vel)
    nfloor = np.zeros(3) if not given_floor else np.asarray(plane_normal, dtype=np.float64)
    pfloor = np.zeros(3) if not given_floor else np.asarray(plane_point, dtype=np.float64)
    prob = Problem(offs, parents, poses3d, root_pos, p2n, pw, dw, vel, nfloor, pfloor, (1000.0, 0.1, 0.5, 0.3, 10.0, 0.0))
    stage = [dict(x0=x.copy())]
    x, cost, nfev, njev, status = lsq(prob.fun, prob.jac, x)
    stage[0].update(x=x.copy(), cost=cost, nfev=nfev, status=status)
    # floor (:693-767)
    rotf, posf, _ = fk(x.reshape(F, NV), offs, parents)
    posf[:, 0] = x.reshape(F, NV)[:, :3]
    gp = ik.positions_global(rotf, posf, parents)
    feet_contact = FORWARD[FEET_IDX]
    sel = vel[:, feet_contact] == 1
    feet_pos = gp[:, FEET_IDX][sel]
    if not given_floor:
        nfloor, pfloor, outl = fit_floor(feet_pos)
        fv = vel[:, feet_contact]
        k = 0
        for fr in range(F):
            for q in range(len(feet_contact)):
                if fv[fr, q] == 1:
                    if outl[k]:
                        fv[fr, q] = 0
                    k += 1
        vel[:, feet_contact] = fv
    prob2 = Problem(offs, parents, poses3d, root_pos, p2n, pw, dw, vel, nfloor, pfloor, (1000.0, 0.1, 0.5, 0.3, 10.0, 10.0))
    stage.append(dict(x0=x.copy()))
    x, cost, nfev, njev, status = lsq(prob2.fun, prob2.jac, x)
    stage[1].update(x=x.copy(), cost=cost, nfev=nfev, status=status)
    X = x.reshape(F, NV)
    rotf, posf, _ = fk(X, offs, parents)
    posf[:, 0] = X[:, :3]
    gp = ik.positions_global(rotf, posf, parents)
    new3d = gp[:, BACKWARD]
    proj = np.stack([focal[0] * new3d[..., 0] / new3d[..., 2] + pp[0], focal[1] * new3d[..., 1] / new3d[..., 2] + pp[1]], axis=-1)
    return dict(offsets=offs, rot=rotf, pos=posf, pose3d=new3d, proj2d=proj, floor_n=nfloor, floor_p=pfloor, vel=vel, stages=stage, ik_rot=rot)

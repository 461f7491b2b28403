"""Scratch prototype of the interior-point algorithm (dense numpy) used to settle the
algorithm that oracle/ipm_solver.hpp and the HIP solver both implement."""
import sys, time
sys.path.insert(0, '.')
import numpy as np
import chd_amd
from chd_amd.synth import make_walk
from oracle.oracle import OracleProblem

INF = 1e19

def ipm(p, max_iter=300, tol=1e-8, verbose=True, delta_w=1e-4, delta_c=1e-9, use_lam_init=False, dw_min=1e-9, max_bt=3, exact=False, mu0=0.1, dur_damp=1.0):
    n, m = p.n, p.m
    off = p.var_offsets()
    Dw = np.ones(n)
    fscale = 73.0*9.8/4
    Dw[off[6]:off[10]] = 1.0/fscale**2
    Dw[off[10]:] = dur_damp
    x = p.get_x()
    cl, cu = p.bounds()
    f, g, c, J, H = p.eval(x, hess=True)
    # --- scaling (IPOPT gradient-based) ---
    gmax = np.abs(g).max()
    sf = min(1.0, 100.0 / gmax) if gmax > 100 else 1.0
    rmax = np.abs(J).max(axis=1) if m else np.zeros(0)
    sc = np.where(rmax > 100, 100.0 / np.maximum(rmax, 1e-300), 1.0)
    sc = np.maximum(sc, 1e-8)
    eq = (cu - cl) <= 0.0
    iq = ~eq
    hasL = iq & (cl > -INF); hasU = iq & (cu < INF)
    l = np.where(cl > -INF, cl * sc, -np.inf); u = np.where(cu < INF, cu * sc, np.inf)
    relax = 1e-8
    l = np.where(iq & (cl > -INF), l - relax * np.maximum(1, np.abs(l)), l)
    u = np.where(iq & (cu < INF), u + relax * np.maximum(1, np.abs(u)), u)
    def ev(x, hess=False):
        f, g, c, J, H = p.eval(x, jac=True, hess=hess)
        return sf * f, sf * g, sc * c, sc[:, None] * J, (sf * H if hess else None)
    def ev0(x):
        f, g, c, J, H = p.eval(x, jac=False)
        return sf * f, sc * c
    f, g, c, J, H = ev(x, True)
    # --- slack init ---
    k1 = k2 = 1e-2
    s = c.copy()
    pl = np.where(hasL, np.minimum(k1 * np.maximum(1, np.abs(np.where(hasL, l, 0))), np.where(hasU & hasL, k2 * (u - l), np.inf)), 0)
    pu = np.where(hasU, np.minimum(k1 * np.maximum(1, np.abs(np.where(hasU, u, 0))), np.where(hasU & hasL, k2 * (u - l), np.inf)), 0)
    s = np.where(hasL, np.maximum(s, l + pl), s)
    s = np.where(hasU, np.minimum(s, u - pu), s)
    mu = mu0
    zL = np.where(hasL, mu / np.where(hasL, s - l, 1.0), 0.0); zU = np.where(hasU, mu / np.where(hasU, u - s, 1.0), 0.0)
    lam = np.zeros(m)
    lam[iq] = (zU - zL)[iq]
    nu = 1.0
    kappa_eps, kappa_mu, theta_mu, smax = 10.0, 0.2, 1.5, 100.0
    tau_min = 0.99
    nfact = 0
    def resid(c, s):
        r = c.copy()
        r[eq] = c[eq] - l[eq]
        r[iq] = c[iq] - s[iq]
        return r
    def barrier(s, mu):
        b = 0.0
        b -= mu * np.sum(np.log(s[hasL] - l[hasL]))
        b -= mu * np.sum(np.log(u[hasU] - s[hasU]))
        return b
    def errors(mu_):
        dual = g + J.T @ lam
        dual_s = np.zeros(m); dual_s[iq] = (-lam - zL + zU)[iq]
        r = resid(c, s)
        compL = np.where(hasL, (s - l) * zL - mu_, 0.0) if hasL.any() else np.zeros(1)
        compU = np.where(hasU, (u - s) * zU - mu_, 0.0) if hasU.any() else np.zeros(1)
        nz = hasL.sum() + hasU.sum()
        sd = max(smax, (np.abs(lam).sum() + np.abs(zL).sum() + np.abs(zU).sum()) / max(1, m + nz)) / smax
        scmp = max(smax, (np.abs(zL).sum() + np.abs(zU).sum()) / max(1, nz)) / smax
        e_d = max(np.abs(dual).max(), np.abs(dual_s).max() if m else 0) / sd
        e_p = np.abs(r).max() if m else 0.0
        e_c = max(np.abs(compL).max(), np.abs(compU).max()) / scmp
        return max(e_d, e_p, e_c), e_d, e_p, e_c
    status = -1
    it = 0
    dw = delta_w
    for it in range(max_iter):
        E0 = errors(0.0)
        if verbose and (it % 1 == 0):
            print(f'{it:4d} f={f/sf:.6e} E0={E0[0]:.2e} (d {E0[1]:.1e} p {E0[2]:.1e} c {E0[3]:.1e}) mu={mu:.1e} nu={nu:.1e}', end='')
        if E0[0] <= tol:
            status = 0
            if verbose: print()
            break
        # barrier update
        while True:
            Emu = errors(mu)[0]
            if Emu <= kappa_eps * mu and mu > tol / 10:
                mu = max(tol / 10, min(kappa_mu * mu, mu ** theta_mu))
            else:
                break
        tau = max(tau_min, 1 - mu)
        # --- KKT ---
        sL = np.where(hasL, s - l, 1.0); sU = np.where(hasU, u - s, 1.0)
        Sigma = np.where(hasL, zL / sL, 0) + np.where(hasU, zU / sU, 0)
        r = resid(c, s)
        rs = -lam - np.where(hasL, mu / sL, 0) + np.where(hasU, mu / sU, 0)   # barrier gradient wrt s (with lam)
        D = np.where(eq, delta_c, 1.0 / np.maximum(Sigma, 1e-300) + 0.0)
        rhs_x = -(g + J.T @ lam)
        rhs_c = -np.where(eq, r, r + rs / np.maximum(Sigma, 1e-300))
        HL = H
        if exact and m:
            hh = 1e-6
            HL = H.copy()
            lam_s = lam * sc   # J scaled = sc*J_raw -> (sc*J)^T lam
            Hc = np.zeros((n, n))
            for j in range(n):
                xp = x.copy(); xp[j] += hh; xm = x.copy(); xm[j] -= hh
                Jp = p.eval(xp, jac=True)[3]; Jm = p.eval(xm, jac=True)[3]
                Hc[:, j] = ((Jp - Jm).T @ lam_s) / (2 * hh)
            Hc = 0.5 * (Hc + Hc.T)
            HL = H + Hc
        for attempt in range(12):
            K = np.block([[HL + dw * np.diag(Dw), J.T], [J, -np.diag(D)]])
            if exact:
                ev_ = np.linalg.eigvalsh(K)
                npos = (ev_ > 0).sum()
                if npos != n:
                    dw = max(dw * 10, 1e-6)
                    if verbose: print(f' [inertia {npos}!={n} -> dw={dw:.1e}]', end='')
                    continue
            sol = np.linalg.solve(K, np.concatenate([rhs_x, rhs_c])); nfact += 1
            # iterative refinement
            res = np.concatenate([rhs_x, rhs_c]) - K @ sol
            sol += np.linalg.solve(K, res)
            dx = sol[:n]; dlam = sol[n:]
            ds = np.where(iq, (dlam - rs) / np.maximum(Sigma, 1e-300), 0.0)
            dzL = np.where(hasL, mu / sL - zL - zL / sL * ds, 0.0)
            dzU = np.where(hasU, mu / sU - zU + zU / sU * ds, 0.0)
            # fraction to boundary
            a_pr = 1.0
            mL = hasL & (ds < 0); mU = hasU & (ds > 0)
            if mL.any(): a_pr = min(a_pr, np.min(-tau * sL[mL] / ds[mL]))
            if mU.any(): a_pr = min(a_pr, np.min(tau * sU[mU] / ds[mU]))
            a_du = 1.0
            mL = hasL & (dzL < 0); mU = hasU & (dzU < 0)
            if mL.any(): a_du = min(a_du, np.min(-tau * zL[mL] / dzL[mL]))
            if mU.any(): a_du = min(a_du, np.min(-tau * zU[mU] / dzU[mU]))
            # merit
            cn = np.abs(r).sum()
            dphi_bar = g @ dx - np.sum(np.where(hasL, mu / sL * ds, 0)) + np.sum(np.where(hasU, mu / sU * ds, 0))
            dHd = dx @ (HL @ dx) + dw * dx @ (Dw*dx) + np.sum(Sigma * ds * ds)
            rho = 0.1
            if cn > 1e-14:
                nu_trial = (dphi_bar + 0.5 * max(dHd, 0)) / ((1 - rho) * cn)
                if nu_trial > nu: nu = nu_trial * 1.1 + 1e-8   # hmm
            Dphi = dphi_bar - nu * cn
            phi0 = f + barrier(s, mu) + nu * cn
            a = a_pr
            ok = False
            nls = 0
            while nls <= max_bt:
                xt = x + a * dx; st = s + a * ds
                ft, ct = ev0(xt)
                phit = ft + barrier(st, mu) + nu * np.abs(resid(ct, st)).sum()
                if phit <= phi0 + 1e-4 * a * Dphi + 1e-12 * abs(phi0):
                    ok = True; break
                a *= 0.5; nls += 1
            if ok: break
            dw = dw * 10
            if verbose: print(f' [ls fail -> dw={dw:.1e}]', end='')
        if not ok:
            if verbose: print(' LINE SEARCH FAILED')
            status = -2
            break
        if attempt == 0 and nls == 0: dw = max(dw_min, dw / 3)
        x = x + a * dx; s = s + a * ds
        lam = lam + a * dlam
        zL = zL + a_du * dzL; zU = zU + a_du * dzU
        # reset z (kappa_sigma)
        ks = 1e10
        zL = np.where(hasL, np.clip(zL, mu / (ks * (s - l)), ks * mu / (s - l)), 0.0)
        zU = np.where(hasU, np.clip(zU, mu / (ks * (u - s)), ks * mu / (u - s)), 0.0)
        f, g, c, J, H = ev(x, True)
        if verbose: print(f'  a_pr={a_pr:.2e} a={a:.2e} a_du={a_du:.2e} |dx|={np.abs(dx).max():.2e} ls={nls} dw={dw:.0e}')
    p.set_x(x)
    return status, it, x, dict(f=f / sf, E=errors(0.0), mu=mu, nfact=nfact)

if __name__ == '__main__':
    F = int(sys.argv[1]) if len(sys.argv) > 1 else 45
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    seq = make_walk(seed=seed, F=F, randomize=True)
    p = OracleProblem(seq)
    mu0 = float(sys.argv[4]) if len(sys.argv) > 4 else 0.1
    dd = float(sys.argv[5]) if len(sys.argv) > 5 else 1.0
    for st in [0, 1, 2, 3, 4]:
        p.set_stage(st)
        t0 = time.time()
        status, it, x, info = ipm(p, verbose=("-v" in sys.argv), tol=float(sys.argv[3]) if len(sys.argv)>3 else 1e-3, mu0=mu0 if st>0 else 0.1, dur_damp=dd)
        print(f'== stage {st}: status {status} iters {it} f={info["f"]:.6e} E={info["E"][0]:.2e} time {time.time()-t0:.1f}s')

// =============================================================================
// ORACLE (test infrastructure only — see nlp_model.hpp).
//
// CPU restatement of the interior-point solve that the reference delegates to
// ifopt::IpoptSolver / IPOPT (phys_optim.cpp:567-580): a primal-dual log-barrier
// method on the slack formulation  c_E(x)=0, c_I(x)-s=0, l<=s<=u  [Waechter &
// Biegler 2006, the published IPOPT algorithm]:
//   * gradient-based NLP scaling (nlp_scaling_max_gradient = 100),
//   * bound relaxation 1e-8, slack push kappa_1 = kappa_2 = 1e-2,
//   * monotone Fiacco-McCormick barrier update (kappa_mu 0.2, theta_mu 1.5,
//     kappa_eps 10), fraction-to-boundary tau = max(0.99, 1-mu),
//   * termination on IPOPT's scaled optimality error E_0 <= tol (tol = 1e-3 is
//     the reference's option, phys_optim.cpp:578),
//   * symmetric indefinite (quasi-definite) KKT solve  [H+dw*Dw  J^T; J  -D].
//   * an optional stall guard (IpmOptions::stall_window; optimality error not halved within that many iterations => status -2).
// Deliberate differences from IPOPT (the reference binary cannot be run here, so
// its iterates are not reproducible anyway; SURVEY.md §7 "Hard parts"):
//   * Hessian: Gauss-Newton Hessian of the sum-of-squares objective, plus the exact duration-duration
//     block of the Lagrangian Hessian when the phase durations are variables (nlp_model.hpp), plus an
//     adaptive Levenberg damping dw*Dw, instead of L-BFGS(6);
//   * globalisation: l1 merit function (penalty parameter recomputed per step) with backtracking and a
//     second-order correction instead of the filter + restoration phase; a failed attempt first retries with the
//     Gauss-Newton model (no exact constraint-curvature blocks), then with more damping;
//   * mu_init = 1e-3 and mu-based bound multipliers for the warm-started stages;
//   * linear algebra: bordered banded LDL^T without pivoting in a time ordering
//     (stance positions / durations in the border) instead of MA57.
// The HIP solver (contact-human-dynamics_amd/csrc) implements the same algorithm
// with the same constants; parity tests compare the two.
// =============================================================================
#pragma once
#include "nlp_model.hpp"
#include <cstdlib>

namespace orc {

struct IpmOptions {
  int max_iter = 500;
  double ref_tol = 1e-3;        // IPOPT tol (phys_optim.cpp:578)
  double mu_init_cold = 0.1;    // IPOPT default mu_init (first stage)
  double mu_init_warm = 1e-3;   // warm-started stages
  double delta_w0 = 1e-4, delta_w_min = 1e-9, delta_c = 1e-9, delta_w_max = 1e8;
  double dw_grow_second = 1.5;      // growth of the damping after an iteration that needed the second model (CHD_DW_GROW_SECOND)
  double ratio_low = 0.0;           // > 0 (0.25 = chd_config.damping_rule 1): an ACCEPTED step below this actual / predicted merit reduction raises the damping too
  double constr_viol_tol = 1e-4;   // IPOPT default (unscaled)
  int max_backtrack = 3;
  bool use_soc = true;
  int max_attempts = 12;
  bool verbose = false;
  // IPOPT-like mode, ORACLE ONLY -- never the shipped algorithm: limited-memory BFGS(6) Hessian of the Lagrangian
  // (phys_optim.cpp:572 hessian_approximation = limited-memory) folded into the KKT solve by the Sherman-Morrison-Woodbury formula,
  // IPOPT's mu_init = 0.1 on every stage.  Used by tests/tools/ipopt_like_distance.py to measure how far a solver of IPOPT's family lands
  // from the shipped Gauss-Newton algorithm at the same tolerance.
  bool lbfgs = false;
  int lbfgs_history = 6;
  // ... with IPOPT's globalisation (round 5): the FILTER line search of Waechter & Biegler (2006, sec. 2.3; IPOPT's constants gamma_theta 1e-5, gamma_phi 1e-8,
  // delta 1, s_theta 1.1, s_phi 2.3, eta_phi 1e-8, theta_max / theta_min = 1e4 / 1e-4 x max(1, theta(x0)), gamma_alpha 0.05), second-order correction on the
  // first trial step, the filter reset at every barrier update -- in place of the l1 merit function.  What is still NOT IPOPT: there is no restoration phase
  // (when the step length falls below alpha_min the damping delta_w grows tenfold and the step is recomputed: IPOPT would minimise the constraint violation
  // instead), and MA57 is a banded L D L^T.  `acceptable_tol` = 1e-6 / 15 iterations is implemented and can never fire here: the
  // reference's tol = 1e-3 is LOOSER than the acceptable level, so the regular test always stops first.
  bool filter = false;
  int stall_window = 0;         // > 0: stall guard (chd_config.stall_window); 0 = off, as IPOPT has no such rule
  bool inertia_retry = true;    // a factorisation that had to replace a pivot counts as a failed attempt (see chd_kernels.hpp, CHD_INERTIA_RETRY)
};

struct IpmResult {
  int status = -2;
  int iters = 0;
  double kkt_error = 0, constr_viol = 0, objective = 0, mu = 0;
  int n_factor = 0, N = 0, bandwidth = 0, stalled = 0;
};

// -----------------------------------------------------------------------------
// Bordered banded LDL^T (no pivoting).  K = [A B; B^T C], A banded (Nb, half-bandwidth w).
// -----------------------------------------------------------------------------
struct BorderedBandLDL {
  int Nb = 0, b = 0, w = 0;
  std::vector<double> A;     // Nb x (w+1), row i holds columns i-w..i
  std::vector<double> B;     // Nb x b
  std::vector<double> C;     // b x b (lower used)
  std::vector<double> A0, B0, C0;   // unfactored copies for residuals
  std::vector<double> Y;     // A^{-1} B
  std::vector<double> S;     // Schur complement factor
  std::vector<char> sign;    // expected pivot sign per position (+1 primal, -1 dual), size Nb+b
  int n_bad_pivots = 0;

  double& a(int i, int j) { return A[(size_t)i * (w + 1) + (j - i + w)]; }
  void resize(int Nb_, int b_, int w_) {
    Nb = Nb_; b = b_; w = w_;
    A.assign((size_t)Nb * (w + 1), 0.0); B.assign((size_t)Nb * b, 0.0); C.assign((size_t)b * b, 0.0);
  }
  void add(int i, int j, double v) {          // symmetric entry, positions in [0, Nb+b)
    if (i < j) std::swap(i, j);
    if (i < Nb) a(i, j) += v;
    else if (j < Nb) B[(size_t)j * b + (i - Nb)] += v;
    else C[(size_t)(i - Nb) * b + (j - Nb)] += v;
  }
  void band_solve(double* r) const {          // in-place A^{-1} r using the factor in A
    const int W = w + 1;
    for (int i = 0; i < Nb; ++i) {
      int k0 = std::max(0, i - w);
      const double* Li = &A[(size_t)i * W + (k0 - i + w)];
      double sum = r[i];
      for (int k = k0; k < i; ++k) sum -= Li[k - k0] * r[k];
      r[i] = sum;
    }
    for (int i = 0; i < Nb; ++i) r[i] /= A[(size_t)i * W + w];
    for (int i = Nb - 1; i >= 0; --i) {
      int k0 = std::max(0, i - w);
      const double* Li = &A[(size_t)i * W + (k0 - i + w)];
      double xi = r[i];
      for (int k = k0; k < i; ++k) r[k] -= Li[k - k0] * xi;
    }
  }
  // every pivot is compared with the sign expected at its position; a pivot of the wrong sign (or below 1e-14) is replaced and counted (chd_kernels.hpp,
  // "Inertia handling": stricter than the inertia, and dependent on the elimination order -- which is the kernel's)
  bool fix_pivot(double& d, int pos) {
    double sg = sign[pos];
    if (!(d * sg > 1e-14)) { d = sg * 1e-10; ++n_bad_pivots; return false; }
    return true;
  }
  void factor() {
    A0 = A; B0 = B; C0 = C;
    n_bad_pivots = 0;
    const int W = w + 1;
    std::vector<double> v(w + 1);
    for (int i = 0; i < Nb; ++i) {
      int k0 = std::max(0, i - w);
      double* Ai = &A[(size_t)i * W + (k0 - i + w)];     // Ai[k-k0] = A(i,k)
      for (int j = k0; j < i; ++j) {
        int j0 = std::max(0, j - w);
        int s0 = std::max(k0, j0);
        const double* Lj = &A[(size_t)j * W + (j0 - j + w)];
        double sum = Ai[j - k0];
        for (int k = s0; k < j; ++k) sum -= v[k - k0] * Lj[k - j0];
        v[j - k0] = sum;                                  // = L(i,j) * D(j)
      }
      double d = Ai[i - k0];
      for (int j = k0; j < i; ++j) {
        double dj = A[(size_t)j * W + w];
        double l = v[j - k0] / dj;
        d -= l * v[j - k0];
        Ai[j - k0] = l;
      }
      fix_pivot(d, i);
      Ai[i - k0] = d;
    }
    // Y = A^{-1} B, S = C - B^T Y
    Y = B;
    if (b > 0) {
      std::vector<double> col(Nb);
      for (int c = 0; c < b; ++c) {
        for (int i = 0; i < Nb; ++i) col[i] = B[(size_t)i * b + c];
        band_solve(col.data());
        for (int i = 0; i < Nb; ++i) Y[(size_t)i * b + c] = col[i];
      }
      S.assign((size_t)b * b, 0.0);
      for (int r = 0; r < b; ++r)
        for (int c = 0; c <= r; ++c) {
          double sum = C[(size_t)r * b + c];
          for (int i = 0; i < Nb; ++i) sum -= B0[(size_t)i * b + r] * Y[(size_t)i * b + c];
          S[(size_t)r * b + c] = sum;
        }
      // dense LDL^T of S (lower), in place
      for (int i = 0; i < b; ++i) {
        for (int j = 0; j < i; ++j) {
          double sum = S[(size_t)i * b + j];
          for (int k = 0; k < j; ++k) sum -= S[(size_t)i * b + k] * S[(size_t)k * b + k] * S[(size_t)j * b + k];
          S[(size_t)i * b + j] = sum / S[(size_t)j * b + j];
        }
        double d = S[(size_t)i * b + i];
        for (int k = 0; k < i; ++k) d -= S[(size_t)i * b + k] * S[(size_t)i * b + k] * S[(size_t)k * b + k];
        fix_pivot(d, Nb + i);
        S[(size_t)i * b + i] = d;
      }
    }
  }
  void solve_once(const double* rhs, double* x) const {
    std::vector<double> y(rhs, rhs + Nb);
    band_solve(y.data());
    std::vector<double> x2(b);
    for (int c = 0; c < b; ++c) {
      double sum = rhs[Nb + c];
      for (int i = 0; i < Nb; ++i) sum -= B0[(size_t)i * b + c] * y[i];
      x2[c] = sum;
    }
    for (int i = 0; i < b; ++i) { double sum = x2[i]; for (int k = 0; k < i; ++k) sum -= S[(size_t)i * b + k] * x2[k]; x2[i] = sum; }
    for (int i = 0; i < b; ++i) x2[i] /= S[(size_t)i * b + i];
    for (int i = b - 1; i >= 0; --i) { double xi = x2[i]; for (int k = 0; k < i; ++k) x2[k] -= S[(size_t)i * b + k] * xi; }
    for (int i = 0; i < Nb; ++i) {
      double sum = y[i];
      for (int c = 0; c < b; ++c) sum -= Y[(size_t)i * b + c] * x2[c];
      x[i] = sum;
    }
    for (int c = 0; c < b; ++c) x[Nb + c] = x2[c];
  }
  void matvec0(const double* x, double* y) const {   // y = K0 x (unfactored)
    const int W = w + 1;
    for (int i = 0; i < Nb + b; ++i) y[i] = 0.0;
    for (int i = 0; i < Nb; ++i) {
      int k0 = std::max(0, i - w);
      for (int k = k0; k < i; ++k) { double v = A0[(size_t)i * W + (k - i + w)]; y[i] += v * x[k]; y[k] += v * x[i]; }
      y[i] += A0[(size_t)i * W + w] * x[i];
      for (int c = 0; c < b; ++c) { double v = B0[(size_t)i * b + c]; y[i] += v * x[Nb + c]; y[Nb + c] += v * x[i]; }
    }
    for (int r = 0; r < b; ++r) {
      for (int c = 0; c < r; ++c) { double v = C0[(size_t)r * b + c]; y[Nb + r] += v * x[Nb + c]; y[Nb + c] += v * x[Nb + r]; }
      y[Nb + r] += C0[(size_t)r * b + r] * x[Nb + r];
    }
  }
  // up to `refine` steps of iterative refinement; a step is taken only while the residual is above refine_skip x |rhs| (max norms)
  static constexpr double refine_skip = 1e-8;
  void solve(const double* rhs, double* x, int refine = 2) const {
    const int N = Nb + b;
    solve_once(rhs, x);
    std::vector<double> r(N), dx(N);
    for (int it = 0; it < refine; ++it) {
      matvec0(x, r.data());
      double rn = 0.0, bn = 0.0;
      for (int i = 0; i < N; ++i) { r[i] = rhs[i] - r[i]; rn = std::fmax(rn, std::fabs(r[i])); bn = std::fmax(bn, std::fabs(rhs[i])); }
      if (rn <= refine_skip * bn) break;
      solve_once(r.data(), dx.data());
      for (int i = 0; i < N; ++i) x[i] += dx[i];
    }
  }
};


// -----------------------------------------------------------------------------
inline IpmResult ipm_solve(Problem& P, const IpmOptions& opt) {
  IpmResult res;
  const int n = P.n, m = P.m;
  const double INF = 1e19;
  std::vector<double> x(n), g(n), c(m), J((size_t)m * n), H((size_t)n * n), graw(n), craw(m);
  P.get_x(x.data());
  double fraw = 0;

  // ---- variable descriptors, damping metric --------------------------------
  std::vector<double> vtime; std::vector<char> vborder, vkind;
  P.var_descriptors(vtime, vborder, vkind);
  std::vector<double> Dw(n, 1.0);
  const double fscale = P.in.mass * kGravity / 4.0;
  for (int j = 0; j < n; ++j) if (vkind[j] == 1) Dw[j] = 1.0 / (fscale * fscale);

  // ---- first evaluation, scaling ------------------------------------------
  P.eval(x.data(), &fraw, graw.data(), craw.data(), J.data(), H.data());
  double gmax = 0; for (int j = 0; j < n; ++j) gmax = std::max(gmax, std::fabs(graw[j]));
  const double sf = gmax > 100.0 ? 100.0 / gmax : 1.0;
  std::vector<double> sc(m, 1.0);
  for (int i = 0; i < m; ++i) {
    double rm = 0; for (int j = 0; j < n; ++j) rm = std::max(rm, std::fabs(J[(size_t)i * n + j]));
    if (rm > 100.0) sc[i] = std::max(100.0 / rm, 1e-8);
  }
  std::vector<char> eq(m), hasL(m), hasU(m);
  std::vector<double> l(m), u(m);
  for (int i = 0; i < m; ++i) {
    eq[i] = (P.cu[i] - P.cl[i]) <= 0.0;
    hasL[i] = !eq[i] && P.cl[i] > -INF; hasU[i] = !eq[i] && P.cu[i] < INF;
    l[i] = P.cl[i] > -INF ? P.cl[i] * sc[i] : -HUGE_VAL;
    u[i] = P.cu[i] < INF ? P.cu[i] * sc[i] : HUGE_VAL;
    if (hasL[i]) l[i] -= 1e-8 * std::max(1.0, std::fabs(l[i]));
    if (hasU[i]) u[i] += 1e-8 * std::max(1.0, std::fabs(u[i]));
  }
  auto apply_scaling = [&](bool with_jac) {
    for (int i = 0; i < m; ++i) c[i] = sc[i] * craw[i];
    if (with_jac) {
      for (int j = 0; j < n; ++j) g[j] = sf * graw[j];
      for (int i = 0; i < m; ++i) { double s_ = sc[i]; if (s_ != 1.0) { double* r = &J[(size_t)i * n]; for (int j = 0; j < n; ++j) r[j] *= s_; } }
      if (sf != 1.0) for (size_t k = 0; k < H.size(); ++k) H[k] *= sf;
    }
  };
  apply_scaling(true);
  double f = sf * fraw;

  // ---- KKT ordering (time-banded, border = long-range variables) ------------
  // structural pattern: union of |J| at x0 and at a deterministic perturbation
  std::vector<char> pat((size_t)m * n, 0);
  {
    std::vector<double> xp(x), cp(m), Jp((size_t)m * n);
    P.pattern_mode = true;          // (structure, not values: nlp_model.hpp)
    P.eval(x.data(), nullptr, nullptr, cp.data(), Jp.data(), nullptr);
    for (size_t k = 0; k < pat.size(); ++k) pat[k] = Jp[k] != 0.0;
    unsigned s_ = 12345u;
    for (int j = 0; j < n; ++j) { s_ = s_ * 1664525u + 1013904223u; double r = ((s_ >> 8) & 0xFFFF) / 65535.0 - 0.5; xp[j] += (vkind[j] == 1 ? 10.0 : vkind[j] == 2 ? 0.0 : 1e-2) * r; }
    P.eval(xp.data(), nullptr, nullptr, cp.data(), Jp.data(), nullptr);
    for (size_t k = 0; k < pat.size(); ++k) pat[k] |= (Jp[k] != 0.0);
    P.pattern_mode = false;
    P.set_x(x.data());
  }
  std::vector<int> band_vars;
  for (int j = 0; j < n; ++j) if (!vborder[j]) band_vars.push_back(j);
  std::stable_sort(band_vars.begin(), band_vars.end(), [&](int a, int b2) { return vtime[a] < vtime[b2]; });
  std::vector<int> rank(n, -1);
  for (size_t r = 0; r < band_vars.size(); ++r) rank[band_vars[r]] = (int)r;
  std::vector<int> row_last(m, -1);
  for (int i = 0; i < m; ++i)
    for (int j = 0; j < n; ++j) if (pat[(size_t)i * n + j] && rank[j] >= 0) row_last[i] = std::max(row_last[i], rank[j]);
  std::vector<std::vector<int>> rows_after(band_vars.size());
  std::vector<int> border_rows;
  for (int i = 0; i < m; ++i) { if (row_last[i] >= 0) rows_after[row_last[i]].push_back(i); else border_rows.push_back(i); }
  std::vector<int> pos_var(n, -1), pos_row(m, -1);
  int Nb = 0;
  for (size_t r = 0; r < band_vars.size(); ++r) { pos_var[band_vars[r]] = Nb++; for (int i : rows_after[r]) pos_row[i] = Nb++; }
  // border variables in the order of the start time of their phase (a shared stance position: its stance phase; a duration: its own phase), ties by index --
  // the same rule as the kernel's table builder (chd_model.hpp), because the pivot test of BorderedBandLDL depends on the elimination order
  int bcount = 0;
  {
    std::vector<int> bvars;
    for (int j = 0; j < n; ++j) if (vborder[j]) bvars.push_back(j);
    std::stable_sort(bvars.begin(), bvars.end(), [&](int a, int b2) { return vtime[a] < vtime[b2]; });
    for (int j : bvars) pos_var[j] = Nb + bcount++;
  }
  for (int i : border_rows) pos_row[i] = Nb + bcount++;
  const int N = Nb + bcount;
  BorderedBandLDL K;
  K.sign.assign(N, 1);
  for (int i = 0; i < m; ++i) K.sign[pos_row[i]] = -1;
  res.N = N;

  // ---- slack / multiplier initialisation ------------------------------------
  double mu = (P.stage == 0 || opt.lbfgs) ? opt.mu_init_cold : opt.mu_init_warm;
  std::vector<double> s(m, 0.0), zL(m, 0.0), zU(m, 0.0), lam(m, 0.0);
  for (int i = 0; i < m; ++i) {
    if (eq[i]) continue;
    double si = c[i];
    const double k1 = 1e-2, k2 = 1e-2;
    if (hasL[i]) { double pl = k1 * std::max(1.0, std::fabs(l[i])); if (hasU[i]) pl = std::min(pl, k2 * (u[i] - l[i])); si = std::max(si, l[i] + pl); }
    if (hasU[i]) { double pu = k1 * std::max(1.0, std::fabs(u[i])); if (hasL[i]) pu = std::min(pu, k2 * (u[i] - l[i])); si = std::min(si, u[i] - pu); }
    s[i] = si;
    if (hasL[i]) zL[i] = mu / (si - l[i]);
    if (hasU[i]) zU[i] = mu / (u[i] - si);
    lam[i] = zU[i] - zL[i];
  }
  double nu = 1.0, dw = opt.delta_w0;
  const double kappa_eps = 10.0, kappa_mu = 0.2, theta_mu = 1.5, smax = 100.0, tol = opt.ref_tol;

  auto resid = [&](const std::vector<double>& cc, const std::vector<double>& ss, std::vector<double>& r) {
    for (int i = 0; i < m; ++i) r[i] = eq[i] ? cc[i] - l[i] : cc[i] - ss[i];
  };
  auto barrier = [&](const std::vector<double>& ss, double mu_) {
    double b = 0;
    for (int i = 0; i < m; ++i) { if (hasL[i]) b -= mu_ * std::log(ss[i] - l[i]); if (hasU[i]) b -= mu_ * std::log(u[i] - ss[i]); }
    return b;
  };
  std::vector<double> r(m), dualx(n);
  double e_d = 0, e_p = 0, e_c = 0, e_p_unscaled = 0;
  auto errors = [&](double mu_) {
    for (int j = 0; j < n; ++j) dualx[j] = g[j];
    for (int i = 0; i < m; ++i) { double li = lam[i]; if (li == 0.0) continue; const double* Jr = &J[(size_t)i * n]; for (int j = 0; j < n; ++j) dualx[j] += Jr[j] * li; }
    double d1 = 0; for (int j = 0; j < n; ++j) d1 = std::max(d1, std::fabs(dualx[j]));
    double sumlam = 0, sumz = 0; int nz = 0;
    e_c = 0;
    for (int i = 0; i < m; ++i) {
      sumlam += std::fabs(lam[i]);
      if (!eq[i]) d1 = std::max(d1, std::fabs(-lam[i] - zL[i] + zU[i]));
      if (hasL[i]) { sumz += std::fabs(zL[i]); ++nz; e_c = std::max(e_c, std::fabs((s[i] - l[i]) * zL[i] - mu_)); }
      if (hasU[i]) { sumz += std::fabs(zU[i]); ++nz; e_c = std::max(e_c, std::fabs((u[i] - s[i]) * zU[i] - mu_)); }
    }
    resid(c, s, r);
    e_p = 0; e_p_unscaled = 0;
    for (int i = 0; i < m; ++i) { e_p = std::max(e_p, std::fabs(r[i])); e_p_unscaled = std::max(e_p_unscaled, std::fabs(r[i]) / sc[i]); }
    double sd = std::max(smax, (sumlam + sumz) / std::max(1, m + nz)) / smax;
    double scmp = std::max(smax, sumz / std::max(1, nz)) / smax;
    e_d = d1 / sd; e_c = e_c / scmp;
    return std::max(e_d, std::max(e_p, e_c));
  };

  std::vector<double> Sigma(m), rs(m), D(m), rhs(N), sol(N), dx(n), dlam(m), ds(m), dzL(m), dzU(m);
  std::vector<double> xt(n), st(m), ct(m), rt(m), Hdx(n), rhs2(N), sol2(N), xs(n), ss2(m), lraw(m);
  int status = -1, it = 0;
  // limited-memory BFGS state (IPOPT-like mode): pairs (s, y) in the scaled variables, sigma = s^T y / s^T s of the newest pair
  std::vector<std::vector<double>> lb_S, lb_Y;
  double lb_sigma = 1.0;
  std::vector<double> lb_xold, lb_Jold, lb_gold;
  std::vector<double> e0_hist(150, 0.0);
  double last_alpha = 0; int last_nls = 0, last_att = 0; bool last_soc = false;
  // Damping safeguard: the dual infeasibility rising in `dual_rise_k` consecutive iterations means the quadratic model is being
  // trusted too far (full steps along a direction whose constraint curvature the Gauss-Newton Hessian lacks: seen on standing-up
  // clips, where the duration stage drifted for 2 000 iterations with delta_w at 1e-8) -- the Levenberg damping is raised to at
  // least 4e-6 (x 4 above 1e-6).  Never fires on the 160 flat / tilted bench sequences of the fixture; -9 % iterations on the 40 hard ones.
  constexpr int dual_rise_k = 6;
  double ed_prev = -1.0; int ed_rise = 0;
  // filter (IPOPT-like mode with opt.filter): pairs (theta, phi) a trial point must improve on; reset when mu changes
  std::vector<std::pair<double, double>> filt;
  double filt_mu = -1.0, theta_max = 0, theta_min = 0;
  int acceptable_count = 0;
  for (it = 0; it < opt.max_iter; ++it) {
    double E0 = errors(0.0);
    if (ed_prev >= 0 && e_d > ed_prev) ++ed_rise; else ed_rise = 0;
    ed_prev = e_d;
    if (ed_rise >= dual_rise_k) { dw = std::min(opt.delta_w_max, std::max(dw, 1e-6) * 4.0); ed_rise = 0; }
    if (opt.verbose) { int wj = 0; double wv2 = 0; for (int j = 0; j < n; ++j) if (std::fabs(dualx[j]) > wv2) { wv2 = std::fabs(dualx[j]); wj = j; } int spl = 10; for (int q = 0; q < 10; ++q) if (wj >= P.sp[q].var_off && wj < P.sp[q].var_off + P.sp[q].n_var) spl = q; std::printf("[dual worst var %d spline %d t=%.2f val %.2e] ", wj, spl, vtime[wj], dualx[wj]); }
    if (opt.verbose) { int wi = 0; double wv = 0; for (int i = 0; i < m; ++i) { double v = std::fabs(eq[i] ? c[i] - l[i] : c[i] - s[i]); if (v > wv) { wv = v; wi = i; } } std::printf("[worst row %d fam %d eq %d c=%.4e s=%.4e l=%.3e u=%.3e lam=%.3e] ", wi, P.row_family[wi], (int)eq[wi], c[wi], s[wi], l[wi], u[wi], lam[wi]); }
    if (opt.verbose) std::printf("%4d f=%.6e E0=%.2e (d %.1e p %.1e pu %.1e c %.1e) mu=%.1e nu=%.1e dw=%.1e | last a=%.2e nls=%d att=%d soc=%d\n", it, f / sf, E0, e_d, e_p, e_p_unscaled, e_c, mu, nu, dw, last_alpha, last_nls, last_att, (int)last_soc);
    if (E0 <= tol && e_p_unscaled <= opt.constr_viol_tol) { status = 0; break; }
    if (opt.filter) {      // IPOPT's acceptable-point exit: acceptable_tol 1e-6 for 15 consecutive iterations (never reached before the test above at tol 1e-3)
      acceptable_count = (E0 <= 1e-6 && e_p_unscaled <= 1e-2) ? acceptable_count + 1 : 0;
      if (acceptable_count >= 15) { status = 1; break; }
    }
    // stall guard (optional, chd_config.stall_window): no factor-2 reduction of the optimality error over the last W iterations
    // -> give up (status -2) instead of running to the iteration cap; stage 3 then takes the reference's stage-4 fallback
    if (opt.stall_window > 0) {
      const int W = std::min(opt.stall_window, N);
      if ((int)e0_hist.size() < W) e0_hist.assign(W, 0.0);
      if (it >= W && E0 > 0.5 * e0_hist[it % W]) { status = -2; res.stalled = 1; break; }
      e0_hist[it % W] = E0;
    }
    while (true) {
      double Emu = errors(mu);
      if (Emu <= kappa_eps * mu && mu > tol / 10) mu = std::max(tol / 10, std::min(kappa_mu * mu, std::pow(mu, theta_mu)));
      else break;
    }
    const double tau = std::max(0.99, 1 - mu);
    resid(c, s, r);
    for (int i = 0; i < m; ++i) {
      if (eq[i]) { Sigma[i] = 0; rs[i] = 0; D[i] = opt.delta_c; continue; }
      double sig = 0, q = -lam[i];
      if (hasL[i]) { sig += zL[i] / (s[i] - l[i]); q -= mu / (s[i] - l[i]); }
      if (hasU[i]) { sig += zU[i] / (u[i] - s[i]); q += mu / (u[i] - s[i]); }
      Sigma[i] = std::max(sig, 1e-300); rs[i] = q; D[i] = 1.0 / Sigma[i];
    }
    // rhs
    for (int j = 0; j < n; ++j) dualx[j] = g[j];
    for (int i = 0; i < m; ++i) { double li = lam[i]; if (li == 0.0) continue; const double* Jr = &J[(size_t)i * n]; for (int j = 0; j < n; ++j) dualx[j] += Jr[j] * li; }
    for (int j = 0; j < n; ++j) rhs[pos_var[j]] = -dualx[j];
    for (int i = 0; i < m; ++i) rhs[pos_row[i]] = eq[i] ? -r[i] : -(r[i] + rs[i] / Sigma[i]);
    // bandwidth of the current pattern
    int w = 0;
    if (!opt.lbfgs) for (int a = 0; a < n; ++a) { if (pos_var[a] >= Nb) continue; const double* Hr = &H[(size_t)a * n]; for (int b2 = 0; b2 < a; ++b2) if (Hr[b2] != 0.0 && pos_var[b2] < Nb) w = std::max(w, std::abs(pos_var[a] - pos_var[b2])); }
    for (int i = 0; i < m; ++i) { if (pos_row[i] >= Nb) continue; const double* Jr = &J[(size_t)i * n]; for (int j = 0; j < n; ++j) if (Jr[j] != 0.0 && pos_var[j] < Nb) w = std::max(w, std::abs(pos_row[i] - pos_var[j])); }
    res.bandwidth = std::max(res.bandwidth, w);

    bool ok = false, used_soc = false; double alpha = 0, a_du = 1.0; int nls = 0, attempt = 0;
    double ratio_num = 0.0, ratio_den = 0.0;      // actual / predicted reduction of the merit function by the accepted step (IpmOptions::ratio_low)
    // Second model of an iteration: when an attempt with the exact blocks (heel-distance curvature, node x duration block) fails and the exact duration-duration block fails -- wrong inertia, or the line
    // search runs out of backtracks -- the Hessian is rebuilt ONCE without them (plain Gauss-Newton, positive semi-definite by construction) and the attempt is repeated
    // with the same damping; only if that fails too does the damping grow.  (Rounds 2-3 kept max(lam, 0) of the heel-distance block instead; near-redundant
    // rows carry multipliers of +-1e4 that cancel in the exact block but not in its clipped copy: profiles/r04_curvature_study.md.)
    bool second_used = false;
    auto second_model = [&]() {
      if (second_used || it == 0 || opt.lbfgs) return false;          // (the first model of a stage has no curvature terms)
      second_used = true;
      P.eval(x.data(), &fraw, graw.data(), craw.data(), J.data(), H.data(), lraw.data(), true);
      apply_scaling(true);
      return true;
    };
    for (attempt = 0; attempt < opt.max_attempts; ++attempt) {
      K.resize(Nb, bcount, w);
      if (opt.lbfgs) {
        for (int a = 0; a < n; ++a) K.add(pos_var[a], pos_var[a], lb_sigma + dw * Dw[a]);      // B = sigma I - W M^-1 W^T: the low-rank part enters through kkt_solve
      } else
      for (int a = 0; a < n; ++a) {
        const double* Hr = &H[(size_t)a * n];
        for (int b2 = 0; b2 < a; ++b2) if (Hr[b2] != 0.0) K.add(pos_var[a], pos_var[b2], Hr[b2]);
        K.add(pos_var[a], pos_var[a], Hr[a] + dw * Dw[a]);
      }
      for (int i = 0; i < m; ++i) {
        const double* Jr = &J[(size_t)i * n];
        for (int j = 0; j < n; ++j) if (Jr[j] != 0.0) K.add(pos_row[i], pos_var[j], Jr[j]);
        K.add(pos_row[i], pos_row[i], -D[i]);
      }
      K.factor(); ++res.n_factor;
      if (opt.verbose && K.n_bad_pivots > 0) std::printf("   attempt %d: %d bad pivots (dw %.1e, second model %d)\n", attempt, K.n_bad_pivots, dw, (int)second_used);
      if (opt.inertia_retry && K.n_bad_pivots > 0) { if (second_model()) continue; dw *= 10.0; if (dw > opt.delta_w_max) break; continue; }
      const int lk = opt.lbfgs ? (int)lb_S.size() : 0;
      std::vector<double> lbZ, lbC;          // Z = K_sigma^-1 What (N x 2k, column major), C = M - What^T Z (2k x 2k)
      if (lk > 0) {
        const int k2 = 2 * lk;
        lbZ.assign((size_t)N * k2, 0.0);
        std::vector<double> col(N);
        for (int c2 = 0; c2 < k2; ++c2) {
          std::fill(col.begin(), col.end(), 0.0);
          const std::vector<double>& v = c2 < lk ? lb_S[c2] : lb_Y[c2 - lk];
          for (int j = 0; j < n; ++j) col[pos_var[j]] = (c2 < lk ? lb_sigma : 1.0) * v[j];
          K.solve(col.data(), &lbZ[(size_t)c2 * N], 1);
        }
        lbC.assign((size_t)k2 * k2, 0.0);
        auto Wd = [&](int c2, int j) { return (c2 < lk ? lb_sigma * lb_S[c2][j] : lb_Y[c2 - lk][j]); };
        for (int a = 0; a < lk; ++a)
          for (int b2 = 0; b2 < lk; ++b2) {
            double ss = 0, sy = 0;
            for (int j = 0; j < n; ++j) { ss += lb_S[a][j] * lb_S[b2][j]; sy += lb_S[a][j] * lb_Y[b2][j]; }
            lbC[(size_t)a * k2 + b2] = lb_sigma * ss;                               // sigma S^T S
            if (a > b2) { lbC[(size_t)a * k2 + lk + b2] = sy; lbC[(size_t)(lk + b2) * k2 + a] = sy; }      // L (strictly lower) and L^T
            if (a == b2) lbC[(size_t)(lk + a) * k2 + lk + a] = -sy;                 // -D
          }
        for (int a = 0; a < k2; ++a)
          for (int b2 = 0; b2 < k2; ++b2) {
            double acc = 0;
            for (int j = 0; j < n; ++j) acc += Wd(a, j) * lbZ[(size_t)b2 * N + pos_var[j]];
            lbC[(size_t)a * k2 + b2] -= acc;
          }
      }
      auto kkt_solve = [&](const double* r_, double* z_) {
        K.solve(r_, z_, 1);      // one step of iterative refinement
        if (lk == 0) return;
        const int k2 = 2 * lk;
        std::vector<double> t(k2), A(lbC);
        for (int a = 0; a < k2; ++a) { double acc = 0; for (int j = 0; j < n; ++j) acc += (a < lk ? lb_sigma * lb_S[a][j] : lb_Y[a - lk][j]) * z_[pos_var[j]]; t[a] = acc; }
        // dense solve A t = t (Gaussian elimination with partial pivoting; 2k <= 12)
        for (int c2 = 0; c2 < k2; ++c2) {
          int piv = c2; for (int r2 = c2 + 1; r2 < k2; ++r2) if (std::fabs(A[(size_t)r2 * k2 + c2]) > std::fabs(A[(size_t)piv * k2 + c2])) piv = r2;
          if (piv != c2) { for (int q = 0; q < k2; ++q) std::swap(A[(size_t)c2 * k2 + q], A[(size_t)piv * k2 + q]); std::swap(t[c2], t[piv]); }
          const double d = A[(size_t)c2 * k2 + c2];
          if (d == 0.0) continue;
          for (int r2 = c2 + 1; r2 < k2; ++r2) { const double f2 = A[(size_t)r2 * k2 + c2] / d; if (f2 == 0.0) continue; for (int q = c2; q < k2; ++q) A[(size_t)r2 * k2 + q] -= f2 * A[(size_t)c2 * k2 + q]; t[r2] -= f2 * t[c2]; }
        }
        for (int c2 = k2 - 1; c2 >= 0; --c2) { double acc = t[c2]; for (int q = c2 + 1; q < k2; ++q) acc -= A[(size_t)c2 * k2 + q] * t[q]; const double d = A[(size_t)c2 * k2 + c2]; t[c2] = d != 0.0 ? acc / d : 0.0; }
        for (int a = 0; a < k2; ++a) { const double ta = t[a]; const double* Zc = &lbZ[(size_t)a * N]; for (int i = 0; i < N; ++i) z_[i] += Zc[i] * ta; }
      };
      kkt_solve(rhs.data(), sol.data());
      for (int j = 0; j < n; ++j) dx[j] = sol[pos_var[j]];
      for (int i = 0; i < m; ++i) dlam[i] = sol[pos_row[i]];
      double a_pr = 1.0; a_du = 1.0;
      double dbar = 0, sSds = 0;
      for (int i = 0; i < m; ++i) {
        ds[i] = 0; dzL[i] = 0; dzU[i] = 0;
        if (eq[i]) continue;
        ds[i] = (dlam[i] - rs[i]) / Sigma[i];
        if (hasL[i]) { double sl = s[i] - l[i]; dzL[i] = mu / sl - zL[i] - zL[i] / sl * ds[i]; if (ds[i] < 0) a_pr = std::min(a_pr, -tau * sl / ds[i]); if (dzL[i] < 0) a_du = std::min(a_du, -tau * zL[i] / dzL[i]); dbar -= mu / sl * ds[i]; }
        if (hasU[i]) { double su = u[i] - s[i]; dzU[i] = mu / su - zU[i] + zU[i] / su * ds[i]; if (ds[i] > 0) a_pr = std::min(a_pr, tau * su / ds[i]); if (dzU[i] < 0) a_du = std::min(a_du, -tau * zU[i] / dzU[i]); dbar += mu / su * ds[i]; }
        sSds += Sigma[i] * ds[i] * ds[i];
      }
      double cn = 0; for (int i = 0; i < m; ++i) cn += std::fabs(r[i]);
      double gdx = 0; for (int j = 0; j < n; ++j) gdx += g[j] * dx[j];
      // dx^T (H + dw Dw) dx without a mat-vec, from the two block rows of the KKT system just solved:
      //   (H + dw Dw) dx + J^T dlam = -dualx ,   J dx - D dlam = rhs_row
      double dHd = 0;
      for (int j = 0; j < n; ++j) dHd -= dx[j] * dualx[j];
      for (int i = 0; i < m; ++i) dHd -= (rhs[pos_row[i]] + D[i] * dlam[i]) * dlam[i];
      dHd += sSds;
      double dphi_bar = gdx + dbar;
      // penalty parameter of the merit function, recomputed for every step (not monotone): a value that was needed once -- typically
      // a quotient by a constraint violation at noise level -- otherwise stays for the rest of the stage and every later step is then
      // judged by second-order changes of a violation of 1e-7 times nu = 1e3 (the stage-3 stragglers of round 2)
      {
        const double nut = cn >= 1e-6 ? (dphi_bar + 0.5 * std::max(dHd, 0.0)) / ((1 - 0.1) * cn) : 0.0;
        nu = std::max(1.0, nut * 1.1 + 1e-8);
      }
      double Dphi = dphi_bar - nu * cn;
      double phi0 = f + barrier(s, mu) + nu * cn;
      alpha = a_pr; ok = false; nls = 0; used_soc = false;
      const double cn_floor = 1e-12;
      if (opt.filter) {
        // ---- filter line search: theta = |r|_1, phi = barrier objective, d phi = dphi_bar
        const double g_th = 1e-5, g_ph = 1e-8, dl = 1.0, s_th = 1.1, s_ph = 2.3, eta = 1e-8, g_al = 0.05;
        const double theta0 = cn, phib0 = f + barrier(s, mu);
        if (filt_mu != mu) { filt.clear(); filt_mu = mu; if (theta_max == 0) { theta_max = 1e4 * std::max(1.0, theta0); theta_min = 1e-4 * std::max(1.0, theta0); } }
        double alpha_min = g_th;
        if (dphi_bar < 0) {
          alpha_min = std::min(g_th, g_ph * theta0 / (-dphi_bar));
          if (theta0 <= theta_min) alpha_min = std::min(alpha_min, dl * std::pow(theta0, s_th) / std::pow(-dphi_bar, s_ph));
        }
        alpha_min *= g_al;
        auto acceptable = [&](double th, double ph) {
          if (th > theta_max) return false;
          for (auto& e : filt) if (!(th < (1 - g_th) * e.first || ph < e.second - g_ph * e.first)) return false;
          return true;
        };
        bool first = true, htype = false;
        while (alpha >= alpha_min && nls < 40) {
          for (int j = 0; j < n; ++j) xt[j] = x[j] + alpha * dx[j];
          for (int i = 0; i < m; ++i) st[i] = s[i] + alpha * ds[i];
          double ft = 0;
          P.eval(xt.data(), &ft, nullptr, ct.data(), nullptr, nullptr);
          ft *= sf; for (int i = 0; i < m; ++i) ct[i] *= sc[i];
          resid(ct, st, rt);
          double tht = 0; for (int i = 0; i < m; ++i) tht += std::fabs(rt[i]);
          const double pht = ft + barrier(st, mu);
          auto test = [&](double th, double ph) {
            if (!std::isfinite(th) || !std::isfinite(ph) || !acceptable(th, ph)) return 0;
            const bool fty = dphi_bar < 0 && alpha * std::pow(-dphi_bar, s_ph) > dl * std::pow(theta0, s_th) && theta0 <= theta_min;
            if (fty) return ph <= phib0 + eta * alpha * dphi_bar ? 1 : 0;
            return (th <= (1 - g_th) * theta0 || ph <= phib0 - g_ph * theta0) ? 2 : 0;
          };
          int acc = test(tht, pht);
          if (acc) { ok = true; htype = acc == 2; break; }
          if (first && opt.use_soc && tht >= theta0 && tht > cn_floor) {      // second-order correction (one step; IPOPT allows four)
            for (int j = 0; j < n; ++j) rhs2[pos_var[j]] = 0.0;
            for (int i = 0; i < m; ++i) rhs2[pos_row[i]] = -rt[i];
            kkt_solve(rhs2.data(), sol2.data());
            bool inside = true;
            for (int j = 0; j < n; ++j) xs[j] = xt[j] + sol2[pos_var[j]];
            for (int i = 0; i < m; ++i) {
              ss2[i] = st[i];
              if (eq[i]) continue;
              ss2[i] = st[i] + sol2[pos_row[i]] / Sigma[i];
              if (hasL[i] && ss2[i] - l[i] < (1 - 1e-8) * (1 - tau) * (s[i] - l[i])) inside = false;
              if (hasU[i] && u[i] - ss2[i] < (1 - 1e-8) * (1 - tau) * (u[i] - s[i])) inside = false;
            }
            if (inside) {
              double fs = 0;
              P.eval(xs.data(), &fs, nullptr, ct.data(), nullptr, nullptr);
              fs *= sf; for (int i = 0; i < m; ++i) ct[i] *= sc[i];
              resid(ct, ss2, rt);
              double ths = 0; for (int i = 0; i < m; ++i) ths += std::fabs(rt[i]);
              acc = test(ths, fs + barrier(ss2, mu));
              if (acc) { ok = true; used_soc = true; htype = acc == 2; break; }
            }
          }
          first = false; alpha *= 0.5; ++nls;
        }
        if (ok && htype) filt.emplace_back((1 - g_th) * theta0, phib0 - g_ph * theta0);
        if (ok) break;
        dw *= 10.0;          // (in place of the restoration phase)
        if (dw > opt.delta_w_max) break;
        continue;
      }
      while (nls <= opt.max_backtrack) {
        for (int j = 0; j < n; ++j) xt[j] = x[j] + alpha * dx[j];
        for (int i = 0; i < m; ++i) st[i] = s[i] + alpha * ds[i];
        double ft = 0;
        P.eval(xt.data(), &ft, nullptr, ct.data(), nullptr, nullptr);
        ft *= sf; for (int i = 0; i < m; ++i) ct[i] *= sc[i];
        resid(ct, st, rt);
        double cnt = 0; for (int i = 0; i < m; ++i) cnt += std::fabs(rt[i]);
        double phit = ft + barrier(st, mu) + nu * cnt;
        if (opt.verbose) std::printf("   ls a=%.3e f %.6e->%.6e bar %.6e->%.6e cn %.6e->%.6e Dphi %.3e gdx %.3e dbar %.3e dHd %.3e a_pr %.3e\n", alpha, f, ft, barrier(s, mu), barrier(st, mu), cn, cnt, Dphi, gdx, dbar, dHd, a_pr);
        if (phit <= phi0 + 1e-4 * alpha * Dphi + 1e-12 * std::fabs(phi0)) { ok = true; ratio_num = phi0 - phit; ratio_den = -(alpha * Dphi + 0.5 * alpha * alpha * std::fmax(dHd, 0.0)); break; }
        if (nls == 0 && opt.use_soc && cnt > cn_floor) {
          // second-order correction (IPOPT sec. 2.4): same factorisation, rhs = constraint
          // residual at the trial point; avoids the Maratos effect of the l1 merit function.
          for (int j = 0; j < n; ++j) rhs2[pos_var[j]] = 0.0;
          for (int i = 0; i < m; ++i) rhs2[pos_row[i]] = -rt[i];
          kkt_solve(rhs2.data(), sol2.data());
          bool inside = true;
          for (int j = 0; j < n; ++j) xs[j] = xt[j] + sol2[pos_var[j]];
          for (int i = 0; i < m; ++i) {
            ss2[i] = st[i];
            if (eq[i]) continue;
            ss2[i] = st[i] + sol2[pos_row[i]] / Sigma[i];
            // (1 - 1e-8): see the same test in chd_kernels.hpp (a slack that limited the step sits exactly on this boundary)
            if (hasL[i] && ss2[i] - l[i] < (1 - 1e-8) * (1 - tau) * (s[i] - l[i])) inside = false;
            if (hasU[i] && u[i] - ss2[i] < (1 - 1e-8) * (1 - tau) * (u[i] - s[i])) inside = false;
          }
          if (inside) {
            double fs = 0;
            P.eval(xs.data(), &fs, nullptr, ct.data(), nullptr, nullptr);
            fs *= sf; for (int i = 0; i < m; ++i) ct[i] *= sc[i];
            resid(ct, ss2, rt);
            double cns = 0; for (int i = 0; i < m; ++i) cns += std::fabs(rt[i]);
            double phis = fs + barrier(ss2, mu) + nu * cns;
            if (phis <= phi0 + 1e-4 * alpha * Dphi + 1e-12 * std::fabs(phi0)) { ok = true; used_soc = true; ratio_num = phi0 - phis; ratio_den = -(alpha * Dphi + 0.5 * alpha * alpha * std::fmax(dHd, 0.0)); break; }
          }
        }
        alpha *= 0.5; ++nls;
      }
      if (ok) break;
      if (second_model()) continue;
      dw *= 10.0;
      if (dw > opt.delta_w_max) break;
    }
    if (!ok) { status = -2; P.set_x(x.data()); break; }
    // the damping follows the exact model: halved after a clean first-model step, raised by half when the iteration had to fall back to the second model
    // (chd_kernels.hpp solve_stage has the same rule and the reason)
    // ... and, with chd_config.damping_rule = 1 (ratio_low = 0.25; off by default), raised like after a backtrack when the accepted step delivered less than that fraction of the
    // reduction the quadratic model of the merit function promised (the filter mode leaves both sides at zero: no effect there)
    const bool poor_ratio = opt.ratio_low > 0.0 && !(ratio_num >= opt.ratio_low * ratio_den);
    if (nls >= 1 || poor_ratio) dw *= 4.0;
    else if (attempt == 0) dw = std::max(opt.delta_w_min, dw / 2.0);
    else if (second_used) dw *= opt.dw_grow_second;
    last_alpha = alpha; last_nls = nls; last_att = attempt; last_soc = used_soc;
    if (opt.lbfgs) lb_xold = x;
    if (used_soc) { for (int j = 0; j < n; ++j) x[j] = xs[j]; } else { for (int j = 0; j < n; ++j) x[j] += alpha * dx[j]; }
    for (int i = 0; i < m; ++i) {
      if (used_soc) s[i] = ss2[i]; else s[i] += alpha * ds[i];
      lam[i] += alpha * dlam[i];
      zL[i] += a_du * dzL[i]; zU[i] += a_du * dzU[i];
      const double ks = 1e10;
      if (hasL[i]) { double sl = s[i] - l[i]; zL[i] = std::min(std::max(zL[i], mu / (ks * sl)), ks * mu / sl); }
      if (hasU[i]) { double su = u[i] - s[i]; zU[i] = std::min(std::max(zU[i], mu / (ks * su)), ks * mu / su); }
    }
    // multipliers of the unscaled rows for the exact duration block of the Lagrangian Hessian (nlp_model.hpp)
    for (int i = 0; i < m; ++i) lraw[i] = lam[i] * sc[i] / sf;
    if (opt.lbfgs) { lb_Jold = J; lb_gold = g; }          // (scaled values of the point just left; lb_xold was taken before the step)
    P.eval(x.data(), &fraw, graw.data(), craw.data(), J.data(), opt.lbfgs ? nullptr : H.data(), opt.lbfgs ? nullptr : lraw.data());
    apply_scaling(true);
    f = sf * fraw;
    if (opt.lbfgs) {
      // y = grad_x L(x+, lam+) - grad_x L(x, lam+)   (IPOPT's limited-memory update uses the new multipliers for both)
      std::vector<double> sv(n), yv(n);
      for (int j = 0; j < n; ++j) { sv[j] = x[j] - lb_xold[j]; yv[j] = g[j] - lb_gold[j]; }
      for (int i = 0; i < m; ++i) { const double li = lam[i]; if (li == 0.0) continue; const double* Jn = &J[(size_t)i * n]; const double* Jo = &lb_Jold[(size_t)i * n]; for (int j = 0; j < n; ++j) yv[j] += (Jn[j] - Jo[j]) * li; }
      double sy = 0, ss = 0, yy = 0;
      for (int j = 0; j < n; ++j) { sy += sv[j] * yv[j]; ss += sv[j] * sv[j]; yy += yv[j] * yv[j]; }
      if (sy > 1.4901161193847656e-08 * std::sqrt(ss) * std::sqrt(yy) && ss > 0) {      // skip the update unless s^T y > sqrt(eps) |s| |y|
        lb_S.push_back(sv); lb_Y.push_back(yv);
        if ((int)lb_S.size() > opt.lbfgs_history) { lb_S.erase(lb_S.begin()); lb_Y.erase(lb_Y.begin()); }
        lb_sigma = std::min(1e8, std::max(1e-8, sy / ss));
      }
    }
  }
  P.set_x(x.data());
  res.status = status; res.iters = it; res.kkt_error = errors(0.0); res.objective = f / sf; res.mu = mu;
  double cv = 0;
  for (int i = 0; i < m; ++i) { double v = craw[i]; cv = std::max(cv, std::max(P.cl[i] - v, v - P.cu[i])); }
  res.constr_viol = std::max(cv, 0.0);
  return res;
}

}  // namespace orc

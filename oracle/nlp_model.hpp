// =============================================================================
// ORACLE (test infrastructure only).  CPU, fp64, dependency-free restatement of
// the reference NLP that `towr_phys_optim/phys_optim` builds (TOWR NlpFormulation
// + the in-tree custom variables / constraints / costs).
//
// Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use
// this code, and only as the checker.  The shipped product path
// (contact-human-dynamics_amd/) never includes, links or calls it.
//
// PARITY UNPINNED: the reference has no golden vectors and its own binary
// cannot be built here (TOWR fork, ifopt, IPOPT/MA57, Eigen, gflags absent), so
// this restatement is validated by finite-difference Jacobian checks
// (tests/test_oracle.py), physical invariants, and for its solver by an
// independent SciPy KKT check + SLSQP solve (tests/test_oracle_second_opinion.py),
// not against reference outputs.
//
// Every function cites the reference file:line it follows.  Pieces that live
// in the absent TOWR fork / ifopt restate the published upstream algorithm
// (ethz-adrl/towr v1.4, ifopt 2.0.x) — marked [UPSTREAM].
// =============================================================================
#pragma once
#include <cmath>
#include <cstdio>
#include <cstring>
#include <vector>
#include <array>
#include <string>
#include <algorithm>
#include <numeric>
#include <stdexcept>

namespace orc {

// ----------------------------------------------------------------------------
// Tiny forward-mode AD scalar (N partials).  Used for the Euler-angle terms so
// the oracle's Jacobians are exact derivatives obtained independently of the
// hand-derived formulas used by the HIP kernels.
// ----------------------------------------------------------------------------
template <int N>
struct Dual {
  double v;
  double d[N];
  Dual() : v(0) { for (int i = 0; i < N; ++i) d[i] = 0; }
  Dual(double c) : v(c) { for (int i = 0; i < N; ++i) d[i] = 0; }
  static Dual var(double c, int k) { Dual r(c); r.d[k] = 1.0; return r; }
};
template <int N> inline Dual<N> operator+(const Dual<N>& a, const Dual<N>& b) { Dual<N> r; r.v = a.v + b.v; for (int i = 0; i < N; ++i) r.d[i] = a.d[i] + b.d[i]; return r; }
template <int N> inline Dual<N> operator-(const Dual<N>& a, const Dual<N>& b) { Dual<N> r; r.v = a.v - b.v; for (int i = 0; i < N; ++i) r.d[i] = a.d[i] - b.d[i]; return r; }
template <int N> inline Dual<N> operator-(const Dual<N>& a) { Dual<N> r; r.v = -a.v; for (int i = 0; i < N; ++i) r.d[i] = -a.d[i]; return r; }
template <int N> inline Dual<N> operator*(const Dual<N>& a, const Dual<N>& b) { Dual<N> r; r.v = a.v * b.v; for (int i = 0; i < N; ++i) r.d[i] = a.d[i] * b.v + a.v * b.d[i]; return r; }
template <int N> inline Dual<N> operator*(double a, const Dual<N>& b) { Dual<N> r; r.v = a * b.v; for (int i = 0; i < N; ++i) r.d[i] = a * b.d[i]; return r; }
template <int N> inline Dual<N> operator*(const Dual<N>& b, double a) { return a * b; }
template <int N> inline Dual<N> sin(const Dual<N>& a) { Dual<N> r; r.v = std::sin(a.v); double c = std::cos(a.v); for (int i = 0; i < N; ++i) r.d[i] = c * a.d[i]; return r; }
template <int N> inline Dual<N> cos(const Dual<N>& a) { Dual<N> r; r.v = std::cos(a.v); double s = -std::sin(a.v); for (int i = 0; i < N; ++i) r.d[i] = s * a.d[i]; return r; }

// ----------------------------------------------------------------------------
// Second-order forward-mode AD scalar (value, gradient, Hessian over N inputs).
// Used for the exact constraint curvature lam . grad^2 c of the Euler-angle terms
// (angular dynamics rows, leg-length rows): the kernel derives the same second
// derivatives by hand, the oracle gets them by differentiating the very function
// that produces the row value.
// ----------------------------------------------------------------------------
template <int N>
struct Jet2 {
  double v;
  double g[N];
  double h[N][N];
  Jet2() : v(0) { for (int i = 0; i < N; ++i) { g[i] = 0; for (int j = 0; j < N; ++j) h[i][j] = 0; } }
  Jet2(double c) : v(c) { for (int i = 0; i < N; ++i) { g[i] = 0; for (int j = 0; j < N; ++j) h[i][j] = 0; } }
  static Jet2 var(double c, int k) { Jet2 r(c); r.g[k] = 1.0; return r; }
};
template <int N> inline Jet2<N> operator+(const Jet2<N>& a, const Jet2<N>& b) { Jet2<N> r; r.v = a.v + b.v; for (int i = 0; i < N; ++i) { r.g[i] = a.g[i] + b.g[i]; for (int j = 0; j < N; ++j) r.h[i][j] = a.h[i][j] + b.h[i][j]; } return r; }
template <int N> inline Jet2<N> operator-(const Jet2<N>& a, const Jet2<N>& b) { Jet2<N> r; r.v = a.v - b.v; for (int i = 0; i < N; ++i) { r.g[i] = a.g[i] - b.g[i]; for (int j = 0; j < N; ++j) r.h[i][j] = a.h[i][j] - b.h[i][j]; } return r; }
template <int N> inline Jet2<N> operator-(const Jet2<N>& a) { Jet2<N> r; r.v = -a.v; for (int i = 0; i < N; ++i) { r.g[i] = -a.g[i]; for (int j = 0; j < N; ++j) r.h[i][j] = -a.h[i][j]; } return r; }
template <int N> inline Jet2<N> operator*(const Jet2<N>& a, const Jet2<N>& b) {
  Jet2<N> r; r.v = a.v * b.v;
  for (int i = 0; i < N; ++i) { r.g[i] = a.g[i] * b.v + a.v * b.g[i]; for (int j = 0; j < N; ++j) r.h[i][j] = a.h[i][j] * b.v + a.g[i] * b.g[j] + a.g[j] * b.g[i] + a.v * b.h[i][j]; }
  return r;
}
template <int N> inline Jet2<N> operator*(double a, const Jet2<N>& b) { Jet2<N> r; r.v = a * b.v; for (int i = 0; i < N; ++i) { r.g[i] = a * b.g[i]; for (int j = 0; j < N; ++j) r.h[i][j] = a * b.h[i][j]; } return r; }
template <int N> inline Jet2<N> operator*(const Jet2<N>& b, double a) { return a * b; }
template <int N> inline Jet2<N> sin(const Jet2<N>& a) { Jet2<N> r; const double s = std::sin(a.v), c = std::cos(a.v); r.v = s; for (int i = 0; i < N; ++i) { r.g[i] = c * a.g[i]; for (int j = 0; j < N; ++j) r.h[i][j] = c * a.h[i][j] - s * a.g[i] * a.g[j]; } return r; }
template <int N> inline Jet2<N> cos(const Jet2<N>& a) { Jet2<N> r; const double s = std::sin(a.v), c = std::cos(a.v); r.v = c; for (int i = 0; i < N; ++i) { r.g[i] = -s * a.g[i]; for (int j = 0; j < N; ++j) r.h[i][j] = -s * a.h[i][j] - c * a.g[i] * a.g[j]; } return r; }

// ----------------------------------------------------------------------------
// Inputs: exactly the contents of the four phys_optim_in_<char>/*.txt files
// (reader: phys_optim.cpp:155-267).
// ----------------------------------------------------------------------------
struct SeqInput {
  int F = 0;
  double dt = 0;
  std::vector<double> hip_l, hip_r;       // F*3   skel_info.txt   (phys_optim.cpp:176-177)
  double leg_len = 0, heel_len = 0, heel_dist = 0, mass = 0;   // :179-182
  std::vector<double> inertia;            // F*6 Ixx Iyy Izz Ixy Ixz Iyz   (:183-187)
  std::vector<double> com, euler;         // F*3 each, motion_info.txt (:199-200)
  std::vector<double> ltoe, lheel, rtoe, rheel;   // F*3 each, file order (:201-204)
  double normal[3] = {0, 0, 1}, point[3] = {0, 0, 0};   // terrain_info.txt (:216-221)
  // contact_info.txt, file order: L-toe, L-heel, R-toe, R-heel (:236-264)
  int start_contact[4] = {0, 0, 0, 0};
  std::vector<double> durations[4];
};

// CLI flags of phys_optim (phys_optim.cpp:27-31) + solver options (:567-578).
struct Config {
  double w_com_lin = 0.4, w_com_ang = 1.7, w_ee = 0.3, w_smooth = 0.1, w_dur = 0.1;
  int max_iter[6] = {7000, 7000, 7000, 2500, 2000, 7000};  // stages 1.1,1.2,2.1,2.2,3,4
  double tol = 1e-3;   // the reference's IPOPT "tol" (success criterion)
};

enum Dx { kPos = 0, kVel = 1, kAcc = 2 };

// ----------------------------------------------------------------------------
// [UPSTREAM] Spline::GetSegmentID (towr/variables/spline.cc): first i with
// sum_{k<=i} T_k >= t - 1e-10; a junction time belongs to the earlier
// polynomial.  Upstream only asserts on out-of-range t (UB in Release); the
// oracle clamps to the last segment (SURVEY §7 "Release-mode UB").
// ----------------------------------------------------------------------------
inline int segment_id(double t_global, const std::vector<double>& durations) {
  const double eps = 1e-10;
  double t = 0;
  int n = (int)durations.size();
  for (int i = 0; i < n; ++i) {
    t += durations[i];
    if (t >= t_global - eps) return i;
  }
  return n - 1;
}

struct PolyInfo {  // [UPSTREAM] NodesVariablesPhaseBased::PolyInfo; built at
  int phase;       // nodes_variables_dynamic_phase_based.cpp:10-34
  int k_in_phase;
  int n_in_phase;
  bool is_const;
};

struct PointEval {
  int poly;
  double tl, T;
  double p[3], v[3], a[3];
  // partials of {pos, vel, acc} wrt (p0, v0, p1, v1) of the active polynomial,
  // identical for the three dimensions ([UPSTREAM] CubicHermitePolynomial::
  // GetDerivativeOf{Pos,Vel,Acc}Wrt{Start,End}Node).
  double w[3][4];
};

// One spline = one ifopt variable set of Hermite nodes + its polynomial
// durations ([UPSTREAM] NodeSpline / PhaseSpline over NodesVariables).
struct Spline {
  bool phase_based = false;
  int ee = -1;
  int n_nodes = 0;
  std::vector<double> node;       // n_nodes * 6 : [node][deriv(2)][dim(3)]
  std::vector<int> var_of;        // n_nodes * 6 -> local optimisation index, -1 = pinned/parameter
  int n_var = 0;
  int var_off = 0;                // offset of this set inside x
  std::vector<double> poly_dur;
  std::vector<PolyInfo> pinfo;    // phase-based only

  double& nv(int n, int deriv, int dim) { return node[(n * 2 + deriv) * 3 + dim]; }
  double nv(int n, int deriv, int dim) const { return node[(n * 2 + deriv) * 3 + dim]; }
  int vi(int n, int deriv, int dim) const { return var_of[(n * 2 + deriv) * 3 + dim]; }
  int n_polys() const { return (int)poly_dur.size(); }
  double total_time() const { return std::accumulate(poly_dur.begin(), poly_dur.end(), 0.0); }

  // [UPSTREAM] NodesVariablesPhaseBased::IsConstantNode: a node is constant if
  // either adjacent polynomial belongs to a constant phase.
  bool is_const_node(int n) const {
    bool c = false;
    if (n > 0 && pinfo[n - 1].is_const) c = true;
    if (n < n_polys() && pinfo[n].is_const) c = true;
    return c;
  }

  void eval_local(int id, double tl, PointEval& e) const {
    const double T = poly_dur[id];
    e.poly = id; e.tl = tl; e.T = T;
    const double t = tl, t2 = t * t, t3 = t2 * t, T2 = T * T, T3 = T2 * T;
    // position partials
    e.w[0][0] = (2 * t3) / T3 - (3 * t2) / T2 + 1;
    e.w[0][1] = t - (2 * t2) / T + t3 / T2;
    e.w[0][2] = (3 * t2) / T2 - (2 * t3) / T3;
    e.w[0][3] = t3 / T2 - t2 / T;
    // velocity partials
    e.w[1][0] = (6 * t2) / T3 - (6 * t) / T2;
    e.w[1][1] = (3 * t2) / T2 - (4 * t) / T + 1;
    e.w[1][2] = (6 * t) / T2 - (6 * t2) / T3;
    e.w[1][3] = (3 * t2) / T2 - (2 * t) / T;
    // acceleration partials
    e.w[2][0] = (12 * t) / T3 - 6 / T2;
    e.w[2][1] = (6 * t) / T2 - 4 / T;
    e.w[2][2] = 6 / T2 - (12 * t) / T3;
    e.w[2][3] = (6 * t) / T2 - 2 / T;
    for (int d = 0; d < 3; ++d) {
      // [UPSTREAM] CubicHermitePolynomial::UpdateCoeff
      const double p0 = nv(id, 0, d), v0 = nv(id, 1, d), p1 = nv(id + 1, 0, d), v1 = nv(id + 1, 1, d);
      const double A = p0, B = v0;
      const double C = -(3 * (p0 - p1) + T * (2 * v0 + v1)) / T2;
      const double D = (2 * (p0 - p1) + T * (v0 + v1)) / T3;
      e.p[d] = A + B * t + C * t2 + D * t3;
      e.v[d] = B + 2 * C * t + 3 * D * t2;
      e.a[d] = 2 * C + 6 * D * t;
    }
  }
  // [UPSTREAM] Spline::GetLocalTime + GetPoint(t_global)
  void eval(double tg, PointEval& e) const {
    int id = segment_id(tg, poly_dur);
    double tl = tg;
    for (int i = 0; i < id; ++i) tl -= poly_dur[i];
    eval_local(id, tl, e);
  }
  // [UPSTREAM] CubicHermitePolynomial::GetDerivativeOfPosWrtDuration
  void dpos_dTpoly(const PointEval& e, double out[3]) const {
    const int id = e.poly;
    const double t = e.tl, t2 = t * t, t3 = t2 * t, T = e.T, T2 = T * T, T3 = T2 * T, T4 = T3 * T;
    for (int d = 0; d < 3; ++d) {
      const double x0 = nv(id, 0, d), v0 = nv(id, 1, d), x1 = nv(id + 1, 0, d), v1 = nv(id + 1, 1, d);
      out[d] = (t3 * (v0 + v1)) / T3 - (t2 * (2 * v0 + v1)) / T2 -
               (3 * t3 * (2 * x0 - 2 * x1 + T * v0 + T * v1)) / T4 +
               (2 * t2 * (3 * x0 - 3 * x1 + 2 * T * v0 + T * v1)) / T3;
    }
  }
};

// ----------------------------------------------------------------------------
// Stage descriptions (phys_optim.cpp:544-749).
// ----------------------------------------------------------------------------
enum RowFamily {
  FAM_BASEACC = 1, FAM_TERRAIN = 2, FAM_ROM = 4, FAM_HEELDIST = 8,
  FAM_DYNAMIC = 16, FAM_FORCE = 32, FAM_HEIGHT = 64, FAM_TOTALTIME = 128
};
struct StageDef {
  int families;
  double w_data[3];     // base-lin, base-ang, ee      (AddDataCosts)
  double w_vel[3];      // position-difference smoothing (AddVelocitySmoothCosts)
  double w_acc[3];      // velocity-difference smoothing (AddAccelSmoothCosts); <0 = absent
  double w_dur;         // DurationCost weight; <0 = absent
  bool opt_durations;
};

inline StageDef stage_def(int stage, const Config& c) {
  StageDef s{};
  const int kin = FAM_TERRAIN | FAM_ROM;          // Parameters::AddLegConstraints  parameters.cpp:78-82
  const int dyn = FAM_DYNAMIC | FAM_FORCE;        // Parameters::AddDynamicsConstraints :95-98
  switch (stage) {
    case 0:  // STAGE 1.1  phys_optim.cpp:544-581 (default Parameters ctor: BaseAcc only, parameters.cpp:62)
      s.families = FAM_BASEACC;
      s.w_data[0] = s.w_data[1] = s.w_data[2] = 1.0;          // :560-562
      s.w_vel[0] = s.w_vel[1] = s.w_vel[2] = 0.1;             // :564
      s.w_acc[0] = s.w_acc[1] = s.w_acc[2] = -1; s.w_dur = -1; s.opt_durations = false; break;
    case 1:  // STAGE 1.2  :591-599  (constraint sets APPENDED to the stage-1.1 problem)
      s.families = FAM_BASEACC | kin | FAM_HEELDIST;
      s.w_data[0] = s.w_data[1] = s.w_data[2] = 1.0;
      s.w_vel[0] = s.w_vel[1] = s.w_vel[2] = 0.1;
      s.w_acc[0] = s.w_acc[1] = s.w_acc[2] = -1; s.w_dur = -1; s.opt_durations = false; break;
    case 2:  // STAGE 2.1  :609-643
    case 3:  // STAGE 2.2  :648-656 (Height appended)
    case 5:  // STAGE 4    :714-749
      s.families = FAM_BASEACC | kin | dyn | FAM_HEELDIST | (stage != 2 ? FAM_HEIGHT : 0);
      s.w_data[0] = c.w_com_lin; s.w_data[1] = c.w_com_ang; s.w_data[2] = c.w_ee;     // :627-633
      s.w_vel[0] = 0.001; s.w_vel[1] = 0.001; s.w_vel[2] = c.w_smooth;                // :635
      s.w_acc[0] = s.w_acc[1] = s.w_acc[2] = 0.0001;                                  // :637
      s.w_dur = -1; s.opt_durations = false; break;
    case 4:  // STAGE 3    :666-711
      s.families = FAM_BASEACC | kin | dyn | FAM_HEIGHT | FAM_HEELDIST | FAM_TOTALTIME;
      s.w_data[0] = c.w_com_lin; s.w_data[1] = c.w_com_ang; s.w_data[2] = c.w_ee;     // :688-690
      s.w_vel[0] = 0.001; s.w_vel[1] = 0.001; s.w_vel[2] = c.w_smooth;                // :692
      s.w_acc[0] = s.w_acc[1] = s.w_acc[2] = -1;                                      // :693 (unsupported)
      s.w_dur = c.w_dur; s.opt_durations = true; break;                               // :696-703
    default: throw std::runtime_error("bad stage");
  }
  return s;
}

static const double kInf = 1e20;      // ifopt's "infinity" for bounds [UPSTREAM]
static const double kGravity = 9.80665;   // [UPSTREAM] DynamicModel::g_
static const double kFriction = 0.5;      // [UPSTREAM] HeightMap::friction_coeff_

// ----------------------------------------------------------------------------
// The NLP for one sequence.
// ----------------------------------------------------------------------------
class Problem {
 public:
  SeqInput in;
  Config cfg;
  double T = 0;                       // total time = sum of L-toe durations (phys_optim.cpp:420-423)
  // splines: 0 base-lin, 1 base-ang, 2..5 ee-motion (NLP ee order), 6..9 ee-force
  Spline sp[10];
  // NLP ee order 0 L-toe, 1 R-toe, 2 L-heel, 3 R-heel (phys_optim.cpp:505-513)
  std::vector<double> phase_dur[4], phase_dur0[4];
  bool ee_start_contact[4];
  const std::vector<double>* ee_data[4];
  double gdir[3];                     // unit gravity direction = -normal/|normal|  (:437, humanoid_rigid_body_dynamics.cpp:208-211)
  double hx = 0, hy = 0;              // plane height derivatives (ground_plane.cpp:29-41)
  double nrm_n[3], nrm_t1[3], nrm_t2[3];   // [UPSTREAM] HeightMap::GetNormalizedBasis for a plane

  // STUDY SWITCHES (tests/tools only; 0 = the shipped algorithm, which the HIP kernel implements).  profiles/r04_curvature_study.md has the measurements.
  //   bit 0 / 1 / 2: ADD the exact node-node curvature lam grad^2 c of the dynamics rows' torque term / of their angular term / of the leg-length rows
  //                  (all three slow the solve down on every family of sequences tried, the kinematic optimisation's clips the most)
  //   bit 3: keep those blocks in the second model of an iteration
  //   bit 4: second model = heel-distance curvature with max(lam, 0), capped at clip_cap (rounds 2-3 shipped this without a cap)
  //   bit 5: no exact node x duration block (round 3's model of the duration stage)
  int study_mask = 0;
  double clip_cap = 1e300;
  // structure mode of eval(): a Jacobian entry of a spline sample is marked 1 for EVERY coefficient of the active polynomial in a dimension with a non-zero
  // factor, whatever its Hermite weight (a sample on a node has weight exactly 0 on the polynomial's other node).  The solver takes the KKT ordering from it,
  // so that it is the ordering the kernel's table builder derives from the structure (chd_model.hpp: at_time / poly_vars).
  bool pattern_mode = false;
  // --- current stage ---
  int stage = -1;
  StageDef sd{};
  int n = 0, n_nodesvars = 0, n_dur = 0;
  int dur_off[4] = {0, 0, 0, 0};
  int m = 0;
  std::vector<double> cl, cu;
  std::vector<int> row_family;

  // sample-time tables [UPSTREAM] TimeDiscretizationConstraint ctor
  std::vector<double> t_dyn, t_rom, t_height;
  std::vector<double> t_height_kept[4];   // per ee, see height_row_kept()

  explicit Problem(const SeqInput& input, const Config& c = Config()) : in(input), cfg(c) { build(); }

  // ------------------------------------------------------------------ build
  static std::vector<double> disc_times(double T, double dt) {
    std::vector<double> ts;
    double t = 0.0;
    ts.push_back(t);
    int nsteps = (int)std::floor(T / dt);
    for (int i = 0; i < nsteps; ++i) { t += dt; ts.push_back(t); }
    ts.push_back(T);
    return ts;
  }

  // phys_optim.cpp:289-312
  static std::vector<int> polys_per_changing_phase(bool start_constant, const std::vector<double>& dur,
                                                   double max_dur, int n_per_change) {
    std::vector<int> out;
    bool is_const = start_constant;
    double per_s = n_per_change / max_dur;
    for (size_t i = 0; i < dur.size(); ++i) {
      if (!is_const) {
        int np = n_per_change;
        if (dur[i] > max_dur) np += (int)std::ceil((dur[i] - max_dur) * per_s);
        out.push_back(np);
      }
      is_const = !is_const;
    }
    return out;
  }

  // nodes_variables_dynamic_phase_based.cpp:10-34
  static std::vector<PolyInfo> build_poly_infos(int phase_count, bool first_const, const std::vector<int>& npoly) {
    std::vector<PolyInfo> v;
    bool c = first_const;
    int k = 0;
    for (int i = 0; i < phase_count; ++i) {
      if (c) v.push_back({i, 0, 1, true});
      else {
        for (int j = 0; j < npoly.at(k); ++j) v.push_back({i, j, npoly.at(k), false});
        ++k;
      }
      c = !c;
    }
    return v;
  }

  // [UPSTREAM] NodesVariablesPhaseBased::ConvertPhaseToPolyDurations
  void update_phase_spline_durations(int ee) {
    for (int which = 0; which < 2; ++which) {
      Spline& s = sp[(which ? 6 : 2) + ee];
      for (size_t i = 0; i < s.pinfo.size(); ++i)
        s.poly_dur[i] = phase_dur[ee][s.pinfo[i].phase] / s.pinfo[i].n_in_phase;
    }
  }

  // [UPSTREAM] NodesVariables::SetByLinearInterpolation: only node values that are
  // optimisation variables are written; a variable shared by two nodes ends up with
  // the value of the LAST node info once IPOPT reads it back via GetValues().
  static void set_by_linear_interpolation(Spline& s, const double a[3], const double b[3], double t_total) {
    double dp[3] = {b[0] - a[0], b[1] - a[1], b[2] - a[2]};
    for (int nidx = 0; nidx < s.n_nodes; ++nidx)
      for (int dim = 0; dim < 3; ++dim) {
        if (s.vi(nidx, 0, dim) >= 0) s.nv(nidx, 0, dim) = a[dim] + nidx / (double)(s.n_nodes - 1) * dp[dim];
        if (s.vi(nidx, 1, dim) >= 0) s.nv(nidx, 1, dim) = dp[dim] / t_total;
      }
  }

  double plane_height(double x, double y) const {   // ground_plane.cpp:18-27
    double z = -in.normal[1] * (y - in.point[1]) - in.normal[0] * (x - in.point[0]);
    z /= in.normal[2];
    z += in.point[2];
    return z;
  }

  void build() {
    const int F = in.F;
    // NLP ee order (phys_optim.cpp:505-513): file slots L-toe(0), R-toe(2), L-heel(1), R-heel(3)
    const int file_slot[4] = {0, 2, 1, 3};
    const std::vector<double>* data_by_file[4] = {&in.ltoe, &in.lheel, &in.rtoe, &in.rheel};
    for (int e = 0; e < 4; ++e) {
      phase_dur[e] = in.durations[file_slot[e]];
      phase_dur0[e] = phase_dur[e];
      ee_start_contact[e] = in.start_contact[file_slot[e]] != 0;
      ee_data[e] = data_by_file[file_slot[e]];
    }
    T = 0;
    for (double d : in.durations[0]) T += d;    // phys_optim.cpp:420-423

    double nn = std::sqrt(in.normal[0] * in.normal[0] + in.normal[1] * in.normal[1] + in.normal[2] * in.normal[2]);
    for (int d = 0; d < 3; ++d) gdir[d] = -in.normal[d] / nn;
    hx = -in.normal[0] / in.normal[2];
    hy = -in.normal[1] / in.normal[2];
    {  // [UPSTREAM] HeightMap::GetNormal/GetTangent1/GetTangent2, normalised
      double nv[3] = {-hx, -hy, 1.0}, t1[3] = {1, 0, hx}, t2[3] = {0, 1, hy};
      auto nz = [](const double* v, double* o) { double l = std::sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]); for (int i = 0; i < 3; ++i) o[i] = v[i] / l; };
      nz(nv, nrm_n); nz(t1, nrm_t1); nz(t2, nrm_t2);
    }

    // ---- base splines: parameters.cpp:109-125 (0.1 s polynomials)
    std::vector<double> base_dur;
    {
      double dtb = 0.1, t_left = T, eps = 1e-10;
      while (t_left > eps) { base_dur.push_back(t_left > dtb ? dtb : t_left); t_left -= dtb; }
    }
    const int vel_avg = 5;   // phys_optim.cpp:442
    double lin0[3], linF[3], vel0[3] = {0, 0, 0}, velF[3] = {0, 0, 0}, ang0[3], angF[3];
    for (int d = 0; d < 3; ++d) {
      lin0[d] = in.com[d]; linF[d] = in.com[(F - 1) * 3 + d];
      ang0[d] = in.euler[d]; angF[d] = in.euler[(F - 1) * 3 + d];
      for (int k = 0; k < vel_avg; ++k) {                         // :446-455, :470-479
        vel0[d] += (in.com[(k + 1) * 3 + d] - in.com[k * 3 + d]) / in.dt;
        velF[d] += (in.com[(F - 1 - k) * 3 + d] - in.com[(F - 2 - k) * 3 + d]) / in.dt;
      }
      vel0[d] /= vel_avg; velF[d] /= vel_avg;
    }
    for (int b = 0; b < 2; ++b) {                                  // nlp_formulation.cpp:106-130
      Spline& s = sp[b];
      s.phase_based = false;
      s.poly_dur = base_dur;
      s.n_nodes = (int)base_dur.size() + 1;
      s.node.assign(s.n_nodes * 6, 0.0);
      s.var_of.assign(s.n_nodes * 6, -1);
      int idx = 0;
      for (int nidx = 0; nidx < s.n_nodes; ++nidx)
        for (int deriv = 0; deriv < 2; ++deriv)
          for (int dim = 0; dim < 3; ++dim) {
            // base-lin start/final velocity are equality-bounded (:120-121) => IPOPT
            // (fixed_variable_treatment=make_parameter) removes them from the NLP.
            bool fixed = (b == 0 && deriv == 1 && (nidx == 0 || nidx == s.n_nodes - 1));
            s.var_of[(nidx * 2 + deriv) * 3 + dim] = fixed ? -1 : idx++;
          }
      s.n_var = idx;
      // SetByLinearInterpolation writes every NodesVariablesAll entry (all are opt. indices there)
      const double* a = b == 0 ? lin0 : ang0;
      const double* e = b == 0 ? linF : angF;
      for (int nidx = 0; nidx < s.n_nodes; ++nidx)
        for (int dim = 0; dim < 3; ++dim) {
          s.nv(nidx, 0, dim) = a[dim] + nidx / (double)(s.n_nodes - 1) * (e[dim] - a[dim]);
          s.nv(nidx, 1, dim) = (e[dim] - a[dim]) / T;
        }
      if (b == 0)
        for (int dim = 0; dim < 3; ++dim) { s.nv(0, 1, dim) = vel0[dim]; s.nv(s.n_nodes - 1, 1, dim) = velF[dim]; }
    }

    // ---- ee motion / force splines
    for (int e = 0; e < 4; ++e) {
      const int n_phase = (int)phase_dur[e].size();
      // phys_optim.cpp:516-534 ; parameters.cpp:51-53 (add_polys_after_dur_=2.0, 6 polys)
      auto np_motion = polys_per_changing_phase(ee_start_contact[e], phase_dur[e], 2.0, 6);
      auto np_force = polys_per_changing_phase(!ee_start_contact[e], phase_dur[e], 2.0, 6);
      {  // NodesVariablesDynamicEEMotion  nodes_variables_dynamic_phase_based.cpp:58-106
        Spline& s = sp[2 + e];
        s.phase_based = true; s.ee = e;
        s.pinfo = build_poly_infos(n_phase, ee_start_contact[e], np_motion);
        s.poly_dur.assign(s.pinfo.size(), 0.0);
        s.n_nodes = (int)s.pinfo.size() + 1;
        s.node.assign(s.n_nodes * 6, 0.0);
        s.var_of.assign(s.n_nodes * 6, -1);
        int idx = 0;
        for (int nidx = 0; nidx < s.n_nodes; ++nidx) {
          if (!s.is_const_node(nidx)) {
            for (int dim = 0; dim < 3; ++dim) {
              s.var_of[(nidx * 2 + 0) * 3 + dim] = idx++;
              s.var_of[(nidx * 2 + 1) * 3 + dim] = idx++;
            }
          } else {
            for (int dim = 0; dim < 3; ++dim) {      // one variable shared by both stance nodes (:94-98)
              s.var_of[(nidx * 2 + 0) * 3 + dim] = idx;
              s.var_of[((nidx + 1) * 2 + 0) * 3 + dim] = idx;
              idx++;
            }
            nidx += 1;
          }
        }
        s.n_var = idx;
      }
      {  // NodesVariablesDynamicEEForce  :108-151
        Spline& s = sp[6 + e];
        s.phase_based = true; s.ee = e;
        s.pinfo = build_poly_infos(n_phase, !ee_start_contact[e], np_force);
        s.poly_dur.assign(s.pinfo.size(), 0.0);
        s.n_nodes = (int)s.pinfo.size() + 1;
        s.node.assign(s.n_nodes * 6, 0.0);
        s.var_of.assign(s.n_nodes * 6, -1);
        int idx = 0;
        for (int nidx = 0; nidx < s.n_nodes; ++nidx) {
          if (!s.is_const_node(nidx)) {
            for (int dim = 0; dim < 3; ++dim) {
              s.var_of[(nidx * 2 + 0) * 3 + dim] = idx++;
              s.var_of[(nidx * 2 + 1) * 3 + dim] = idx++;
            }
          } else {
            nidx += 1;   // swing: both nodes pinned to zero force / zero derivative
          }
        }
        s.n_var = idx;
      }
      update_phase_spline_durations(e);

      // initial guesses: nlp_formulation.cpp:148-156 (motion), :174-181 (force)
      {
        double x = linF[0], y = linF[1];
        double target[3] = {x, y, plane_height(x, y)};
        double start[3] = {(*ee_data[e])[0], (*ee_data[e])[1], (*ee_data[e])[2]};   // phys_optim.cpp:491-503
        set_by_linear_interpolation(sp[2 + e], start, target, T);
        // stance variables: both nodes take the value of the later node (GetValues -> SetVariables round trip)
        Spline& s = sp[2 + e];
        for (int nidx = 0; nidx + 1 < s.n_nodes; ++nidx)
          if (s.pinfo[nidx].is_const)
            for (int dim = 0; dim < 3; ++dim) s.nv(nidx, 0, dim) = s.nv(nidx + 1, 0, dim);
        double fs[3] = {0.0, 0.0, in.mass * kGravity / 4.0};
        set_by_linear_interpolation(sp[6 + e], fs, fs, T);
      }
    }
    int off = 0;
    for (int i = 0; i < 10; ++i) { sp[i].var_off = off; off += sp[i].n_var; }
    n_nodesvars = off;

    t_dyn = disc_times(T, 0.1);      // parameters.cpp:59  dt_constraint_dynamic_
    t_rom = disc_times(T, 0.08);     // :57 dt_constraint_range_of_motion_
    t_height = disc_times(T, 0.1);   // :58 dt_constraint_height_
    set_stage(0);
  }

  // ---------------------------------------------------------------- x <-> nodes
  void get_x(double* x) const {
    for (int i = 0; i < 10; ++i) {
      const Spline& s = sp[i];
      for (int k = 0; k < s.n_nodes * 6; ++k)
        if (s.var_of[k] >= 0) x[s.var_off + s.var_of[k]] = s.node[k];
    }
    if (sd.opt_durations)
      for (int e = 0; e < 4; ++e)
        for (size_t k = 0; k + 1 < phase_dur[e].size(); ++k) x[dur_off[e] + k] = phase_dur[e][k];
  }
  void set_x(const double* x) {
    for (int i = 0; i < 10; ++i) {
      Spline& s = sp[i];
      for (int k = 0; k < s.n_nodes * 6; ++k)
        if (s.var_of[k] >= 0) s.node[k] = x[s.var_off + s.var_of[k]];
    }
    if (sd.opt_durations)
      for (int e = 0; e < 4; ++e) {       // [UPSTREAM] PhaseDurations::SetVariables
        double sum = 0;
        size_t np = phase_dur[e].size();
        for (size_t k = 0; k + 1 < np; ++k) { phase_dur[e][k] = x[dur_off[e] + k]; sum += phase_dur[e][k]; }
        phase_dur[e][np - 1] = T - sum;
        update_phase_spline_durations(e);
      }
  }

  // ---------------------------------------------------------------- stages
  void set_stage(int st) {
    stage = st;
    sd = stage_def(st, cfg);
    n = n_nodesvars;
    n_dur = 0;
    if (sd.opt_durations)
      for (int e = 0; e < 4; ++e) { dur_off[e] = n; n += (int)phase_dur[e].size() - 1; n_dur += (int)phase_dur[e].size() - 1; }
    count_rows();
  }

  // Per-variable descriptors used by the solver to order the KKT system: the time of the
  // node a variable belongs to, whether the variable couples over a long time span
  // (stance positions, phase durations -> "border"), and the damping scale class.
  // kind: 0 base/ee-motion node value, 1 force node value, 2 duration.
  void var_descriptors(std::vector<double>& time, std::vector<char>& border, std::vector<char>& kind) const {
    time.assign(n, 0.0); border.assign(n, 0); kind.assign(n, 0);
    for (int i = 0; i < 10; ++i) {
      const Spline& s = sp[i];
      std::vector<double> tn(s.n_nodes, 0.0);
      for (int k = 1; k < s.n_nodes; ++k) tn[k] = tn[k - 1] + s.poly_dur[k - 1];
      for (int nd = 0; nd < s.n_nodes; ++nd)
        for (int q = 0; q < 6; ++q) {
          int v = s.var_of[nd * 6 + q];
          if (v < 0) continue;
          int g = s.var_off + v;
          bool stance_shared = s.phase_based && i < 6 && s.is_const_node(nd);
          if (stance_shared) { border[g] = 1; if (nd + 1 < s.n_nodes && s.pinfo[std::min(nd, s.n_polys() - 1)].is_const) time[g] = tn[nd]; }
          else time[g] = tn[nd];
          kind[g] = (i >= 6) ? 1 : 0;
        }
    }
    for (int g = n_nodesvars; g < n; ++g) { border[g] = 1; kind[g] = 2; }
    if (sd.opt_durations)
      for (int e = 0; e < 4; ++e) {          // a duration variable: the start time of its phase (orders the border, ipm_solver.hpp)
        double t0 = 0;
        for (size_t k = 0; k + 1 < phase_dur[e].size(); ++k) { time[dur_off[e] + k] = t0; t0 += phase_dur[e][k]; }
      }
  }

  // Constraint-row layout (order is irrelevant to the NLP; chosen once here).
  struct RowBlock { int family; int ee; int row0; int count; };
  std::vector<RowBlock> blocks;

  bool terrain_row_kept(const Spline& s, int node) const {
    // [UPSTREAM] TerrainConstraint skips node 0.  Additionally the two nodes of a stance
    // polynomial share their position variables (nodes_variables_dynamic_phase_based.cpp:94-98)
    // so upstream emits the same equality twice; the oracle keeps the second node only
    // (identical feasible set, avoids a structurally rank-deficient Jacobian).
    if (node == 0) return false;
    if (node < s.n_polys() && s.pinfo[node].is_const) return false;
    return true;
  }

  // HeightConstraint rows (height_constraint.cpp:24-36) whose sample evaluates to a stance
  // (constant) node are identically zero once the TerrainConstraint equality of that stance
  // holds: n.(p-p0) = n_z (z - h(x,y)).  They are always-active degenerate inequalities with
  // an empty interior; the oracle drops them (same feasible set).  The mask is taken with the
  // phase durations in force when the stage is set up.
  bool height_row_kept(int e, double t) const {
    const Spline& s = sp[2 + e];
    int id = segment_id(t, s.poly_dur);
    if (s.pinfo[id].is_const) return false;
    double tl = t;
    for (int i = 0; i < id; ++i) tl -= s.poly_dur[i];
    if (tl >= s.poly_dur[id] - 1e-9 && id + 1 < s.n_polys() && s.pinfo[id + 1].is_const) return false;
    if (tl <= 1e-9 && id > 0 && s.pinfo[id - 1].is_const) return false;
    return true;
  }

  void count_rows() {
    blocks.clear(); cl.clear(); cu.clear(); row_family.clear();
    int r = 0;
    auto push = [&](int fam, int ee, int cnt) { blocks.push_back({fam, ee, r, cnt}); r += cnt; };
    if (sd.families & FAM_BASEACC) {
      push(FAM_BASEACC, 0, 3 * (sp[0].n_polys() - 1));
      push(FAM_BASEACC, 1, 3 * (sp[1].n_polys() - 1));
    }
    if (sd.families & FAM_TERRAIN)
      for (int e = 0; e < 4; ++e) {
        int c = 0;
        for (int nd = 0; nd < sp[2 + e].n_nodes; ++nd) c += terrain_row_kept(sp[2 + e], nd);
        push(FAM_TERRAIN, e, c);
      }
    if (sd.families & FAM_ROM) for (int e = 0; e < 4; ++e) push(FAM_ROM, e, (int)t_rom.size());
    if (sd.families & FAM_HEELDIST) { push(FAM_HEELDIST, 0, (int)t_rom.size()); push(FAM_HEELDIST, 1, (int)t_rom.size()); }
    if (sd.families & FAM_DYNAMIC) push(FAM_DYNAMIC, 0, 6 * (int)t_dyn.size());
    if (sd.families & FAM_FORCE)
      for (int e = 0; e < 4; ++e) {
        int c = 0;
        for (int nd = 0; nd < sp[6 + e].n_nodes; ++nd) c += !sp[6 + e].is_const_node(nd);
        push(FAM_FORCE, e, 5 * c);
      }
    if (sd.families & FAM_HEIGHT)
      for (int e = 0; e < 4; ++e) {
        t_height_kept[e].clear();
        for (double t : t_height) if (height_row_kept(e, t)) t_height_kept[e].push_back(t);
        push(FAM_HEIGHT, e, (int)t_height_kept[e].size());
      }
    if (sd.families & FAM_TOTALTIME)
      for (int e = 0; e < 4; ++e) {
        push(FAM_TOTALTIME, e, 1);
        push(-1, e, (int)phase_dur[e].size() - 1);   // PhaseDurations variable bounds (0,500) as rows
      }
    m = r;
    cl.assign(m, 0.0); cu.assign(m, 0.0); row_family.assign(m, 0);
    for (auto& b : blocks) for (int i = 0; i < b.count; ++i) row_family[b.row0 + i] = b.family;
    // bounds are filled by eval() (they do not depend on x)
  }

  // ------------------------------------------------------ duration Jacobian
  // [UPSTREAM] PhaseSpline::GetJacobianOfPosWrtDurations(t) = PhaseDurations::
  // GetJacobianOfPos(current_phase, dx_dT, xd).  out: 3 x (n_phases-1), row-major.
  void jac_pos_wrt_durations(const Spline& s, double t, const PointEval& e, std::vector<double>& out) const {
    const int ee = s.ee;
    const int nvar = (int)phase_dur[ee].size() - 1;
    out.assign(3 * nvar, 0.0);
    double dxdTpoly[3];
    s.dpos_dTpoly(e, dxdTpoly);
    const PolyInfo& pi = s.pinfo[e.poly];
    double dx_dT[3];
    for (int d = 0; d < 3; ++d) dx_dT[d] = (1.0 / pi.n_in_phase) * (dxdTpoly[d] - pi.k_in_phase * e.v[d]);
    int cur = segment_id(t, phase_dur[ee]);
    bool last = (cur == (int)phase_dur[ee].size() - 1);
    if (!last) for (int d = 0; d < 3; ++d) out[d * nvar + cur] = dx_dT[d];
    for (int ph = 0; ph < cur; ++ph)
      for (int d = 0; d < 3; ++d) {
        out[d * nvar + ph] = -e.v[d];
        if (last) out[d * nvar + ph] -= dx_dT[d];
      }
  }

  // ------------------------------------------------- second-order duration terms
  // Not in the reference (IPOPT runs with an L-BFGS Hessian there, phys_optim.cpp:572): the solver restated
  // in ipm_solver.hpp uses the exact duration-duration block of the Lagrangian Hessian.  For a sample at time t
  // in polynomial e.poly (local time tau, duration Tp) both tau and Tp are affine in the duration variables:
  //   d tau / dT_k = u_k ,  d Tp / dT_k = v_k   (see dur_uv), hence
  //   dp/dT_k = h_tau u_k + h_T v_k ,   d2p/dT_k dT_l = h_tautau u_k u_l + h_tauT (u_k v_l + v_k u_l) + h_TT v_k v_l.
  void dur_uv(const Spline& s, double t, const PointEval& e, int k, double& u, double& v) const {
    const int ee = s.ee;
    const int cur = segment_id(t, phase_dur[ee]);
    const bool last = cur == (int)phase_dur[ee].size() - 1;
    const PolyInfo& pi = s.pinfo[e.poly];
    const double n = pi.n_in_phase, kin = pi.k_in_phase;
    u = 0; v = 0;
    if (last) { u = -1.0 + kin / n; v = -1.0 / n; return; }      // every variable precedes the last phase; T_last = T - sum
    if (k < cur) { u = -1.0; v = 0.0; }
    else if (k == cur) { u = -kin / n; v = 1.0 / n; }
  }
  void hermite_T_derivs(const Spline& s, const PointEval& e, double hT[3], double htT[3], double hTT[3]) const {
    const int id = e.poly;
    const double tau = e.tl, T = e.T;
    for (int d = 0; d < 3; ++d) {
      const double p0 = s.nv(id, 0, d), v0 = s.nv(id, 1, d), p1 = s.nv(id + 1, 0, d), v1 = s.nv(id + 1, 1, d);
      const double dl = p0 - p1, s2 = 2 * v0 + v1, s1 = v0 + v1;
      const double cT = 6 * dl / std::pow(T, 3) + s2 / (T * T), dT = -6 * dl / std::pow(T, 4) - 2 * s1 / std::pow(T, 3);
      const double cTT = -18 * dl / std::pow(T, 4) - 2 * s2 / std::pow(T, 3), dTT = 24 * dl / std::pow(T, 5) + 6 * s1 / std::pow(T, 4);
      hT[d] = cT * tau * tau + dT * tau * tau * tau;
      htT[d] = 2 * cT * tau + 3 * dT * tau * tau;
      hTT[d] = cTT * tau * tau + dTT * tau * tau * tau;
    }
  }
  void d1pos(const Spline& s, double t, const PointEval& e, int k, double out[3]) const {
    double u, v, hT[3], htT[3], hTT[3];
    dur_uv(s, t, e, k, u, v); hermite_T_derivs(s, e, hT, htT, hTT);
    for (int d = 0; d < 3; ++d) out[d] = e.v[d] * u + hT[d] * v;
  }
  void d2pos(const Spline& s, double t, const PointEval& e, int k, int l, double out[3]) const {
    double uk, vk, ul, vl, hT[3], htT[3], hTT[3];
    dur_uv(s, t, e, k, uk, vk); dur_uv(s, t, e, l, ul, vl); hermite_T_derivs(s, e, hT, htT, hTT);
    for (int d = 0; d < 3; ++d) out[d] = e.a[d] * uk * ul + htT[d] * (uk * vl + vk * ul) + hTT[d] * vk * vl;
  }

  // Mixed node x duration derivatives.  p(t) = sum_j w_j(tau, Tp) x_j over the four Hermite coefficients of the active polynomial, and both tau and Tp
  // are affine in the duration variables (dur_uv), so for a duration T_k of class x in {e: earlier phase, c: current phase}
  //   d p / d T_k         = G_x   = h_tau u_x + h_T v_x                       (first derivative: d1pos)
  //   d2 p / d x_j d T_k  = om_x,j = (d w_j / d tau) u_x + (d w_j / d Tp) v_x  (the derivative of the weight itself)
  // Not in the reference (L-BFGS there); with the duration-duration block this makes the duration stage's Hessian of the Lagrangian exact.
  struct DurX { int cur, nvar; bool last; double Ge[3], Gc[3], ome[4], omc[4]; };
  void dur_cross(const Spline& s, double t, const PointEval& e, DurX& dx) const {
    const int ee = s.ee;
    dx.cur = segment_id(t, phase_dur[ee]);
    dx.nvar = (int)phase_dur[ee].size() - 1;
    dx.last = dx.cur == dx.nvar;
    const PolyInfo& pi = s.pinfo[e.poly];
    const double nn = pi.n_in_phase, kin = pi.k_in_phase;
    const double ue = dx.last ? -1.0 + kin / nn : -1.0, ve = dx.last ? -1.0 / nn : 0.0;
    const double uc = -kin / nn, vc = 1.0 / nn;
    const double tau = e.tl, T = e.T, t2 = tau * tau, t3 = t2 * tau, T2 = T * T, T3 = T2 * T, T4 = T3 * T;
    const double wT[4] = {-6 * t3 / T4 + 6 * t2 / T3, 2 * t2 / T2 - 2 * t3 / T3, -6 * t2 / T3 + 6 * t3 / T4, -2 * t3 / T3 + t2 / T2};
    double hT[3], htT[3], hTT[3];
    hermite_T_derivs(s, e, hT, htT, hTT);
    for (int d = 0; d < 3; ++d) { dx.Ge[d] = e.v[d] * ue + hT[d] * ve; dx.Gc[d] = dx.last ? 0.0 : e.v[d] * uc + hT[d] * vc; }
    for (int j = 0; j < 4; ++j) { dx.ome[j] = e.w[kVel][j] * ue + wT[j] * ve; dx.omc[j] = dx.last ? 0.0 : e.w[kVel][j] * uc + wT[j] * vc; }
  }

  // [UPSTREAM] EulerConverter::GetRotationMatrixBaseToWorld (ZYX, kindr cheat-sheet)
  template <class S> static void rot_zyx(const S e[3], S R[3][3]) {
    S x = e[0], y = e[1], z = e[2];
    S cx = cos(x), sx = sin(x), cy = cos(y), sy = sin(y), cz = cos(z), sz = sin(z);
    R[0][0] = cy * cz; R[0][1] = cz * sx * sy - cx * sz; R[0][2] = sx * sz + cx * cz * sy;
    R[1][0] = cy * sz; R[1][1] = cx * cz + sx * sy * sz; R[1][2] = cx * sy * sz - cz * sx;
    R[2][0] = -sy;     R[2][1] = cy * sx;               R[2][2] = cx * cy;
  }

  // Angular part of HumanoidRigidBodyDynamics::GetDynamicViolation
  // (humanoid_rigid_body_dynamics.cpp:89-115) as a function of (euler, euler', euler''):
  //   I_w * omega_dot + omega x (I_w * omega),  I_w = R I_b R^T,
  // omega = M e', omega_dot = Mdot e' + M e''  ([UPSTREAM] EulerConverter::GetM/GetMdot).
  template <class S> static void angular_term(const S e[3], const S ed[3], const S edd[3], const double Ib[3][3], S out[3]) {
    S R[3][3];
    rot_zyx(e, R);
    S y = e[1], z = e[2], yd = ed[1], zd = ed[2];
    S cy = cos(y), sy = sin(y), cz = cos(z), sz = sin(z);
    S zero(0.0), one(1.0);
    S M[3][3] = {{cy * cz, -sz, zero}, {cy * sz, cz, zero}, {-sy, zero, one}};
    S Md[3][3] = {{-(cz * sy * yd) - cy * sz * zd, -(cz * zd), zero},
                  {cy * cz * zd - sy * sz * yd, -(sz * zd), zero},
                  {-(cy * yd), zero, zero}};
    S om[3], omd[3];
    for (int i = 0; i < 3; ++i) {
      om[i] = M[i][0] * ed[0] + M[i][1] * ed[1] + M[i][2] * ed[2];
      omd[i] = Md[i][0] * ed[0] + Md[i][1] * ed[1] + Md[i][2] * ed[2] + M[i][0] * edd[0] + M[i][1] * edd[1] + M[i][2] * edd[2];
    }
    auto apply_Iw = [&](const S v[3], S o[3]) {   // R Ib R^T v
      S a[3], b[3];
      for (int i = 0; i < 3; ++i) a[i] = R[0][i] * v[0] + R[1][i] * v[1] + R[2][i] * v[2];
      for (int i = 0; i < 3; ++i) b[i] = Ib[i][0] * a[0] + Ib[i][1] * a[1] + Ib[i][2] * a[2];
      for (int i = 0; i < 3; ++i) o[i] = R[i][0] * b[0] + R[i][1] * b[1] + R[i][2] * b[2];
    };
    S Iwd[3], Iw[3];
    apply_Iw(omd, Iwd);
    apply_Iw(om, Iw);
    out[0] = Iwd[0] + (om[1] * Iw[2] - om[2] * Iw[1]);
    out[1] = Iwd[1] + (om[2] * Iw[0] - om[0] * Iw[2]);
    out[2] = Iwd[2] + (om[0] * Iw[1] - om[1] * Iw[0]);
  }

  // humanoid_rigid_body_dynamics.cpp:81-87 / leg_length_constraint.cpp:40-42
  int frame_index(double t) const {
    int idx = (int)((t / T) * in.F);
    if (idx >= in.F) idx = in.F - 1;
    if (idx < 0) idx = 0;
    return idx;
  }

  // ------------------------------------------------------------------ eval
  // J: dense m x n row-major (may be null).  grad: n (may be null).
  // H: dense n x n Gauss-Newton Hessian of the (sum-of-squares) objective (may be null).
  // lam (optional, m entries): multipliers of the unscaled rows; when given together with H and the durations are
  // variables, the exact duration-duration block of sum_i lam_i grad^2 c_i + (grad^2 f - Gauss-Newton part) is added to H.
  // second_model: the solver's second model of an iteration (ipm_solver.hpp) -- plain Gauss-Newton, positive semi-definite by construction, WITHOUT the exact
  // blocks (heel-distance curvature, duration-duration and node x duration blocks), any of which can be indefinite: it is what is tried when the first model gives
  // the wrong inertia or no acceptable step.
  void eval(const double* x, double* f_out, double* grad, double* c, double* J, double* H = nullptr, const double* lam = nullptr, bool second_model = false) {
    set_x(x);
    if (J) std::fill(J, J + (size_t)m * n, 0.0);
    if (grad) std::fill(grad, grad + n, 0.0);
    if (H) std::fill(H, H + (size_t)n * n, 0.0);
    std::vector<double> djac, djac2;
    const bool D2 = H && lam && sd.opt_durations && !second_model;      // exact duration-duration block (first model of an iteration)
    auto hdd = [&](int ea, int k, int eb, int l, double v) {      // symmetric entry of the duration block
      const int a = dur_off[ea] + k, b = dur_off[eb] + l;
      H[(size_t)a * n + b] += v;
      if (a != b) H[(size_t)b * n + a] += v;
    };
    auto nvar_of = [&](int e) { return (int)phase_dur[e].size() - 1; };

    // J[row, vars of spline s touched at e] += coef[dim] * w[which][j]
    // (`mask`: in structure mode, the dimensions that are structurally non-zero -- given where a coefficient can be identically zero although the entry is
    //  structural, e.g. the position entries of a foot whose force is pinned to zero in a swing phase; default: the dimensions with a non-zero coefficient)
    auto add_nodes = [&](int row, const Spline& s, const PointEval& e, int which, const double coef[3], int mask = -1) {
      if (!J) return;
      double* Jr = J + (size_t)row * n;
      for (int side = 0; side < 2; ++side)
        for (int deriv = 0; deriv < 2; ++deriv) {
          double w = e.w[which][side * 2 + deriv];
          for (int dim = 0; dim < 3; ++dim) {
            int v = s.vi(e.poly + side, deriv, dim);
            if (v >= 0) Jr[s.var_off + v] += pattern_mode ? ((mask >= 0 ? ((mask >> dim) & 1) != 0 : coef[dim] != 0.0) ? 1.0 : 0.0) : coef[dim] * w;
          }
        }
    };
    auto add_durs = [&](int row, const Spline& s, double t, const PointEval& e, const double coef[3]) {
      if (!J || !sd.opt_durations) return;
      jac_pos_wrt_durations(s, t, e, djac);
      int nv = (int)phase_dur[s.ee].size() - 1;
      double* Jr = J + (size_t)row * n;
      for (int k = 0; k < nv; ++k)
        Jr[dur_off[s.ee] + k] += coef[0] * djac[0 * nv + k] + coef[1] * djac[1 * nv + k] + coef[2] * djac[2 * nv + k];
    };

    // H[vars of (sa, ea, dim ka)] x [vars of (sb, eb, dim kb)] += val * (weights of `wha` at ea) (weights of `whb` at eb)^T, and the transposed
    // block when `sym` (two different input groups of a symmetric local Hessian visited once)
    auto hblock = [&](const Spline& sa, const PointEval& ea, int wha, int ka, const Spline& sb, const PointEval& eb, int whb, int kb, double val, bool sym) {
      if (val == 0.0) return;
      for (int q1 = 0; q1 < 4; ++q1) {
        const int va = sa.vi(ea.poly + q1 / 2, q1 % 2, ka);
        if (va < 0) continue;
        for (int q2 = 0; q2 < 4; ++q2) {
          const int vb = sb.vi(eb.poly + q2 / 2, q2 % 2, kb);
          if (vb < 0) continue;
          const double t = val * ea.w[wha][q1] * eb.w[whb][q2];
          H[(size_t)(sa.var_off + va) * n + sb.var_off + vb] += t;
          if (sym) H[(size_t)(sb.var_off + vb) * n + sa.var_off + va] += t;
        }
      }
    };
    // node x duration block: H[(spline sx, coefficient j of the polynomial at ex, dimension dim)][T_k of end-effector dx's] += wts[j] * (k < cur ? ce : k == cur ? cc : 0), both triangles
    auto xcross = [&](const Spline& sx, const PointEval& ex, int dim, const double* wts, int ee, const DurX& dx, double ce, double cc) {
      for (int j = 0; j < 4; ++j) {
        const int v = sx.vi(ex.poly + j / 2, j % 2, dim);
        if (v < 0 || wts[j] == 0.0) continue;
        const int xv = sx.var_off + v;
        const int ne = dx.cur < dx.nvar ? dx.cur : dx.nvar;
        if (ce != 0.0) for (int k = 0; k < ne; ++k) { const int tv = dur_off[ee] + k; H[(size_t)xv * n + tv] += wts[j] * ce; H[(size_t)tv * n + xv] += wts[j] * ce; }
        if (cc != 0.0 && !dx.last) { const int tv = dur_off[ee] + dx.cur; H[(size_t)xv * n + tv] += wts[j] * cc; H[(size_t)tv * n + xv] += wts[j] * cc; }
      }
    };
    // one (x-spline, T-end-effector) record of a sample: entry = w_j A_class[dim] + om_class,j B[dim]   (om only when the x-spline belongs to that end-effector)
    auto xrecord = [&](const Spline& sx, const PointEval& ex, int ee, const DurX& dx, const double* Ae, const double* Ac, const double* B) {
      for (int dim = 0; dim < 3; ++dim) {
        if (Ae) xcross(sx, ex, dim, ex.w[kPos], ee, dx, Ae[dim], Ac[dim]);
        if (B) { xcross(sx, ex, dim, dx.ome, ee, dx, B[dim], 0.0); xcross(sx, ex, dim, dx.omc, ee, dx, 0.0, B[dim]); }
      }
    };
    const bool DX = D2 && !(study_mask & 32);      // exact node x duration block (first model of an iteration)
    const bool XC = H && lam && (study_mask & 7) && (!second_model || (study_mask & 8));      // study blocks
    PointEval pe, pe2, pl, pa;
    for (const RowBlock& b : blocks) {
      switch (b.family) {
        case FAM_BASEACC: {   // [UPSTREAM] SplineAccConstraint
          const Spline& s = sp[b.ee];
          for (int j = 0; j + 1 < s.n_polys(); ++j) {
            s.eval_local(j, s.poly_dur[j], pe);
            s.eval_local(j + 1, 0.0, pe2);
            for (int d = 0; d < 3; ++d) {
              int row = b.row0 + 3 * j + d;
              c[row] = pe.a[d] - pe2.a[d];
              cl[row] = cu[row] = 0.0;
              double cf[3] = {0, 0, 0};
              cf[d] = 1.0; add_nodes(row, s, pe, kAcc, cf);
              cf[d] = -1.0; add_nodes(row, s, pe2, kAcc, cf);
            }
          }
        } break;
        case FAM_TERRAIN: {   // [UPSTREAM] TerrainConstraint
          const Spline& s = sp[2 + b.ee];
          int row = b.row0;
          for (int nd = 0; nd < s.n_nodes; ++nd) {
            if (!terrain_row_kept(s, nd)) continue;
            double px = s.nv(nd, 0, 0), py = s.nv(nd, 0, 1), pz = s.nv(nd, 0, 2);
            c[row] = pz - plane_height(px, py);
            if (s.is_const_node(nd)) { cl[row] = cu[row] = 0.0; } else { cl[row] = 0.0; cu[row] = kInf; }
            if (J) {
              double* Jr = J + (size_t)row * n;
              int vz = s.vi(nd, 0, 2), vx = s.vi(nd, 0, 0), vy = s.vi(nd, 0, 1);
              if (vz >= 0) Jr[s.var_off + vz] += 1.0;
              if (vx >= 0) Jr[s.var_off + vx] += -hx;
              if (vy >= 0) Jr[s.var_off + vy] += -hy;
            }
            ++row;
          }
        } break;
        case FAM_ROM: {       // leg_length_constraint.cpp:36-111
          const int e = b.ee;
          const Spline& sm = sp[2 + e];
          const double L = (e == 0 || e == 1) ? in.leg_len : in.heel_len;     // :21-27
          const std::vector<double>& hips = (e == 0 || e == 2) ? in.hip_l : in.hip_r;   // humanoid.h:45-48
          for (size_t k = 0; k < t_rom.size(); ++k) {
            double t = t_rom[k];
            int row = b.row0 + (int)k;
            int fi = frame_index(t);
            const double* h = &hips[fi * 3];
            sp[0].eval(t, pl); sp[1].eval(t, pa); sm.eval(t, pe);
            Dual<3> eu[3] = {Dual<3>::var(pa.p[0], 0), Dual<3>::var(pa.p[1], 1), Dual<3>::var(pa.p[2], 2)};
            Dual<3> R[3][3];
            rot_zyx(eu, R);
            double d[3];
            double dRh[3][3];   // dRh[i][k] = d (R h)_i / d euler_k
            for (int i = 0; i < 3; ++i) {
              Dual<3> rh = R[i][0] * h[0] + R[i][1] * h[1] + R[i][2] * h[2];
              d[i] = pe.p[i] - (rh.v + pl.p[i]);
              for (int kk = 0; kk < 3; ++kk) dRh[i][kk] = rh.d[kk];
            }
            c[row] = 0.5 * (d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
            cl[row] = 0.0; cu[row] = 0.5 * L * L;                       // :59
            double cf[3] = {-d[0], -d[1], -d[2]};
            add_nodes(row, sp[0], pl, kPos, cf, 7);
            double ca[3];
            for (int kk = 0; kk < 3; ++kk) ca[kk] = -(d[0] * dRh[0][kk] + d[1] * dRh[1][kk] + d[2] * dRh[2][kk]);
            add_nodes(row, sp[1], pa, kPos, ca, 7);
            add_nodes(row, sm, pe, kPos, d, 7);
            add_durs(row, sm, t, pe, d);
            if (XC && (study_mask & 4) && lam[row] != 0.0) {
              // exact node-node block of lam grad^2 c, c = 1/2 |d|^2, d = p_ee - R(euler) h - c_com (leg_length_constraint.cpp:46-52):
              //   grad^2 c = sum_i grad d_i grad d_i^T + sum_i d_i grad^2 d_i, the second part living in the Euler angles only
              Jet2<3> e2[3] = {Jet2<3>::var(pa.p[0], 0), Jet2<3>::var(pa.p[1], 1), Jet2<3>::var(pa.p[2], 2)};
              Jet2<3> R2[3][3];
              rot_zyx(e2, R2);
              double Hth[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}};
              for (int i = 0; i < 3; ++i) {
                Jet2<3> rh = R2[i][0] * h[0] + R2[i][1] * h[1] + R2[i][2] * h[2];
                for (int a = 0; a < 3; ++a) for (int b2 = 0; b2 < 3; ++b2) Hth[a][b2] += dRh[i][a] * dRh[i][b2] - d[i] * rh.h[a][b2];
              }
              const double lm = lam[row];
              for (int a = 0; a < 3; ++a) {
                hblock(sp[0], pl, kPos, a, sp[0], pl, kPos, a, lm, false);            // com x com: I
                hblock(sm, pe, kPos, a, sm, pe, kPos, a, lm, false);                  // ee x ee: I
                hblock(sm, pe, kPos, a, sp[0], pl, kPos, a, -lm, true);               // ee x com: -I
                for (int b2 = 0; b2 < 3; ++b2) {
                  hblock(sp[1], pa, kPos, a, sp[1], pa, kPos, b2, lm * Hth[a][b2], false);
                  hblock(sp[0], pl, kPos, a, sp[1], pa, kPos, b2, lm * dRh[a][b2], true);       // (-e_a) . (-dRh[a][b])
                  hblock(sm, pe, kPos, a, sp[1], pa, kPos, b2, -lm * dRh[a][b2], true);
                }
              }
            }
            if (D2)
              for (int k2 = 0; k2 < nvar_of(e); ++k2)
                for (int l2 = 0; l2 <= k2; ++l2) {
                  double gk[3], gl[3], q2[3];
                  d1pos(sm, t, pe, k2, gk); d1pos(sm, t, pe, l2, gl); d2pos(sm, t, pe, k2, l2, q2);
                  hdd(e, k2, e, l2, lam[row] * (gk[0] * gl[0] + gk[1] * gl[1] + gk[2] * gl[2] + d[0] * q2[0] + d[1] * q2[1] + d[2] * q2[2]));
                }
            if (DX && lam[row] != 0.0) {
              DurX dx; dur_cross(sm, t, pe, dx);
              const double lm = lam[row];
              double Ae[3], Ac[3], Bv[3], nAe[3], nAc[3], tAe[3], tAc[3];
              for (int i = 0; i < 3; ++i) { Ae[i] = lm * dx.Ge[i]; Ac[i] = lm * dx.Gc[i]; Bv[i] = lm * d[i]; nAe[i] = -Ae[i]; nAc[i] = -Ac[i]; }
              for (int kk = 0; kk < 3; ++kk) { tAe[kk] = -(dRh[0][kk] * Ae[0] + dRh[1][kk] * Ae[1] + dRh[2][kk] * Ae[2]); tAc[kk] = -(dRh[0][kk] * Ac[0] + dRh[1][kk] * Ac[1] + dRh[2][kk] * Ac[2]); }
              xrecord(sm, pe, e, dx, Ae, Ac, Bv);
              xrecord(sp[0], pl, e, dx, nAe, nAc, nullptr);
              xrecord(sp[1], pa, e, dx, tAe, tAc, nullptr);
            }
          }
        } break;
        case FAM_HEELDIST: {  // ee_dist_constraint.cpp:29-94 ; pairs nlp_formulation.cpp:249-257
          const int e1 = b.ee, e2 = b.ee + 2;
          for (size_t k = 0; k < t_rom.size(); ++k) {
            double t = t_rom[k];
            int row = b.row0 + (int)k;
            sp[2 + e1].eval(t, pe); sp[2 + e2].eval(t, pe2);
            double d[3] = {pe.p[0] - pe2.p[0], pe.p[1] - pe2.p[1], pe.p[2] - pe2.p[2]};
            double md[3] = {-d[0], -d[1], -d[2]};
            c[row] = 0.5 * (d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
            cl[row] = cu[row] = 0.5 * in.heel_dist * in.heel_dist;     // :39
            add_nodes(row, sp[2 + e1], pe, kPos, d, 7);
            add_nodes(row, sp[2 + e2], pe2, kPos, md, 7);
            add_durs(row, sp[2 + e1], t, pe, d);
            add_durs(row, sp[2 + e2], t, pe2, md);
            if (H && lam) {
              // exact node-node block of lam * grad^2 c for this row: c = 1/2 |p_toe - p_heel|^2 is quadratic in the node values
              // (for fixed durations), grad^2 c = sum_dim g_dim g_dim^T with g_dim = d(p_toe - p_heel)[dim] / d(node variables).
              // The Gauss-Newton model lacks it, and the multipliers of these equality rows reach 10^2..10^3: without the term
              // the damping has to stand in for it and the optimality error hovers just above tol for 100+ iterations.
              for (int dim = 0; dim < 3; ++dim) {
                int gv[8]; double gw[8]; int ng = 0;
                auto push = [&](const Spline& s, const PointEval& e, double sign) {
                  for (int side = 0; side < 2; ++side)
                    for (int deriv = 0; deriv < 2; ++deriv) {
                      const int v = s.vi(e.poly + side, deriv, dim);
                      if (v < 0) continue;
                      const int gidx = s.var_off + v;
                      const double w = sign * e.w[kPos][side * 2 + deriv];
                      int at = -1;
                      for (int q = 0; q < ng; ++q) if (gv[q] == gidx) at = q;      // a stance position is one variable for two nodes
                      if (at >= 0) gw[at] += w; else { gv[ng] = gidx; gw[ng] = w; ++ng; }
                    }
                };
                push(sp[2 + e1], pe, 1.0); push(sp[2 + e2], pe2, -1.0);
                for (int a = 0; a < ng; ++a)
                  for (int b2 = 0; b2 < ng; ++b2) H[(size_t)gv[a] * n + gv[b2]] += (second_model ? ((study_mask & 16) ? std::min(std::max(lam[row], 0.0), clip_cap) : 0.0) : lam[row]) * gw[a] * gw[b2];
              }
            }
            if (D2) {
              auto dot = [](const double* a, const double* b2) { return a[0] * b2[0] + a[1] * b2[1] + a[2] * b2[2]; };
              for (int k2 = 0; k2 < nvar_of(e1); ++k2)
                for (int l2 = 0; l2 <= k2; ++l2) {
                  double gk[3], gl[3], q2[3];
                  d1pos(sp[2 + e1], t, pe, k2, gk); d1pos(sp[2 + e1], t, pe, l2, gl); d2pos(sp[2 + e1], t, pe, k2, l2, q2);
                  hdd(e1, k2, e1, l2, lam[row] * (dot(gk, gl) + dot(d, q2)));
                }
              for (int k2 = 0; k2 < nvar_of(e2); ++k2)
                for (int l2 = 0; l2 <= k2; ++l2) {
                  double gk[3], gl[3], q2[3];
                  d1pos(sp[2 + e2], t, pe2, k2, gk); d1pos(sp[2 + e2], t, pe2, l2, gl); d2pos(sp[2 + e2], t, pe2, k2, l2, q2);
                  hdd(e2, k2, e2, l2, lam[row] * (dot(gk, gl) - dot(d, q2)));
                }
              for (int k2 = 0; k2 < nvar_of(e1); ++k2)
                for (int l2 = 0; l2 < nvar_of(e2); ++l2) {
                  double ga[3], gb[3];
                  d1pos(sp[2 + e1], t, pe, k2, ga); d1pos(sp[2 + e2], t, pe2, l2, gb);
                  hdd(e1, k2, e2, l2, -lam[row] * dot(ga, gb));
                }
            }
            if (DX && lam[row] != 0.0) {
              DurX da, db; dur_cross(sp[2 + e1], t, pe, da); dur_cross(sp[2 + e2], t, pe2, db);
              const double lm = lam[row];
              double Aae[3], Aac[3], Abe[3], Abc[3], nAae[3], nAac[3], nAbe[3], nAbc[3], Ba[3], Bb[3];
              for (int i = 0; i < 3; ++i) {
                Aae[i] = lm * da.Ge[i]; Aac[i] = lm * da.Gc[i]; Abe[i] = lm * db.Ge[i]; Abc[i] = lm * db.Gc[i];
                nAae[i] = -Aae[i]; nAac[i] = -Aac[i]; nAbe[i] = -Abe[i]; nAbc[i] = -Abc[i]; Ba[i] = lm * d[i]; Bb[i] = -lm * d[i];
              }
              xrecord(sp[2 + e1], pe, e1, da, Aae, Aac, Ba);            // toe nodes x toe durations
              xrecord(sp[2 + e2], pe2, e1, da, nAae, nAac, nullptr);    // heel nodes x toe durations
              xrecord(sp[2 + e1], pe, e2, db, nAbe, nAbc, nullptr);     // toe nodes x heel durations
              xrecord(sp[2 + e2], pe2, e2, db, Abe, Abc, Bb);           // heel nodes x heel durations
            }
          }
        } break;
        case FAM_DYNAMIC: {   // humanoid_dynamic_constraint.cpp:63-143 + humanoid_rigid_body_dynamics.cpp:89-206
          PointEval pf[4], pm[4];
          for (size_t k = 0; k < t_dyn.size(); ++k) {
            double t = t_dyn[k];
            int row0 = b.row0 + 6 * (int)k;
            sp[0].eval(t, pl); sp[1].eval(t, pa);
            for (int e = 0; e < 4; ++e) { sp[2 + e].eval(t, pm[e]); sp[6 + e].eval(t, pf[e]); }
            int fi = frame_index(t);
            const double* I6 = &in.inertia[fi * 6];
            double Ib[3][3] = {{I6[0], I6[3], I6[4]}, {I6[3], I6[1], I6[5]}, {I6[4], I6[5], I6[2]}};   // :47-56
            Dual<9> eu[3], ed[3], edd[3], ang[3];
            for (int i = 0; i < 3; ++i) { eu[i] = Dual<9>::var(pa.p[i], i); ed[i] = Dual<9>::var(pa.v[i], 3 + i); edd[i] = Dual<9>::var(pa.a[i], 6 + i); }
            angular_term(eu, ed, edd, Ib, ang);
            double tau[3] = {0, 0, 0}, fsum[3] = {0, 0, 0};
            for (int e = 0; e < 4; ++e) {
              const double* f = pf[e].p;
              double r[3] = {pl.p[0] - pm[e].p[0], pl.p[1] - pm[e].p[1], pl.p[2] - pm[e].p[2]};
              tau[0] += f[1] * r[2] - f[2] * r[1];
              tau[1] += f[2] * r[0] - f[0] * r[2];
              tau[2] += f[0] * r[1] - f[1] * r[0];
              for (int d = 0; d < 3; ++d) fsum[d] += f[d];
            }
            for (int d = 0; d < 3; ++d) {
              c[row0 + d] = ang[d].v - tau[d];                                                   // rows AX..AZ
              c[row0 + 3 + d] = in.mass * pl.a[d] - fsum[d] - in.mass * kGravity * gdir[d];      // rows LX..LZ
              cl[row0 + d] = cu[row0 + d] = 0.0; cl[row0 + 3 + d] = cu[row0 + 3 + d] = 0.0;
            }
            if (!J) continue;
            // cross(v) row i as coefficient vector: (v x u)_i = sum_j X[i][j] u_j
            auto crossmat = [](const double v[3], double X[3][3]) {
              X[0][0] = 0; X[0][1] = -v[2]; X[0][2] = v[1];
              X[1][0] = v[2]; X[1][1] = 0; X[1][2] = -v[0];
              X[2][0] = -v[1]; X[2][1] = v[0]; X[2][2] = 0;
            };
            for (int i = 0; i < 3; ++i) {
              // base-lin: angular rows -sum_e cross(f_e) dc ; linear rows m * d(acc)
              double cf[3] = {0, 0, 0};
              for (int e = 0; e < 4; ++e) { double X[3][3]; crossmat(pf[e].p, X); for (int j = 0; j < 3; ++j) cf[j] -= X[i][j]; }
              add_nodes(row0 + i, sp[0], pl, kPos, cf, 7 & ~(1 << i));
              double cm[3] = {0, 0, 0}; cm[i] = in.mass;
              add_nodes(row0 + 3 + i, sp[0], pl, kAcc, cm, 1 << i);
              // base-ang: chain rule through (e, e', e'')
              double c0[3] = {ang[i].d[0], ang[i].d[1], ang[i].d[2]};
              double c1[3] = {ang[i].d[3], ang[i].d[4], ang[i].d[5]};
              double c2[3] = {ang[i].d[6], ang[i].d[7], ang[i].d[8]};
              add_nodes(row0 + i, sp[1], pa, kPos, c0, 7);
              add_nodes(row0 + i, sp[1], pa, kVel, c1, 7);
              add_nodes(row0 + i, sp[1], pa, kAcc, c2, 7);
              for (int e = 0; e < 4; ++e) {
                double r[3] = {pl.p[0] - pm[e].p[0], pl.p[1] - pm[e].p[1], pl.p[2] - pm[e].p[2]};
                double Xr[3][3], Xf[3][3];
                crossmat(r, Xr); crossmat(pf[e].p, Xf);
                // force: angular +cross(r) df ; linear -df
                add_nodes(row0 + i, sp[6 + e], pf[e], kPos, Xr[i], 7 & ~(1 << i));
                double ml[3] = {0, 0, 0}; ml[i] = -1.0;
                add_nodes(row0 + 3 + i, sp[6 + e], pf[e], kPos, ml, 1 << i);
                // ee position: angular +cross(f) dp
                add_nodes(row0 + i, sp[2 + e], pm[e], kPos, Xf[i], 7 & ~(1 << i));
                // schedule: force spline then motion spline (humanoid_dynamic_constraint.cpp:112-118)
                add_durs(row0 + i, sp[6 + e], t, pf[e], Xr[i]);
                add_durs(row0 + 3 + i, sp[6 + e], t, pf[e], ml);
                add_durs(row0 + i, sp[2 + e], t, pm[e], Xf[i]);
              }
            }
            if (XC && (study_mask & 3)) {
              const double* la = lam + row0;
              if (study_mask & 1)
                for (int e = 0; e < 4; ++e)
                  for (int j = 0; j < 3; ++j)
                    for (int k2 = 0; k2 < 3; ++k2) {
                      if (j == k2) continue;
                      // rows ang_i - sum_e (f_e x r_e)_i, r_e = c - p_e (humanoid_rigid_body_dynamics.cpp:97-99): d2 L / d f_j d r_k = -sum_i lam_i eps_ijk
                      const int i = 3 - j - k2;
                      const double eps = ((j + 1) % 3 == k2) ? 1.0 : -1.0;         // eps_ijk with (i, j, k) a permutation of (0, 1, 2): +1 when k follows j cyclically
                      const double mjk = -la[i] * eps;
                      hblock(sp[6 + e], pf[e], kPos, j, sp[0], pl, kPos, k2, mjk, true);
                      hblock(sp[6 + e], pf[e], kPos, j, sp[2 + e], pm[e], kPos, k2, -mjk, true);
                    }
              if (study_mask & 2) {
                // angular term I_w wd + w x I_w w as a function of (euler, euler', euler''): second derivatives by second-order AD
                Jet2<9> e2[3], ed2[3], edd2[3], ang2[3];
                for (int i = 0; i < 3; ++i) { e2[i] = Jet2<9>::var(pa.p[i], i); ed2[i] = Jet2<9>::var(pa.v[i], 3 + i); edd2[i] = Jet2<9>::var(pa.a[i], 6 + i); }
                angular_term(e2, ed2, edd2, Ib, ang2);
                for (int a = 0; a < 9; ++a)
                  for (int b2 = 0; b2 < 9; ++b2) {
                    const double v = la[0] * ang2[0].h[a][b2] + la[1] * ang2[1].h[a][b2] + la[2] * ang2[2].h[a][b2];
                    hblock(sp[1], pa, a / 3, a % 3, sp[1], pa, b2 / 3, b2 % 3, v, false);
                  }
              }
            }
            if (D2) {
              auto cross = [](const double* a, const double* b2, double* o) { o[0] = a[1] * b2[2] - a[2] * b2[1]; o[1] = a[2] * b2[0] - a[0] * b2[2]; o[2] = a[0] * b2[1] - a[1] * b2[0]; };
              for (int e = 0; e < 4; ++e) {
                const double r[3] = {pl.p[0] - pm[e].p[0], pl.p[1] - pm[e].p[1], pl.p[2] - pm[e].p[2]};
                for (int k2 = 0; k2 < nvar_of(e); ++k2)
                  for (int l2 = 0; l2 <= k2; ++l2) {
                    double gFk[3], gFl[3], gPk[3], gPl[3], qF[3], qP[3], a1[3], a2[3], a3[3], a4[3];
                    d1pos(sp[6 + e], t, pf[e], k2, gFk); d1pos(sp[6 + e], t, pf[e], l2, gFl);
                    d1pos(sp[2 + e], t, pm[e], k2, gPk); d1pos(sp[2 + e], t, pm[e], l2, gPl);
                    d2pos(sp[6 + e], t, pf[e], k2, l2, qF); d2pos(sp[2 + e], t, pm[e], k2, l2, qP);
                    // rows: ang - sum F x (c - p) ; m a - sum F
                    cross(qF, r, a1); cross(gFk, gPl, a2); cross(gFl, gPk, a3); cross(pf[e].p, qP, a4);
                    double v = 0;
                    for (int i = 0; i < 3; ++i) v += -lam[row0 + i] * (a1[i] - a2[i] - a3[i] - a4[i]) - lam[row0 + 3 + i] * qF[i];
                    hdd(e, k2, e, l2, v);
                  }
              }
            }
            if (DX) {
              auto cross = [](const double* a, const double* b2, double* o) { o[0] = a[1] * b2[2] - a[2] * b2[1]; o[1] = a[2] * b2[0] - a[0] * b2[2]; o[2] = a[0] * b2[1] - a[1] * b2[0]; };
              const double* la = lam + row0; const double* ll = lam + row0 + 3;
              for (int e = 0; e < 4; ++e) {
                // L = la . [ang - sum_e f_e x r_e] + ll . [m a - sum_e f_e], r_e = c - p_e:
                //   dL/df_e = -(r_e x la) - ll,  dL/dp_e = la x f_e,  dL/dc = -sum_e la x f_e
                DurX dF, dP; dur_cross(sp[6 + e], t, pf[e], dF); dur_cross(sp[2 + e], t, pm[e], dP);
                const double r[3] = {pl.p[0] - pm[e].p[0], pl.p[1] - pm[e].p[1], pl.p[2] - pm[e].p[2]};
                double rxl[3], lxf[3], lxGPe[3], lxGPc[3], lxGFe[3], lxGFc[3], BF[3], AFe[3], AFc[3], nlxGFe[3], nlxGFc[3];
                cross(r, la, rxl); cross(la, pf[e].p, lxf);
                cross(la, dP.Ge, lxGPe); cross(la, dP.Gc, lxGPc); cross(la, dF.Ge, lxGFe); cross(la, dF.Gc, lxGFc);
                for (int i = 0; i < 3; ++i) { BF[i] = -rxl[i] - ll[i]; AFe[i] = -lxGPe[i]; AFc[i] = -lxGPc[i]; nlxGFe[i] = -lxGFe[i]; nlxGFc[i] = -lxGFc[i]; }
                xrecord(sp[6 + e], pf[e], e, dF, AFe, AFc, BF);              // force nodes
                xrecord(sp[2 + e], pm[e], e, dP, lxGFe, lxGFc, lxf);         // position nodes
                xrecord(sp[0], pl, e, dP, nlxGFe, nlxGFc, nullptr);          // centre-of-mass nodes
              }
            }
          }
        } break;
        case FAM_FORCE: {     // [UPSTREAM] ForceConstraint (5 rows per non-constant force node)
          const Spline& s = sp[6 + b.ee];
          int row = b.row0;
          const double mu = kFriction, fmax = 1000.0;    // parameters.cpp:56
          for (int nd = 0; nd < s.n_nodes; ++nd) {
            if (s.is_const_node(nd)) continue;
            double f[3] = {s.nv(nd, 0, 0), s.nv(nd, 0, 1), s.nv(nd, 0, 2)};
            double dirs[5][3];
            for (int d = 0; d < 3; ++d) {
              dirs[0][d] = nrm_n[d];
              dirs[1][d] = nrm_t1[d] - mu * nrm_n[d];
              dirs[2][d] = nrm_t1[d] + mu * nrm_n[d];
              dirs[3][d] = nrm_t2[d] - mu * nrm_n[d];
              dirs[4][d] = nrm_t2[d] + mu * nrm_n[d];
            }
            const double lo[5] = {0.0, -kInf, 0.0, -kInf, 0.0};
            const double hi[5] = {fmax, 0.0, kInf, 0.0, kInf};
            for (int q = 0; q < 5; ++q) {
              c[row + q] = f[0] * dirs[q][0] + f[1] * dirs[q][1] + f[2] * dirs[q][2];
              cl[row + q] = lo[q]; cu[row + q] = hi[q];
              if (J) for (int d = 0; d < 3; ++d) { int v = s.vi(nd, 0, d); if (v >= 0) J[(size_t)(row + q) * n + s.var_off + v] += dirs[q][d]; }
            }
            row += 5;
          }
        } break;
        case FAM_HEIGHT: {    // height_constraint.cpp:24-58 (un-normalised file normal)
          const Spline& s = sp[2 + b.ee];
          for (size_t k = 0; k < t_height_kept[b.ee].size(); ++k) {
            double t = t_height_kept[b.ee][k];
            int row = b.row0 + (int)k;
            s.eval(t, pe);
            c[row] = in.normal[0] * (pe.p[0] - in.point[0]) + in.normal[1] * (pe.p[1] - in.point[1]) + in.normal[2] * (pe.p[2] - in.point[2]);
            cl[row] = 0.0; cu[row] = kInf;
            add_nodes(row, s, pe, kPos, in.normal);
            add_durs(row, s, t, pe, in.normal);
            if (D2)
              for (int k2 = 0; k2 < nvar_of(b.ee); ++k2)
                for (int l2 = 0; l2 <= k2; ++l2) {
                  double q2[3];
                  d2pos(s, t, pe, k2, l2, q2);
                  hdd(b.ee, k2, b.ee, l2, lam[row] * (in.normal[0] * q2[0] + in.normal[1] * q2[1] + in.normal[2] * q2[2]));
                }
            if (DX && lam[row] != 0.0) {
              DurX dx; dur_cross(s, t, pe, dx);
              const double Bv[3] = {lam[row] * in.normal[0], lam[row] * in.normal[1], lam[row] * in.normal[2]};
              xrecord(s, pe, b.ee, dx, nullptr, nullptr, Bv);
            }
          }
        } break;
        case FAM_TOTALTIME: { // total_duration_constraint.cpp:60-82
          int nv = (int)phase_dur[b.ee].size() - 1;
          double sum = 0;
          for (int k = 0; k < nv; ++k) sum += phase_dur[b.ee][k];
          c[b.row0] = sum;
          cl[b.row0] = std::max(0.0, T - 500.0); cu[b.row0] = T - 0.0;     // parameters.cpp:60
          if (J) for (int k = 0; k < nv; ++k) J[(size_t)b.row0 * n + dur_off[b.ee] + k] = 1.0;
        } break;
        case -1: {            // [UPSTREAM] PhaseDurations::GetBounds -> (0, 500) on each duration variable
          for (int k = 0; k < b.count; ++k) {
            c[b.row0 + k] = phase_dur[b.ee][k];
            cl[b.row0 + k] = 0.0; cu[b.row0 + k] = 500.0;
            if (J) J[(size_t)(b.row0 + k) * n + dur_off[b.ee] + k] = 1.0;
          }
        } break;
      }
    }

    // ------------------------------------------------------------ objective
    // f = sum 1/2 w r^2 ; every residual r is evaluated with its sparse gradient so that
    // grad = sum w r dr and the Gauss-Newton Hessian H = sum w dr dr^T are consistent.
    double f = 0.0;
    std::vector<int> gi; std::vector<double> gv;
    auto flush = [&](double w, double r) {
      f += 0.5 * w * r * r;
      if (grad) for (size_t a = 0; a < gi.size(); ++a) grad[gi[a]] += w * r * gv[a];
      if (H) for (size_t a = 0; a < gi.size(); ++a) for (size_t bb = 0; bb < gi.size(); ++bb) H[(size_t)gi[a] * n + gi[bb]] += w * gv[a] * gv[bb];
    };
    auto push_nodes = [&](const Spline& s, const PointEval& e, int which, int dim, double sign) {
      for (int side = 0; side < 2; ++side)
        for (int deriv = 0; deriv < 2; ++deriv) {
          int v = s.vi(e.poly + side, deriv, dim);
          if (v >= 0) { gi.push_back(s.var_off + v); gv.push_back(sign * e.w[which][side * 2 + deriv]); }
        }
    };
    auto push_durs = [&](const Spline& s, double t, const PointEval& e, int dim, double sign) {
      if (!sd.opt_durations || !s.phase_based) return;
      jac_pos_wrt_durations(s, t, e, djac2);
      int nv = (int)phase_dur[s.ee].size() - 1;
      for (int k = 0; k < nv; ++k) { gi.push_back(dur_off[s.ee] + k); gv.push_back(sign * djac2[dim * nv + k]); }
    };
    const bool need_g = grad || H;
    // DataCost  data_cost.cpp:40-96 ; six terms phys_optim.cpp:314-333
    for (int term = 0; term < 6; ++term) {
      const Spline& s = sp[term];
      const std::vector<double>& data = term == 0 ? in.com : term == 1 ? in.euler : *ee_data[term - 2];
      double w = term == 0 ? sd.w_data[0] : term == 1 ? sd.w_data[1] : sd.w_data[2];
      double t = 0.0;
      for (int i = 0; i < in.F; ++i) {
        s.eval(t, pe);
        for (int d = 0; d < 3; ++d) {
          double r = data[i * 3 + d] - pe.p[d];
          gi.clear(); gv.clear();
          if (need_g) { push_nodes(s, pe, kPos, d, -1.0); push_durs(s, t, pe, d, -1.0); }
          flush(w, r);
          if (D2 && s.phase_based)
            for (int k2 = 0; k2 < nvar_of(s.ee); ++k2)
              for (int l2 = 0; l2 <= k2; ++l2) { double q2[3]; d2pos(s, t, pe, k2, l2, q2); hdd(s.ee, k2, s.ee, l2, -w * r * q2[d]); }
          if (DX && s.phase_based) {      // residual curvature of the node x duration block: -w r d2p/dx dT (the product of first derivatives is the Gauss-Newton part)
            DurX dx; dur_cross(s, t, pe, dx);
            xcross(s, pe, d, dx.ome, s.ee, dx, -w * r, 0.0); xcross(s, pe, d, dx.omc, s.ee, dx, 0.0, -w * r);
          }
        }
        t += in.dt;                                                   // :48
      }
    }
    // VelSmoothCost  vel_smooth_cost.cpp:37-100 ; terms phys_optim.cpp:335-373
    for (int pass = 0; pass < 2; ++pass) {
      const double* ww = pass == 0 ? sd.w_vel : sd.w_acc;
      if (ww[0] < 0) continue;
      const int which = pass == 0 ? kPos : kVel;
      for (int term = 0; term < 6; ++term) {
        const Spline& s = sp[term];
        double w = term == 0 ? ww[0] : term == 1 ? ww[1] : ww[2];
        double Ttot = s.total_time();
        for (double t = 0.0; t < (Ttot - in.dt) - 1e-9; t += in.dt) {   // :41; the last sample sits exactly on the limit — decided with a tolerance instead of by rounding
          s.eval(t + in.dt, pe2); s.eval(t, pe);
          for (int d = 0; d < 3; ++d) {
            double r = (which == kPos ? pe2.p[d] - pe.p[d] : pe2.v[d] - pe.v[d]);
            gi.clear(); gv.clear();
            if (need_g) {
              push_nodes(s, pe2, which, d, 1.0); push_nodes(s, pe, which, d, -1.0);
              if (which == kPos) { push_durs(s, t + in.dt, pe2, d, 1.0); push_durs(s, t, pe, d, -1.0); }   // :55-70
            }
            flush(w, r);
            if (D2 && s.phase_based && which == kPos)
              for (int k2 = 0; k2 < nvar_of(s.ee); ++k2)
                for (int l2 = 0; l2 <= k2; ++l2) {
                  double qa[3], qb[3];
                  d2pos(s, t + in.dt, pe2, k2, l2, qb); d2pos(s, t, pe, k2, l2, qa);
                  hdd(s.ee, k2, s.ee, l2, w * r * (qb[d] - qa[d]));
                }
            if (DX && s.phase_based && which == kPos) {
              DurX da, db; dur_cross(s, t, pe, da); dur_cross(s, t + in.dt, pe2, db);
              xcross(s, pe2, d, db.ome, s.ee, db, w * r, 0.0); xcross(s, pe2, d, db.omc, s.ee, db, 0.0, w * r);
              xcross(s, pe, d, da.ome, s.ee, da, -w * r, 0.0); xcross(s, pe, d, da.omc, s.ee, da, 0.0, -w * r);
            }
          }
        }
      }
    }
    // DurationCost  duration_cost.cpp:25-50 (compares with the INPUT durations)
    if (sd.w_dur >= 0 && sd.opt_durations)
      for (int e = 0; e < 4; ++e)
        for (size_t k = 0; k + 1 < phase_dur[e].size(); ++k) {
          double r = phase_dur0[e][k] - phase_dur[e][k];
          gi.clear(); gv.clear();
          gi.push_back(dur_off[e] + (int)k); gv.push_back(-1.0);
          flush(sd.w_dur, r);
        }
    if (f_out) *f_out = f;
  }

  // --------------------------------------------------------------- output
  // SaveSolution phys_optim.cpp:63-143.  Arrays are resized to the number of samples
  // produced by the reference's `while (t <= T + 1e-5)` loop.
  struct Solution {
    int num_frames_header = 0;   // int((T+1e-5)/dt)+1   (:71)
    int n_samples = 0;           // loop count
    std::vector<double> base_lin, base_ang_deg;    // n_samples*3
    std::vector<double> ee_pos[4], ee_force[4];    // n_samples*3
    std::vector<int> contact[4];                   // n_samples
  };
  Solution sample_solution() const {
    Solution so;
    const double tot = sp[0].total_time();          // :69
    const double dt = in.dt;
    so.num_frames_header = (int)((tot + 1e-5) / dt) + 1;
    PointEval pe;
    double t = 0.0;
    while (t <= tot + 1e-5) {
      sp[0].eval(t, pe);
      for (int d = 0; d < 3; ++d) so.base_lin.push_back(pe.p[d]);
      sp[1].eval(t, pe);
      for (int d = 0; d < 3; ++d) so.base_ang_deg.push_back(pe.p[d] / M_PI * 180);     // :97
      for (int e = 0; e < 4; ++e) {
        sp[2 + e].eval(t, pe);
        for (int d = 0; d < 3; ++d) so.ee_pos[e].push_back(pe.p[d]);
        sp[6 + e].eval(t, pe);
        for (int d = 0; d < 3; ++d) so.ee_force[e].push_back(pe.p[d]);
        int ph = segment_id(t, phase_dur[e]);       // [UPSTREAM] PhaseDurations::IsContactPhase
        bool cflag = (ph % 2 == 0) ? ee_start_contact[e] : !ee_start_contact[e];
        so.contact[e].push_back(cflag ? 1 : 0);
      }
      t += dt;
      so.n_samples++;
    }
    return so;
  }
};

}  // namespace orc

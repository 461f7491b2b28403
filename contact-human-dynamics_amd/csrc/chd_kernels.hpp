// chd_kernels.hpp — the physics hot path: one workgroup solves one sequence's staged NLP.
//
// What this replaces in the reference (all of it runs per IPOPT iteration on one CPU thread):
//   spline evaluation / Jacobians        TOWR NodeSpline, PhaseSpline, CubicHermitePolynomial (absent fork)
//   dynamics residual + Jacobians        humanoid_dynamic_constraint.cpp:63-143, humanoid_rigid_body_dynamics.cpp:89-206
//   leg length / heel distance / height  leg_length_constraint.cpp:36-111, ee_dist_constraint.cpp:29-94, height_constraint.cpp:24-58
//   terrain / force / base-acc rows      TOWR TerrainConstraint, ForceConstraint, SplineAccConstraint
//   duration rows                        total_duration_constraint.cpp:60-82
//   costs                                data_cost.cpp:40-96, vel_smooth_cost.cpp:37-100, duration_cost.cpp:25-50
//   the NLP solve                        ifopt::IpoptSolver / IPOPT (phys_optim.cpp:567-580)
//   SaveSolution                         phys_optim.cpp:63-143
//
// Execution model: the code is written as workgroup-wide phases — PAR_FOR loops separated by
// CHD_SYNC() — over data that lives in the sequence's HBM workspace, with the KKT panel, the
// substitution vector and the border Schur complement staged in LDS.  Every accumulation is
// owner-computes (one thread owns a KKT entry / gradient entry), so results are bitwise
// reproducible for a given workgroup size and independent of where the workgroup runs.
//
// The same source compiles for gfx950 (hipcc) and, with CHD_HOST_EMU, as a single-"thread"
// host function that tests use to debug phases without a GPU.  The emulation build is test
// infrastructure; libchd_phys.so contains only the HIP build.
#pragma once
#include <math.h>
#ifdef CHD_HOST_EMU
#include <cstdio>
#include <cstdlib>
#endif

#include "chd_device.hpp"

#ifdef CHD_HOST_EMU
#define CHD_DEV static inline
#define CHD_TID 0
#define CHD_NT 1
#define CHD_SYNC() ((void)0)
#define CHD_GL 1
#define CHD_NOINLINE
#define CHD_ALWAYS_INLINE
#else
#define CHD_DEV __device__ inline
#define CHD_TID ((int)threadIdx.x)
#define CHD_NT ((int)blockDim.x)
#define CHD_SYNC() __syncthreads()
#define CHD_GL 16
// internal linkage: with every caller known the compiler drops the callee-saved register convention (no
// prologue/epilogue scratch traffic in functions that need more than the 144 caller-saved VGPRs)
#define CHD_NOINLINE static __attribute__((noinline))
#define CHD_ALWAYS_INLINE __attribute__((always_inline))
#endif
// LDS data is addressed through an explicit local-address-space pointer: a generic `double*` would make
// hipcc emit flat_load/flat_store for every access instead of ds_read/ds_write
#ifdef CHD_HOST_EMU
typedef double LdsD;
typedef int LdsI;
#define CHD_SCHED_FENCE() ((void)0)
#define CHD_LOAD_T0 0
#else
typedef __attribute__((address_space(3))) double LdsD;
typedef __attribute__((address_space(3))) int LdsI;
#define CHD_SCHED_FENCE() __builtin_amdgcn_sched_barrier(0)
#define CHD_LOAD_T0 64         // threads below this one factor the diagonal block while the rest load the panel
#endif
#ifdef CHD_HOST_EMU
#define CHD_WAVE_ID 0
#define CHD_NWAVES 1
#define CHD_LANE 0
#define CHD_WAVE_SZ 1
#else
#define CHD_WAVE_ID ((int)(threadIdx.x >> 6))
#define CHD_NWAVES ((int)(blockDim.x >> 6))
#define CHD_LANE ((int)(threadIdx.x & 63))
#define CHD_WAVE_SZ 64
#endif
#define PAR_FOR(i, n) for (int i = CHD_TID; i < (n); i += CHD_NT)
// the same loop shared by the threads t0 .. CHD_NT - 1 only (threads below t0 skip it)
#define PAR_FOR_FROM(i, n, t0) for (int i = CHD_TID - (t0); i >= 0 && i < (n); i += CHD_NT - (t0))
// one group of CHD_GL consecutive lanes per item; `lane_` is the lane inside the group
#define GROUP_FOR(i, n) for (int i = CHD_TID / CHD_GL, lane_ = CHD_TID % CHD_GL; i < (n); i += CHD_NT / CHD_GL)

namespace chd {

// ---- solver constants (IPOPT defaults unless noted) ----------------------
#define CHD_MU_INIT_COLD 0.1
#define CHD_MU_INIT_WARM 1e-3
#define CHD_DELTA_W0 1e-4
#define CHD_DELTA_W_MIN 1e-9
#define CHD_DELTA_W_MAX 1e8
// Inertia handling.  The factorisation is L D L^T without pivoting in a fixed elimination order; every pivot is compared with the sign EXPECTED AT ITS
// POSITION (+ at a variable, - at a constraint row).  A pivot of the wrong sign (or below 1e-14 in magnitude) is replaced by +-1e-10 so that the
// factorisation can finish, and the attempt counts as failed (second model, then more damping; solve_stage) without a solve or trial evaluations -- the
// role of IPOPT's inertia correction.  The test is stricter than the inertia (n, m, 0): a negative pivot at a variable's position can be balanced by a positive
// one at a row's.  Measured on the fixture's 200 sequences + the kinematic optimisation's four hard clips (tests/tools/emu_sweep.py), the plain inertia count
// costs robustness: 3 failed stages instead of 1, slowest sequence 915 iterations instead of 231 -- steps from factorisations with pivots of the "wrong" sign
// are poor ones without pivoting.  Unlike the inertia, the positional test depends on the elimination order: the oracle uses the SAME order (band by node
// time, border variables by the start time of their phase: chd_model.hpp / ipm_solver.hpp), which is what keeps the two in lockstep.
#ifndef CHD_INERTIA_RETRY
#define CHD_INERTIA_RETRY 1
#endif
#define CHD_DELTA_C 1e-9
#define CHD_CONSTR_VIOL_TOL 1e-4
#define CHD_MAX_BACKTRACK 3
#define CHD_DUAL_RISE_K 6
#define CHD_DW_GROW_SECOND 1.5
#define CHD_BLOCKED_BORDER_MIN 256
#define CHD_MAX_ATTEMPTS 12
#ifndef CHD_ABORT_BAD_FACTOR
#define CHD_ABORT_BAD_FACTOR 1
#endif
#ifndef CHD_TRAIL_TP
#define CHD_TRAIL_TP 1      // independent 16x16 tiles per wavefront pass of the trailing update (two passes are in flight).  Measured on one
                            // box, sequences/s of the bench workload: 1: 553, 2: 545, 3: 538, 4: 517, 6: 433 -- the update is inlined into the
                            // panel loop, and what more tiles per pass buy in loads in flight they cost in registers spilled there
#endif
#define CHD_G 9.80665
#define CHD_MU_FRICTION 0.5
#define CHD_INF 1e19

enum { EV_VALUES = 0, EV_FULL = 1 };
// n-sized vectors
enum { VN_X = 0, VN_G, VN_DUALX, VN_DX, VN_XT, VN_XS, VN_COUNT };
// m-sized vectors
enum { VM_C = 0, VM_S, VM_ZL, VM_ZU, VM_LAM, VM_L, VM_U, VM_SC, VM_SIGMA, VM_RS, VM_D, VM_R, VM_DLAM, VM_DS, VM_DZL, VM_DZU,
       VM_ST, VM_CT, VM_RT, VM_SS2, VM_COUNT };
// N-sized vectors
enum { VK_RHS = 0, VK_SOL, VK_RHS2, VK_SOL2, VK_T1, VK_T2, VK_DIAG, VK_Y, VK_HIST, VK_COUNT };
// row flags
enum { RF_EQ = 1, RF_L = 2, RF_U = 4 };

// The sequence descriptor and the solver context are workgroup-uniform: both live in LDS (one copy per workgroup, filled
// when the workgroup takes a sequence from the queue) and are addressed through local-address-space pointers, so a
// field access is a ds_read -- a generic `const SeqDesc*` compiles to flat_load, a context on the stack to scratch
// traffic in every noinline phase.
#ifdef CHD_HOST_EMU
typedef const SeqDesc* QP;
typedef const StageDesc* SDP;
#else
typedef const __attribute__((address_space(3))) SeqDesc* QP;
typedef const __attribute__((address_space(3))) StageDesc* SDP;
#endif
struct Ctx {
  QP q;
  SDP S;
  LdsD* lds;            // workgroup scratch (LDS on the device)
  int lds_cap;          // doubles available in lds
  int n, m, N, Nb, bc, w, W2, LD;
  GD* K0b; GD* K0x; GD* Kfb; GD* Kfx;
  GU* pmb; GU* pmx; GU* pmt;      // occupancy masks of K0 (band rows, border rows, border transposed): one bit per stored entry that has ever been written in this stage
  int MW, LW, CW;                 // 64-bit words per band row / border row / band column of the three masks
  GI* csr_rp; GI* csr_col; GI* csr_row;      // the marked entries as a row-sorted list, rebuilt from the masks when an evaluation has added bits
  int csr_nnz, csr_cap;
  int pm_dirty;                   // an entry was marked since the list was built (any thread sets it; read after a barrier)
  int side_n;                     // entries of the side list (below) filled by the last evaluation with exact blocks; any thread increments it
  int side_ok;                    // 1: the side list describes the difference between the K0 that is stored now (first model) and the second model of the same point
  const GI* pos_var; const GI* pos_row;
  GI* env;              // [2p] first, [2p+1] last band position coupled to p (envelope): the stage's working copy
  GI* rcnt;             // [k]: the border rows [0, rcnt[k]) are the ones that can reach band column k (working copy)
  double sf;            // objective scaling
  double tol;           // IPOPT tol of the stage (phys_optim.cpp:578)
  int stall_window;     // 0 = no stall guard (chd_config.stall_window)
  int second_model;     // 1 while the second model of an iteration is built: Gauss-Newton, without the exact constraint-curvature blocks (solve_stage)
  int err;              // sticky error flag (band overflow): any thread may set it, read after a barrier
  int n_bad_pivots;     // pivots that did not have the sign expected at their position and were replaced (thread 0 counts)
  long long tacc[24];    // cycles per phase (thread 0): 0 eval full, 1 eval values, 2 factor, 3 solve, 4 matvec, 5 total
};
#ifdef CHD_HOST_EMU
typedef Ctx LCtx;
#define CHD_CLOCK() 0LL
#else
typedef __attribute__((address_space(3))) Ctx LCtx;
#define CHD_CLOCK() ((long long)wall_clock64())
#endif
#define TACC(c, k, v) do { if (CHD_TID == 0) (c).tacc[k] += (v); } while (0)
#ifdef CHD_EVAL_TIMING      // study build: the substitution's sub-phase slots 16..20 time parts of the evaluation instead (with a barrier in front of every reading)
#define EVT_BEGIN() CHD_SYNC(); long long evt_ = CHD_CLOCK()
#define EVT(c, k) do { CHD_SYNC(); TACC(c, k, CHD_CLOCK() - evt_); evt_ = CHD_CLOCK(); } while (0)
#else
#define EVT_BEGIN() ((void)0)
#define EVT(c, k) ((void)0)
#endif
#define TIC() const long long tic_ = CHD_CLOCK()
#define TOC(c, k) TACC(c, k, CHD_CLOCK() - tic_)

#define VN(c, k) ((c).q->wd + (c).q->o_vec_n + (long long)(k) * (c).q->max_n)
#define VM(c, k) ((c).q->wd + (c).q->o_vec_m + (long long)(k) * (c).q->max_m)
#define VK(c, k) ((c).q->wd + (c).q->o_vec_N + (long long)(k) * (c).q->max_N)

// ------------------------------------------------------------------------------------------
// workgroup reductions (deterministic: fixed tree, every thread combines the per-wave partials
// in the same order)
// ------------------------------------------------------------------------------------------
#ifdef CHD_HOST_EMU
CHD_DEV double block_sum(LCtx&, double v) { return v; }
CHD_DEV double block_max(LCtx&, double v) { return v; }
CHD_DEV double block_min(LCtx&, double v) { return v; }
CHD_DEV double group_sum(double v) { return v; }
#else
CHD_DEV double block_sum(LCtx& c, double v) {
  for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) c.lds[threadIdx.x >> 6] = v;
  __syncthreads();
  double t = 0; const int nw = (blockDim.x + 63) >> 6;
  for (int k = 0; k < nw; ++k) t += c.lds[k];
  return t;
}
CHD_DEV double block_max(LCtx& c, double v) {
  for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_down(v, o));
  __syncthreads();
  if ((threadIdx.x & 63) == 0) c.lds[threadIdx.x >> 6] = v;
  __syncthreads();
  double t = c.lds[0]; const int nw = (blockDim.x + 63) >> 6;
  for (int k = 1; k < nw; ++k) t = fmax(t, c.lds[k]);
  return t;
}
CHD_DEV double block_min(LCtx& c, double v) { return -block_max(c, -v); }
CHD_DEV double group_sum(double v) {
  v += __shfl_xor(v, 8); v += __shfl_xor(v, 4); v += __shfl_xor(v, 2); v += __shfl_xor(v, 1);
  return v;
}
#endif
#ifdef CHD_HOST_EMU
#define CHD_PAIR 1
CHD_DEV double pair_sum(double v) { return v; }
#else
#define CHD_PAIR 2
CHD_DEV double pair_sum(double v) { return v + __shfl_xor(v, 32); }
#endif
#define LDS_RED 64     // doubles reserved at the start of lds for the reductions

// ------------------------------------------------------------------------------------------
// Cubic Hermite splines (TOWR CubicHermitePolynomial / NodeSpline)
// ------------------------------------------------------------------------------------------
// Cumulative end times of the polynomials / phases, staged in LDS by refresh_durations: every spline evaluation starts with a binary
// search over them -- five dependent loads, ten searches per dynamics unit -- which from HBM / L2 was a chain of ~50 round trips
// per thread in the row phases.  (Larger problems than the tables hold keep searching the workspace copy.)
#define CHD_PEND_CAP 640
#define CHD_PHEND_CAP 64
#ifdef CHD_HOST_EMU
static thread_local double chd_pend_l[CHD_PEND_CAP], chd_phend_l[CHD_PHEND_CAP];      // (per emulated workgroup = per host thread: the sanitizer builds run several at once)
static thread_local int chd_tab_ok = 0;
#else
__shared__ double chd_pend_l[CHD_PEND_CAP];
__shared__ double chd_phend_l[CHD_PHEND_CAP];
__shared__ int chd_tab_ok;
#endif

struct PE {
  int poly;
  double tl, T;
  double p[3], v[3], a[3];
  double w[3][4];     // d{pos,vel,acc}/d(p0, v0, p1, v1) of the active polynomial
};

template <class CP>
CHD_DEV int seg_lookup(CP cum_end, int n, double t) {    // first i with cum_end[i] >= t - 1e-10, clamped
  int lo = 0, hi = n - 1;
  const double tt = t - 1e-10;
  while (lo < hi) { int mid = (lo + hi) >> 1; if (cum_end[mid] >= tt) hi = mid; else lo = mid + 1; }
  return lo;
}

CHD_DEV void hermite_eval(QP q, int s, int id, double tl, PE& e) {
  const auto& sp = q->sp[s];
  const double T = q->wd[q->o_poly_dur + sp.poly_off + id];
  e.poly = id; e.tl = tl; e.T = T;
  const double t = tl, t2 = t * t, t3 = t2 * t, iT = 1.0 / T, iT2 = iT * iT, iT3 = iT2 * iT;
  e.w[0][0] = 2 * t3 * iT3 - 3 * t2 * iT2 + 1;   e.w[0][1] = t - 2 * t2 * iT + t3 * iT2;
  e.w[0][2] = 3 * t2 * iT2 - 2 * t3 * iT3;       e.w[0][3] = t3 * iT2 - t2 * iT;
  e.w[1][0] = 6 * t2 * iT3 - 6 * t * iT2;        e.w[1][1] = 3 * t2 * iT2 - 4 * t * iT + 1;
  e.w[1][2] = 6 * t * iT2 - 6 * t2 * iT3;        e.w[1][3] = 3 * t2 * iT2 - 2 * t * iT;
  e.w[2][0] = 12 * t * iT3 - 6 * iT2;            e.w[2][1] = 6 * t * iT2 - 4 * iT;
  e.w[2][2] = 6 * iT2 - 12 * t * iT3;            e.w[2][3] = 6 * t * iT2 - 2 * iT;
  const GD* nv = q->wd + q->o_node + sp.node_off + id * 6;   // [node id: p xyz, v xyz][node id+1: ...]
  for (int k = 0; k < 3; ++k) {
    const double p0 = nv[k], v0 = nv[3 + k], p1 = nv[6 + k], v1 = nv[9 + k];
    e.p[k] = e.w[0][0] * p0 + e.w[0][1] * v0 + e.w[0][2] * p1 + e.w[0][3] * v1;
    e.v[k] = e.w[1][0] * p0 + e.w[1][1] * v0 + e.w[1][2] * p1 + e.w[1][3] * v1;
    e.a[k] = e.w[2][0] * p0 + e.w[2][1] * v0 + e.w[2][2] * p1 + e.w[2][3] * v1;
  }
}

CHD_DEV void spline_eval(QP q, int s, double tg, PE& e) {
  const auto& sp = q->sp[s];
  if (chd_tab_ok) {
    const LdsD* pend = (const LdsD*)chd_pend_l + sp.poly_off;
    const int id = seg_lookup(pend, sp.n_polys, tg);
    hermite_eval(q, s, id, tg - (id > 0 ? pend[id - 1] : 0.0), e);
    return;
  }
  const GD* pend = q->wd + q->o_pend + sp.poly_off;
  const int id = seg_lookup(pend, sp.n_polys, tg);
  hermite_eval(q, s, id, tg - (id > 0 ? pend[id - 1] : 0.0), e);
}

// d p(t) / d T_poly for the active polynomial (CubicHermitePolynomial::GetDerivativeOfPosWrtDuration)
CHD_DEV void dpos_dT(QP q, int s, const PE& e, double* out) {
  const auto& sp = q->sp[s];
  const GD* nv = q->wd + q->o_node + sp.node_off + e.poly * 6;
  const double t = e.tl, t2 = t * t, t3 = t2 * t, T = e.T, iT = 1.0 / T, iT2 = iT * iT, iT3 = iT2 * iT, iT4 = iT2 * iT2;
  for (int k = 0; k < 3; ++k) {
    const double x0 = nv[k], v0 = nv[3 + k], x1 = nv[6 + k], v1 = nv[9 + k];
    out[k] = t3 * (v0 + v1) * iT3 - t2 * (2 * v0 + v1) * iT2 - 3 * t3 * (2 * x0 - 2 * x1 + T * v0 + T * v1) * iT4 +
             2 * t2 * (3 * x0 - 3 * x1 + 2 * T * v0 + T * v1) * iT3;
  }
}

// The samples of a uniform time table (tab[k] = k * step, then T; SeqModel::sample_times) that can lie in the polynomials pa .. pb of spline s at the current
// durations: [k_lo, k_hi], a superset by a sample on either side -- the loops that gather per-sample records by polynomial keep their own test and only start
// and stop closer
CHD_DEV void sample_range(QP q, int s, int pa, int pb, const GD* tab, int len, int& k_lo, int& k_hi) {
  const auto& sp = q->sp[s];
  double t_lo, t_hi;
  if (chd_tab_ok) { const LdsD* pend = (const LdsD*)chd_pend_l + sp.poly_off; t_lo = pa > 0 ? pend[pa - 1] : 0.0; t_hi = pend[pb]; }
  else { const GD* pend = q->wd + q->o_pend + sp.poly_off; t_lo = pa > 0 ? pend[pa - 1] : 0.0; t_hi = pend[pb]; }
  const double step = tab[1] - tab[0];
  const double idt = step > 0.0 ? 1.0 / step : 0.0;
  k_lo = (int)(t_lo * idt) - 1; if (k_lo < 0) k_lo = 0;
  k_hi = step > 0.0 ? (int)(t_hi * idt) + 2 : len - 1; if (k_hi > len - 1) k_hi = len - 1;
}
// phase of end-effector e at time t, and whether it is the last one
CHD_DEV int phase_lookup(QP q, int e, double t) {
  if (chd_tab_ok) return seg_lookup((const LdsD*)chd_phend_l + q->phase_off[e], q->n_phase[e], t);
  return seg_lookup(q->wd + q->o_phend + q->phase_off[e], q->n_phase[e], t);
}

// d p(t) / d(phase durations) for a phase-based spline (PhaseSpline::GetJacobianOfPosWrtDurations,
// PhaseDurations::GetJacobianOfPos).  Returns cur phase; dj_k[d] for k < cur is `early[d]`,
// for k == cur (when not last) is `own[d]`; zero beyond.
struct DurJac { int cur, last, nvar; double own[3], early[3]; };
CHD_DEV void dur_jac(QP q, int s, double t, const PE& e, DurJac& dj) {
  const auto& sp = q->sp[s];
  const int ee = sp.ee;
  const GI* pi = q->ci + q->o_pinfo + (sp.poly_off + e.poly) * 4;
  double dT[3];
  dpos_dT(q, s, e, dT);
  const double inv = 1.0 / pi[2];
  dj.cur = phase_lookup(q, ee, t);
  dj.last = dj.cur == q->n_phase[ee] - 1;
  dj.nvar = q->n_phase[ee] - 1;
  for (int k = 0; k < 3; ++k) {
    const double dx = inv * (dT[k] - pi[1] * e.v[k]);
    dj.own[k] = dj.last ? 0.0 : dx;
    dj.early[k] = -e.v[k] - (dj.last ? dx : 0.0);
  }
}

// First AND second derivatives of p(t) with respect to the phase durations of a phase-based spline.
// With tau = local time in the active polynomial and Tp its duration, both are affine in the duration
// variables T_k:  d tau/dT_k = u_k, d Tp/dT_k = v_k, where (u, v) only depends on whether k is an
// earlier phase ("e") or the current one ("c"):
//     current phase not the last:  e: (-1, 0)            c: (-k_in/n, 1/n)
//     current phase is the last :  e: (-1 + k_in/n, -1/n)   (the last duration is T - sum of the variables)
//   dp/dT_k       = G_x     = h_tau u_x + h_T v_x
//   d2p/dT_k dT_l = Q_xy    = h_tautau u_x u_y + h_tauT (u_x v_y + v_x u_y) + h_TT v_x v_y
// (the reference only needs first derivatives because IPOPT runs with an L-BFGS Hessian, phys_optim.cpp:572;
//  this solver uses the exact duration block of the Lagrangian Hessian instead — DESIGN.md)
// Mixed derivatives: p = sum_j w_j(tau, Tp) x_j over the polynomial's four Hermite coefficients, so d2p / d x_j d T_k is the derivative of the weight,
//   om_x,j = (d w_j / d tau) u_x + (d w_j / d Tp) v_x  (d w_j / d tau = the velocity weight).
struct DurJac2 { int cur, last, nvar; double Ge[3], Gc[3], Qee[3], Qec[3], Qcc[3], ome[4], omc[4]; };
CHD_DEV void dur_jac2(QP q, int s, double t, const PE& e, DurJac2& dj) {
  const auto& sp = q->sp[s];
  const int ee = sp.ee;
  const GI* pi = q->ci + q->o_pinfo + (sp.poly_off + e.poly) * 4;
  const GD* nv = q->wd + q->o_node + sp.node_off + e.poly * 6;
  const double kin = pi[1], inv = 1.0 / pi[2];
  dj.cur = phase_lookup(q, ee, t);
  dj.last = dj.cur == q->n_phase[ee] - 1;
  dj.nvar = q->n_phase[ee] - 1;
  const double ue = dj.last ? -1.0 + kin * inv : -1.0, ve = dj.last ? -inv : 0.0;
  const double uc = -kin * inv, vc = inv;
  const double tau = e.tl, tau2 = tau * tau, tau3 = tau2 * tau, T = e.T, iT = 1.0 / T, iT2 = iT * iT, iT3 = iT2 * iT, iT4 = iT2 * iT2, iT5 = iT4 * iT;
  for (int k = 0; k < 3; ++k) {
    const double dl = nv[k] - nv[6 + k], s2 = 2 * nv[3 + k] + nv[9 + k], s1 = nv[3 + k] + nv[9 + k];
    const double cT = 6 * dl * iT3 + s2 * iT2, dT = -6 * dl * iT4 - 2 * s1 * iT3;
    const double cTT = -18 * dl * iT4 - 2 * s2 * iT3, dTT = 24 * dl * iT5 + 6 * s1 * iT4;
    const double hT = cT * tau2 + dT * tau3, htT = 2 * cT * tau + 3 * dT * tau2, hTT = cTT * tau2 + dTT * tau3;
    const double ht = e.v[k], htt = e.a[k];
    dj.Ge[k] = ht * ue + hT * ve;
    dj.Gc[k] = dj.last ? 0.0 : ht * uc + hT * vc;
    dj.Qee[k] = htt * ue * ue + 2 * htT * ue * ve + hTT * ve * ve;
    dj.Qec[k] = dj.last ? 0.0 : htt * ue * uc + htT * (ue * vc + ve * uc) + hTT * ve * vc;
    dj.Qcc[k] = dj.last ? 0.0 : htt * uc * uc + 2 * htT * uc * vc + hTT * vc * vc;
  }
  const double wT[4] = {-6 * tau3 * iT4 + 6 * tau2 * iT3, 2 * tau2 * iT2 - 2 * tau3 * iT3, -6 * tau2 * iT3 + 6 * tau3 * iT4, -2 * tau3 * iT3 + tau2 * iT2};
  for (int j = 0; j < 4; ++j) { dj.ome[j] = e.w[1][j] * ue + wT[j] * ve; dj.omc[j] = dj.last ? 0.0 : e.w[1][j] * uc + wT[j] * vc; }
}
// one record of the node x duration table (chd_device.hpp, XR_*): block `blk` of duration end-effector `ee`, sample `smp`
CHD_DEV GD* xrec(QP q, int ee, int blk, int smp) {
  const int nd_ = q->n_tdyn, nr_ = q->n_trom;
  // block starts in the order XB_HEIGHT, ROM_M, HEEL_OWN, DYN_P, HEEL_X, DYN_F, ROM_C, DYN_C, ROM_A
  const int start = blk == XB_HEIGHT ? 0 : blk == XB_ROM_M ? nd_ : blk == XB_HEEL_OWN ? nd_ + nr_ : blk == XB_DYN_P ? nd_ + 2 * nr_ : blk == XB_HEEL_X ? 2 * nd_ + 2 * nr_
                  : blk == XB_DYN_F ? 2 * nd_ + 3 * nr_ : blk == XB_ROM_C ? 3 * nd_ + 3 * nr_ : blk == XB_DYN_C ? 3 * nd_ + 4 * nr_ : 4 * nd_ + 4 * nr_;
  return q->wd + q->o_xtab + ((long long)ee * (4 * nd_ + 5 * nr_) + start + smp) * XR_STRIDE;
}
CHD_DEV int xblock_times(QP q, int blk) { return (blk == XB_HEIGHT || blk == XB_DYN_P || blk == XB_DYN_F || blk == XB_DYN_C) ? q->o_tdyn : q->o_trom; }      // the block's sample times (offset in cd)
CHD_DEV int xblock_len(QP q, int blk) { return (blk == XB_HEIGHT || blk == XB_DYN_P || blk == XB_DYN_F || blk == XB_DYN_C) ? q->n_tdyn : q->n_trom; }
// (pe: the sample of the spline whose node values the entries differentiate by; dj: the duration derivatives of the end-effector's own spline at the sample --
//  only its cur / last and, when B is given, its om's are used; Ae / Ac / B may be null = zero)
CHD_DEV void xrec_store(GD* r, const PE& pe, const DurJac2& dj, const double* Ae, const double* Ac, const double* B) {
  r[XR_POLY] = pe.poly; r[XR_CUR] = dj.cur; r[XR_LAST] = dj.last;
  for (int j = 0; j < 4; ++j) { r[XR_W + j] = pe.w[0][j]; r[XR_OME + j] = B ? dj.ome[j] : 0.0; r[XR_OMC + j] = B ? dj.omc[j] : 0.0; }
  for (int k = 0; k < 3; ++k) { r[XR_AE + k] = Ae ? Ae[k] : 0.0; r[XR_AC + k] = Ac ? Ac[k] : 0.0; r[XR_B + k] = B ? B[k] : 0.0; }
}
CHD_DEV double dot3(const double* a, const double* b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
CHD_DEV void d2_store(QP q, int ee, int slot, int cur, double see, double sec, double scc) {
  GD* t = q->wd + q->o_d2tab + ((long long)ee * q->d2_slots + slot) * D2_STRIDE;
  t[0] = cur; t[1] = see; t[2] = sec; t[3] = scc;
}
// class of the pair (k, l), l <= k, for a sample whose current phase is cur: 1 = S_ee, 2 = S_ec, 3 = S_cc, 0 = none
CHD_DEV double d2_select(const double* t, int k, int l) {
  const int cur = (int)t[0];
  if (k < cur) return t[1];
  if (k == cur) return l < cur ? t[2] : t[3];
  return 0.0;
}

// ------------------------------------------------------------------------------------------
// Euler angles (TOWR EulerConverter, ZYX), hand-derived derivatives
// ------------------------------------------------------------------------------------------
CHD_DEV void mat3_mul(const double A[3][3], const double B[3][3], double C[3][3]) {
  for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) C[i][j] = A[i][0] * B[0][j] + A[i][1] * B[1][j] + A[i][2] * B[2][j];
}
// R = Rz(z) Ry(y) Rx(x) and dR[k] = dR/d e_k
CHD_DEV void rot_and_derivs(const double e[3], double R[3][3], double dR[3][3][3]) {
  const double cx = cos(e[0]), sx = sin(e[0]), cy = cos(e[1]), sy = sin(e[1]), cz = cos(e[2]), sz = sin(e[2]);
  const double Rx[3][3] = {{1, 0, 0}, {0, cx, -sx}, {0, sx, cx}}, dRx[3][3] = {{0, 0, 0}, {0, -sx, -cx}, {0, cx, -sx}};
  const double Ry[3][3] = {{cy, 0, sy}, {0, 1, 0}, {-sy, 0, cy}}, dRy[3][3] = {{-sy, 0, cy}, {0, 0, 0}, {-cy, 0, -sy}};
  const double Rz[3][3] = {{cz, -sz, 0}, {sz, cz, 0}, {0, 0, 1}}, dRz[3][3] = {{-sz, -cz, 0}, {cz, -sz, 0}, {0, 0, 0}};
  double ZY[3][3], t[3][3];
  mat3_mul(Rz, Ry, ZY); mat3_mul(ZY, Rx, R);
  mat3_mul(ZY, dRx, dR[0]);
  mat3_mul(Rz, dRy, t); mat3_mul(t, Rx, dR[1]);
  mat3_mul(dRz, Ry, t); mat3_mul(t, Rx, dR[2]);
}
CHD_DEV void cross3(const double a[3], const double b[3], double o[3]) {
  o[0] = a[1] * b[2] - a[2] * b[1]; o[1] = a[2] * b[0] - a[0] * b[2]; o[2] = a[0] * b[1] - a[1] * b[0];
}
CHD_DEV void matvec3(const double A[3][3], const double v[3], double o[3]) {
  for (int i = 0; i < 3; ++i) o[i] = A[i][0] * v[0] + A[i][1] * v[1] + A[i][2] * v[2];
}
// selections by a run-time row number i in {0, 1, 2}: compare-and-select keeps small vectors in registers (indexing a
// local array with i puts it in scratch memory)
CHD_DEV double pick3(const double a, const double b, const double cc, const int i) { return i == 0 ? a : (i == 1 ? b : cc); }
// coefficients on u of (v x u)_i
CHD_DEV void cross_row(const double v[3], const int i, double o[3]) {
  o[0] = i == 0 ? 0.0 : (i == 1 ? v[2] : -v[1]);
  o[1] = i == 0 ? -v[2] : (i == 1 ? 0.0 : v[0]);
  o[2] = i == 0 ? v[1] : (i == 1 ? -v[0] : 0.0);
}

// Angular part of the centroidal dynamics (humanoid_rigid_body_dynamics.cpp:89-115):
//   ang = I_w wd + w x (I_w w),  I_w = R I_b R^T,  w = M(e) e',  wd = Md(e,e') e' + M(e) e''.
// Outputs ang[3] and its partials d0 (wrt e), d1 (wrt e'), d2 (wrt e''): dX[i][k] = d ang_i / d (.)_k.
CHD_DEV void angular_term(const double e[3], const double ed[3], const double edd[3], const double Ib[3][3], int want_jac,
                          double ang[3], double d0[3][3], double d1[3][3], double d2[3][3]) {
  double R[3][3], dR[3][3][3];
  rot_and_derivs(e, R, dR);
  const double cy = cos(e[1]), sy = sin(e[1]), cz = cos(e[2]), sz = sin(e[2]);
  const double M[3][3] = {{cy * cz, -sz, 0}, {cy * sz, cz, 0}, {-sy, 0, 1}};
  // dM/dy, dM/dz (dM/dx = 0) and the second derivatives
  const double My[3][3] = {{-sy * cz, 0, 0}, {-sy * sz, 0, 0}, {-cy, 0, 0}};
  const double Mz[3][3] = {{-cy * sz, -cz, 0}, {cy * cz, -sz, 0}, {0, 0, 0}};
  const double Myy[3][3] = {{-cy * cz, 0, 0}, {-cy * sz, 0, 0}, {sy, 0, 0}};
  const double Myz[3][3] = {{sy * sz, 0, 0}, {-sy * cz, 0, 0}, {0, 0, 0}};
  const double Mzz[3][3] = {{-cy * cz, sz, 0}, {-cy * sz, -cz, 0}, {0, 0, 0}};
  double Md[3][3];
  for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) Md[i][j] = My[i][j] * ed[1] + Mz[i][j] * ed[2];
  double om[3], omd[3], t1[3], t2[3];
  matvec3(M, ed, om);
  matvec3(Md, ed, t1); matvec3(M, edd, t2);
  for (int i = 0; i < 3; ++i) omd[i] = t1[i] + t2[i];
  double RI[3][3], Rt[3][3], Iw[3][3];
  mat3_mul(R, Ib, RI);
  for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) Rt[i][j] = R[j][i];
  mat3_mul(RI, Rt, Iw);
  double Iw_om[3], Iw_omd[3], cr[3];
  matvec3(Iw, om, Iw_om); matvec3(Iw, omd, Iw_omd);
  cross3(om, Iw_om, cr);
  for (int i = 0; i < 3; ++i) ang[i] = Iw_omd[i] + cr[i];
  if (!want_jac) return;
  for (int k = 0; k < 3; ++k) {
    // ---- wrt e''_k : I_w M[:,k]
    double mk[3] = {M[0][k], M[1][k], M[2][k]}, o[3];
    matvec3(Iw, mk, o);
    for (int i = 0; i < 3; ++i) d2[i][k] = o[i];
    // ---- wrt e'_k : d om = M[:,k] ; d omd = Md[:,k] + (dM/de_k) e'
    double domd[3], dMk_ed[3] = {0, 0, 0};
    if (k == 1) matvec3(My, ed, dMk_ed);
    if (k == 2) matvec3(Mz, ed, dMk_ed);
    for (int i = 0; i < 3; ++i) domd[i] = Md[i][k] + dMk_ed[i];
    double a1[3], a2[3], a3[3], Iw_mk[3];
    matvec3(Iw, domd, a1);
    cross3(mk, Iw_om, a2);
    matvec3(Iw, mk, Iw_mk); cross3(om, Iw_mk, a3);
    for (int i = 0; i < 3; ++i) d1[i][k] = a1[i] + a2[i] + a3[i];
    // ---- wrt e_k
    double dIw[3][3], A[3][3], B[3][3], dRt[3][3], IbRt[3][3];
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) dRt[i][j] = dR[k][j][i];
    mat3_mul(Ib, Rt, IbRt);
    mat3_mul(dR[k], IbRt, A);          // dR Ib R^T
    mat3_mul(RI, dRt, B);              // R Ib dR^T
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) dIw[i][j] = A[i][j] + B[i][j];
    double dom[3] = {0, 0, 0}, domd_e[3] = {0, 0, 0};
    if (k == 1) {
      matvec3(My, ed, dom);
      double Mdk[3][3];
      for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) Mdk[i][j] = Myy[i][j] * ed[1] + Myz[i][j] * ed[2];
      double u1[3], u2[3]; matvec3(Mdk, ed, u1); matvec3(My, edd, u2);
      for (int i = 0; i < 3; ++i) domd_e[i] = u1[i] + u2[i];
    } else if (k == 2) {
      matvec3(Mz, ed, dom);
      double Mdk[3][3];
      for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) Mdk[i][j] = Myz[i][j] * ed[1] + Mzz[i][j] * ed[2];
      double u1[3], u2[3]; matvec3(Mdk, ed, u1); matvec3(Mz, edd, u2);
      for (int i = 0; i < 3; ++i) domd_e[i] = u1[i] + u2[i];
    }
    double b1[3], b2[3], b3[3], b4[3], b5[3], tmp[3];
    matvec3(dIw, omd, b1);
    matvec3(Iw, domd_e, b2);
    cross3(dom, Iw_om, b3);
    matvec3(dIw, om, tmp); cross3(om, tmp, b4);
    matvec3(Iw, dom, tmp); cross3(om, tmp, b5);
    for (int i = 0; i < 3; ++i) d0[i][k] = b1[i] + b2[i] + b3[i] + b4[i] + b5[i];
  }
}

// ------------------------------------------------------------------------------------------
// KKT storage.  K = [H + dw Dw, J^T; J, -D] in a time ordering: positions < Nb form a band of
// half-width w, the last bc positions (shared stance positions, durations, rows touching only
// those) form a dense border.  K0 (unfactored, both triangles): band rows Nb x (2w+1), border
// rows bc x N.  Kf (factor, lower): band rows Nb x (w+1), border rows bc x N.
// ------------------------------------------------------------------------------------------
// The envelope comes from the structure at the durations the stage starts with (plus head-room); in stage 3 the
// durations can carry a sample into a polynomial that structure did not foresee.  An entry (row hi, column lo < hi)
// left of row hi's envelope start widens the working copy: the row starts at lo, and the columns in between are
// reached by row hi.  (min / max updates: the result does not depend on the order the threads arrive in.)
//
// Occupancy.  The band envelope holds ~18 x as many entries as the matrix has non-zeros (537 k against 30 k for a 90-frame walk; the border: 79 k
// against 1.5 k), so everything that READS K0 -- the two KKT products of an iteration, the copy into the factor, clearing it for the next evaluation -- goes
// through bit masks of the entries that have ever been written in this stage: one bit per stored entry of a band row (pmb), of a border row (pmx), and
// the border transposed (pmt: per band column, the border rows with an entry in it).  The masks maintain themselves at no cost in traffic: kzero
// stores -0.0 in the marked entries and everything else is +0.0 since kreset, so a writer whose read-modify-write finds the bit pattern of +0.0 knows the
// entry may be unmarked and sets its bits (atomic OR: idempotent, order independent).  -0.0 + v == v for every v != 0, so values are unchanged; an entry that
// cancels to +0.0 exactly is marked again by its next writer, harmlessly.  In the duration stage the pattern moves with the iterate (a sample changes
// polynomial): bits are only ever added within a stage.  The readers do not walk the masks themselves (a dependent load and a divergent bit loop per word)
// but a row-sorted list of the marked entries, (column, row) pairs + row starts, rebuilt from the masks after an evaluation that added bits
// (csr_rebuild: the first evaluation of a stage, and in the duration stage whenever the pattern moved).  A band row's list holds its band entries and
// then the border rows coupled to its column, so one pass gives the whole product.
#ifdef CHD_HOST_EMU
CHD_DEV void env_min(GI* p, int v) { if (v < *p) *p = v; }
CHD_DEV void env_max(GI* p, int v) { if (v > *p) *p = v; }
#else
CHD_DEV void env_min(GI* p, int v) { __hip_atomic_fetch_min(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
CHD_DEV void env_max(GI* p, int v) { __hip_atomic_fetch_max(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
#endif
CHD_DEV void env_cover(LCtx& c, const int hi, const int lo) {
  const int first = c.env[2 * hi];
  if (lo >= first) return;
  env_min(c.env + 2 * hi, lo);
  if (hi < c.Nb) for (int k = lo; k < first && k < hi; ++k) env_max(c.env + 2 * k + 1, hi);
  else for (int k = lo; k < first && k < c.Nb; ++k) env_max(c.rcnt + k, hi - c.Nb + 1);      // border row hi - Nb now reaches columns lo ..
}
#ifdef CHD_HOST_EMU
CHD_DEV unsigned long long dbits(double v) { unsigned long long u; __builtin_memcpy(&u, &v, 8); return u; }
CHD_DEV bool mask_or(GU* p, unsigned long long b) { const bool was = (*p & b) != 0; *p |= b; return !was; }      // true: the bit is new
#else
CHD_DEV unsigned long long dbits(double v) { return (unsigned long long)__double_as_longlong(v); }
CHD_DEV bool mask_or(GU* p, unsigned long long b) { return (__hip_atomic_fetch_or(p, b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) & b) == 0; }
#endif
// the entry stored at `p` (inside K0b or K0x) becomes a marked one
CHD_NOINLINE CHD_DEV void kmark(LCtx& c, const GD* p) {
  bool fresh;
  if (p < c.K0x) {
    const long long off = p - c.K0b;
    const int i = (int)(off / c.W2), o = (int)(off - (long long)i * c.W2);
    fresh = mask_or(c.pmb + (long long)i * c.MW + (o >> 6), 1ull << (o & 63));
  } else {
    const long long off = p - c.K0x;
    const int r = (int)(off / c.LD), k = (int)(off - (long long)r * c.LD);
    fresh = mask_or(c.pmx + (long long)r * c.LW + (k >> 6), 1ull << (k & 63));
    if (k < c.Nb) mask_or(c.pmt + (long long)k * c.CW + (r >> 6), 1ull << (r & 63));
  }
  if (fresh) c.pm_dirty = 1;      // (an entry of the list that cancelled to +0.0 exactly comes here too, and changes nothing)
}
// *p += val for a slot of K0 whose previous content `old` the caller has loaded
#define K0_ADD(c, p, old, val) do { const double o__ = (old), v__ = (val); if (v__ != 0.0) { if (dbits(o__) == 0ull) kmark(c, p); *(p) = o__ + v__; } } while (0)      // (a zero contribution must not turn the -0.0 marker into +0.0)
CHD_DEV void kadd(LCtx& c, int p, int qq, double val) {
  if (p < c.Nb && qq < c.Nb) {
    int dlt = qq - p;
    if (dlt > c.w || dlt < -c.w) { c.err = 1; return; }
    const int hi_ = dlt < 0 ? p : qq, lo_ = dlt < 0 ? qq : p;
    const int first_ = c.env[2 * hi_];                  // (looked up alongside the K0 accesses; the widening itself is rare)
    GD* a = c.K0b + (long long)p * c.W2 + (dlt + c.w);
    GD* b = c.K0b + (long long)qq * c.W2 + (c.w - dlt);
    const double oa = *a, ob = *b;
    K0_ADD(c, a, oa, val);
    if (dlt != 0) K0_ADD(c, b, ob, val);
    if (lo_ < first_) env_cover(c, hi_, lo_);
  } else {
    const int hi = p > qq ? p : qq, lo = p > qq ? qq : p;
    const int first_ = c.env[2 * hi];
    GD* a = c.K0x + (long long)(hi - c.Nb) * c.LD + lo;
    const bool two = lo >= c.Nb && lo != hi;
    GD* b = two ? c.K0x + (long long)(lo - c.Nb) * c.LD + hi : a;
    const double oa = *a, ob = *b;
    K0_ADD(c, a, oa, val);
    if (two) K0_ADD(c, b, ob, val);
    if (lo < c.Nb && lo < first_) env_cover(c, hi, lo);
  }
}
// the (up to two: both triangles are stored) locations of the KKT entry (p, qq); same checks as kadd
struct KSlot { GD* a; GD* b; };
CHD_DEV KSlot kslot(LCtx& c, int p, int qq) {
  KSlot sl; sl.a = nullptr; sl.b = nullptr;
  if (p < c.Nb && qq < c.Nb) {
    const int dlt = qq - p;
    if (dlt > c.w || dlt < -c.w) {
      c.err = 1; return sl; }
    const int hi_ = dlt < 0 ? p : qq, lo_ = dlt < 0 ? qq : p;
    if (lo_ < c.env[2 * hi_]) env_cover(c, hi_, lo_);
    sl.a = c.K0b + (long long)p * c.W2 + (dlt + c.w);
    if (dlt != 0) sl.b = c.K0b + (long long)qq * c.W2 + (c.w - dlt);
  } else {
    const int hi = p > qq ? p : qq, lo = p > qq ? qq : p;
    if (lo < c.Nb && lo < c.env[2 * hi]) env_cover(c, hi, lo);
    sl.a = c.K0x + (long long)(hi - c.Nb) * c.LD + lo;
    if (lo >= c.Nb && lo != hi) sl.b = c.K0x + (long long)(lo - c.Nb) * c.LD + hi;
  }
  return sl;
}
// N distinct KKT entries (p[i], q[i]) += val[i] (entries with p[i] < 0 are skipped): look-ups, loads and stores of all
// of them are issued together
template <int N>
CHD_DEV void kadd_batch(LCtx& c, const int* p, const int* qq, const double* val) {
  KSlot sl[N]; double oa[N], ob[N];
#pragma unroll
  for (int i = 0; i < N; ++i) { sl[i].a = nullptr; sl[i].b = nullptr; if (p[i] >= 0) sl[i] = kslot(c, p[i], qq[i]); }
#pragma unroll
  for (int i = 0; i < N; ++i) { oa[i] = *(sl[i].a ? sl[i].a : c.K0b); ob[i] = *(sl[i].b ? sl[i].b : c.K0b); }
#pragma unroll
  for (int i = 0; i < N; ++i) { if (sl[i].a) K0_ADD(c, sl[i].a, oa[i], val[i]); if (sl[i].b) K0_ADD(c, sl[i].b, ob[i], val[i]); }
}
// ---- The second model without a second evaluation (round 6).  The exact blocks of the Lagrangian Hessian enter K0 in eval_cost_grad_hess only, added to (or beside) the
// Gauss-Newton value of an entry by the ONE thread that owns it.  That thread also knows what the entry holds in the second model of the same point -- the Gauss-Newton
// value alone, bit for bit what a fresh evaluation with c.second_model = 1 computes, or nothing (-0.0, the cleared marker) -- and leaves it in a side list when the two differ.
// When the first model's factorisation meets a pivot of the wrong sign (17 % of the iterations), model_switch writes those values over the entries instead of evaluating
// splines, rows and costs again (0.9 ms each): c, g and f do not depend on the model.  An entry whose first-model value cancelled to zero exactly was never stored and
// cannot be switched: the evaluation then clears side_ok and the second model is evaluated the old way.
#ifdef CHD_HOST_EMU
CHD_DEV int side_take(LCtx& c) { return c.side_n++; }
#else
CHD_DEV int side_take(LCtx& c) { return __hip_atomic_fetch_add(&c.side_n, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
#endif
CHD_DEV void side_put(LCtx& c, const int p, const int qq, const double gn) {
  const int k = side_take(c);
  if (k < c.q->side_cap) {
    GI* pq = c.q->wi + c.q->o_side_pq;
    pq[2 * k] = p; pq[2 * k + 1] = qq; c.q->wd[c.q->o_side_v + k] = gn;
  }
}
CHD_DEV double side_gn(const double v) { return v != 0.0 ? v : -0.0; }      // what K0_ADD leaves in a cleared entry when it is handed v
CHD_DEV void model_switch(LCtx& c) {
  const GI* pq = c.q->wi + c.q->o_side_pq;
  const GD* sv = c.q->wd + c.q->o_side_v;
  PAR_FOR(e, c.side_n) {
    const KSlot sl = kslot(c, pq[2 * e], pq[2 * e + 1]);
    if (sl.a) *sl.a = sv[e];
    if (sl.b) *sl.b = sv[e];
  }
  CHD_SYNC();
}
CHD_DEV double kget(const LCtx& c, int p, int qq) {
  if (p < c.Nb && qq < c.Nb) {
    int dlt = qq - p;
    if (dlt > c.w || dlt < -c.w) return 0.0;
    return c.K0b[(long long)p * c.W2 + (dlt + c.w)];
  }
  const int hi = p > qq ? p : qq, lo = p > qq ? qq : p;
  return c.K0x[(long long)(hi - c.Nb) * c.LD + lo];
}

// where the entry (row, col) of the list lives in K0
CHD_DEV GD* k0_at(const LCtx& c, const int row, const int col) {
  if (row < c.Nb) return col < c.Nb ? c.K0b + (long long)row * c.W2 + (col - row + c.w) : c.K0x + (long long)(col - c.Nb) * c.LD + row;
  return c.K0x + (long long)(row - c.Nb) * c.LD + col;
}
// every marked entry <- -0.0 (the writers' "seen before" marker, see above): ~30 k scattered stores instead of the envelope's ~600 k
CHD_DEV void kzero(LCtx& c) {
  const double mz = -0.0;
  PAR_FOR(e, c.csr_nnz) *k0_at(c, c.csr_row[e], c.csr_col[e]) = mz;
}
CHD_DEV int popc64(unsigned long long m) { return __builtin_popcountll(m); }
// the list of marked entries from the masks: count per row, exclusive scan, fill
CHD_NOINLINE CHD_DEV void csr_rebuild(LCtx& c) {
  const int N = c.N, Nb = c.Nb, MW = c.MW, LW = c.LW, CW = c.CW, w = c.w;
#ifdef CHD_HOST_EMU
  if (std::getenv("CHD_EMU_TRACE_CSR")) std::fprintf(stderr, "csr_rebuild stage %d nnz before %d\n", c.S->stage, c.csr_nnz);
#endif
  PAR_FOR(i, N) {
    int cnt = 0;
    if (i < Nb) {
      for (int k = 0; k < MW; ++k) cnt += popc64(c.pmb[(long long)i * MW + k]);
      for (int k = 0; k < CW; ++k) cnt += popc64(c.pmt[(long long)i * CW + k]);
    } else for (int k = 0; k < LW; ++k) cnt += popc64(c.pmx[(long long)(i - Nb) * LW + k]);
    c.csr_rp[i] = cnt;
  }
  CHD_SYNC();
  // exclusive scan: every thread scans a contiguous chunk of rows, the chunk totals go through LDS
  LdsI* part = (LdsI*)(c.lds + LDS_RED);
  const int chunk = (N + CHD_NT - 1) / CHD_NT;
  const int r0 = CHD_TID * chunk < N ? CHD_TID * chunk : N, r1 = r0 + chunk < N ? r0 + chunk : N;
  int tot = 0;
  for (int i = r0; i < r1; ++i) tot += c.csr_rp[i];
  part[CHD_TID] = tot;
  CHD_SYNC();
  int off = 0, total = 0;
  for (int t = 0; t < CHD_NT; ++t) { const int v = part[t]; if (t < CHD_TID) off += v; total += v; }
  CHD_SYNC();
  if (total > c.csr_cap) {           // (does not happen for borders up to half dense: chd_model.hpp csr_cap)
    if (CHD_TID == 0) { c.err = 1; c.csr_nnz = 0; c.pm_dirty = 0; }
    PAR_FOR(i, N + 1) c.csr_rp[i] = 0;
    CHD_SYNC();
    return;
  }
  for (int i = r0; i < r1; ++i) { const int v = c.csr_rp[i]; c.csr_rp[i] = off; off += v; }
  if (CHD_TID == 0) { c.csr_rp[N] = total; c.csr_nnz = total; c.pm_dirty = 0; }
  CHD_SYNC();
  PAR_FOR(i, N) {
    int e = c.csr_rp[i];
    if (i < Nb) {
      for (int k = 0; k < MW; ++k) {
        unsigned long long m = c.pmb[(long long)i * MW + k];
        while (m) { c.csr_col[e] = i - w + k * 64 + __builtin_ctzll(m); c.csr_row[e] = i; ++e; m &= m - 1; }
      }
      for (int k = 0; k < CW; ++k) {
        unsigned long long m = c.pmt[(long long)i * CW + k];
        while (m) { c.csr_col[e] = Nb + k * 64 + __builtin_ctzll(m); c.csr_row[e] = i; ++e; m &= m - 1; }
      }
    } else {
      for (int k = 0; k < LW; ++k) {
        unsigned long long m = c.pmx[(long long)(i - Nb) * LW + k];
        while (m) { c.csr_col[e] = k * 64 + __builtin_ctzll(m); c.csr_row[e] = i; ++e; m &= m - 1; }
      }
    }
  }
  CHD_SYNC();
}
// start of a stage: the border blocks are reused with a new layout, and their structurally-zero left parts are
// neither cleared nor copied again afterwards
CHD_DEV void kreset(LCtx& c) {
  PAR_FOR(i, 2 * c.N) c.env[i] = c.q->ci[c.S->o_env + i];
  PAR_FOR(i, c.Nb) c.rcnt[i] = c.q->ci[c.S->o_rcnt + i];
  const long long nx_ = (long long)c.bc * c.LD;
  for (long long i = CHD_TID; i < nx_; i += CHD_NT) { c.K0x[i] = 0.0; c.Kfx[i] = 0.0; }
  const long long nf_ = (long long)c.Nb * (c.w + 1);
  for (long long i = CHD_TID; i < nf_; i += CHD_NT) c.Kfb[i] = 0.0;       // the copy into the factor only covers each row's envelope
  const long long n0_ = (long long)c.Nb * c.W2;
  for (long long i = CHD_TID; i < n0_; i += CHD_NT) c.K0b[i] = 0.0;       // kzero only touches marked entries
  PAR_FOR(i, c.Nb * c.MW) c.pmb[i] = 0ull;
  PAR_FOR(i, c.bc * c.LW) c.pmx[i] = 0ull;
  PAR_FOR(i, c.Nb * c.CW) c.pmt[i] = 0ull;
  PAR_FOR(i, c.N + 1) c.csr_rp[i] = 0;
  if (CHD_TID == 0) { c.csr_nnz = 0; c.pm_dirty = 0; }
  CHD_SYNC();
}


// sum_k a[k] * b[k] over k = k0, k0 + st, ... < kend with up to 16 loads in flight per lane (the operands live in HBM /
// L2 and these loops are latency bound), four accumulators
template <class AP, class BP>
CHD_DEV double dot_strided(AP a, BP b, int k, const int kend, const int st) {
  double s0 = 0, s1 = 0, s2 = 0, s3 = 0;
  for (; k + 15 * st < kend; k += 16 * st) {
    double v[16];
#pragma unroll
    for (int q = 0; q < 16; ++q) v[q] = a[k + q * st];
#pragma unroll
    for (int q = 0; q < 16; q += 4) {
      s0 += v[q] * b[k + q * st]; s1 += v[q + 1] * b[k + (q + 1) * st];
      s2 += v[q + 2] * b[k + (q + 2) * st]; s3 += v[q + 3] * b[k + (q + 3) * st];
    }
  }
  for (; k + 3 * st < kend; k += 4 * st) {
    s0 += a[k] * b[k]; s1 += a[k + st] * b[k + st]; s2 += a[k + 2 * st] * b[k + 2 * st]; s3 += a[k + 3 * st] * b[k + 3 * st];
  }
  for (; k < kend; k += st) s0 += a[k] * b[k];
  return (s0 + s1) + (s2 + s3);
}
// sum_r col[r * ld] * b[r] for r = r0 .. rend-1 (a column of a row-major matrix), same load depth
template <class AP, class BP>
CHD_DEV double dot_column(AP col, const long long ld, BP b, int r, const int rend) {
  double s0 = 0, s1 = 0, s2 = 0, s3 = 0;
  for (; r + 15 < rend; r += 16) {
    double v[16];
#pragma unroll
    for (int q = 0; q < 16; ++q) v[q] = col[(r + q) * ld];
#pragma unroll
    for (int q = 0; q < 16; q += 4) { s0 += v[q] * b[r + q]; s1 += v[q + 1] * b[r + q + 1]; s2 += v[q + 2] * b[r + q + 2]; s3 += v[q + 3] * b[r + q + 3]; }
  }
  for (; r + 3 < rend; r += 4) {
    s0 += col[r * ld] * b[r]; s1 += col[(r + 1) * ld] * b[r + 1]; s2 += col[(r + 2) * ld] * b[r + 2]; s3 += col[(r + 3) * ld] * b[r + 3];
  }
  for (; r < rend; ++r) s0 += col[r * ld] * b[r];
  return (s0 + s1) + (s2 + s3);
}
// sum over the list entries e0, e0 + st, ... < e1 of row `row`: K0(row, col_e) x[col_e], eight value loads in flight per lane
template <class XP>
CHD_DEV double list_dot(const LCtx& c, const int row, XP x, int e, const int e1, const int st) {
  double s0 = 0, s1 = 0;
  for (; e < e1; e += 8 * st) {
    int col[8]; double v[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) col[q] = e + q * st < e1 ? c.csr_col[e + q * st] : -1;
#pragma unroll
    for (int q = 0; q < 8; ++q) v[q] = col[q] >= 0 ? *k0_at(c, row, col[q]) : 0.0;
#pragma unroll
    for (int q = 0; q < 8; q += 2) { if (col[q] >= 0) s0 += v[q] * x[col[q]]; if (col[q + 1] >= 0) s1 += v[q + 1] * x[col[q + 1]]; }
  }
  return s0 + s1;
}
// y = K0 x (+ diag .* x), over the marked entries only
// `only` (optional): rows whose entry is <= 0 are skipped (their y is left untouched)
template <class XP>
CHD_DEV void kmatvec_impl(LCtx& c, XP x, GD* y, const GD* diag, const GI* only) {
  const int Nb = c.Nb, bc = c.bc;
  PAR_FOR(i, Nb) {             // a lane per band row (~12 band entries + the border rows coupled to the column)
    if (only && only[i] <= 0) continue;
    y[i] = list_dot(c, i, x, c.csr_rp[i], c.csr_rp[i + 1], 1) + (diag ? diag[i] * x[i] : 0.0);
  }
  GROUP_FOR(r, bc) {           // a lane group per border row (tens to thousands of entries)
    if (only && only[Nb + r] <= 0) continue;
    const double acc = group_sum(list_dot(c, Nb + r, x, c.csr_rp[Nb + r] + lane_, c.csr_rp[Nb + r + 1], CHD_GL));
    if (lane_ == 0) y[Nb + r] = acc + (diag ? diag[Nb + r] * x[Nb + r] : 0.0);
  }
  CHD_SYNC();
}
CHD_NOINLINE CHD_DEV void kmatvec(LCtx& c, const GD* x, GD* y, const GD* diag, const GI* only) {
  TIC();
  if (c.N <= c.lds_cap - LDS_RED) {          // x is read 2 w + 1 times: keep it in LDS
    LdsD* xs = c.lds + LDS_RED;
    PAR_FOR(i, c.N) xs[i] = x[i];
    CHD_SYNC();
    kmatvec_impl(c, (const LdsD*)xs, y, diag, only);
  } else kmatvec_impl(c, x, y, diag, only);
  TOC(c, 4);
}

// ---- factorisation: K0 + diag -> L D L^T in Kf (no pivoting; expected pivot sign from `sign`) ----
CHD_DEV double pivot_fix(LCtx& c, double d, int sg) {
  if (!(d * sg > 1e-14)) { d = sg * 1e-10; if (CHD_TID == 0) c.n_bad_pivots++; }
  return d;
}
// the factorisation just made is unusable: a pivot did not have the expected sign
CHD_DEV bool factor_failed(LCtx& c) { return block_sum(c, CHD_TID == 0 ? (double)c.n_bad_pivots : 0.0) > 0.0; }

// wave-0-only sections: one wavefront works through a short dependent chain in LDS while the
// other waves wait at the next workgroup barrier
#ifdef CHD_HOST_EMU
#define CHD_WAVE0 true
#define CHD_WLANE 0
#define CHD_WSTEP 1
#define CHD_WSYNC() ((void)0)
#else
#define CHD_WAVE0 (threadIdx.x < 64)
#define CHD_WLANE ((int)threadIdx.x)
#define CHD_WSTEP 64
#define CHD_WSYNC() do { __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup"); __builtin_amdgcn_wave_barrier(); } while (0)
#endif


// ---- diagonal block of a panel: the NB x NB block of Kf at (c0, c0) -> unit-lower L (dense LDS copy DL[k * NB + j]
//      = L(j, k)), pivots dv[0..NB) and their reciprocals dv[32..).  Rows >= jb of a short last block are identity.
//      The block is read straight from HBM by one wavefront (lane a = row a), so it can be factored while the other
//      wavefronts are still busy with the previous panel's trailing update (look-ahead).
#ifdef CHD_HOST_EMU
template <int NB>
CHD_DEV void diag_block_g(LCtx& c, const GI* sign, LdsD* dv, LdsD* DL, const int c0, const int jb) {
  const int W1 = c.w + 1, w = c.w;
  double A[NB][NB];
  for (int a = 0; a < NB; ++a)
    for (int j = 0; j < NB; ++j) A[a][j] = (a < jb && j <= a) ? c.Kfb[(long long)(c0 + a) * W1 + (j - a + w)] : (a == j ? 1.0 : 0.0);
  for (int j = 0; j < NB; ++j) {
    double d = A[j][j];
    if (j < jb) d = pivot_fix(c, d, sign[c0 + j]);
    const double inv = 1.0 / d;
    for (int a = j + 1; a < NB; ++a) { A[a][j] *= inv; DL[j * NB + a] = A[a][j]; }
    dv[j] = d; dv[32 + j] = inv;
    for (int jj = j + 1; jj < NB; ++jj)
      for (int a = jj; a < NB; ++a) A[a][jj] -= A[a][j] * d * A[jj][j];
  }
}
#else
CHD_DEV double readlane_f64(double v, int l) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_readlane(lo, l); hi = __builtin_amdgcn_readlane(hi, l);
  return __hiloint2double(hi, lo);
}
CHD_DEV double rcp_f64(double d) {
  double x = __builtin_amdgcn_rcp(d);
  x = __builtin_fma(__builtin_fma(-d, x, 1.0), x, x);
  x = __builtin_fma(__builtin_fma(-d, x, 1.0), x, x);
  return x;
}
// Left-looking: at column j lane a forms A(a,j) - sum_{k<j} [L(a,k) d_k] L(j,k).  Row j of L was written to the dense
// LDS copy column by column at the earlier steps, so it arrives as pipelined broadcast reads issued one step ahead;
// only L(j, j-1) and the pivot travel by v_readlane (whose result takes tens of cycles to reach the VALU -- the
// right-looking form needed 31 - j of them per column and took ~11 us per block).
// (first wavefront only)  ar[j] = row a = lane of the block, as the factor storage holds it
template <int NB>
__device__ __forceinline__ void diag_load(LCtx& c, double (&ar)[NB], const int c0, const int jb) {
  const int W1 = c.w + 1, w = c.w;
  const int a = threadIdx.x;
  const GD* src = c.Kfb + (long long)(c0 + (a < jb ? a : 0)) * W1 + (w - (a < jb ? a : 0));      // (left of a row's envelope the factor storage is zero)
#pragma unroll
  for (int j = 0; j < NB; ++j) { const bool in = a < jb && j <= a; const double t = *(in ? src + j : c.Kfb + w); ar[j] = in ? t : (a == j ? 1.0 : 0.0); }
}
template <int NB>
__device__ __forceinline__ void diag_chain(LCtx& c, const GI* sign, LdsD* dv, LdsD* DL, const double (&ar)[NB], const int c0, const int jb) {
  const int a = threadIdx.x;
  const bool act = a < NB;
  const int sg_a = a < jb ? sign[c0 + a] : 1;          // expected pivot signs, fetched once
  const unsigned long long sg_pos = __ballot(sg_a > 0);
  double u[NB], row[NB];           // u[k] = L(a,k) d_k; row[k] = L(j,k) of the column being formed
#pragma unroll
  for (int k = 0; k < NB; ++k) { u[k] = 0.0; row[k] = 0.0; }
  double lprev = 0.0;
  int bad = 0;
#pragma unroll
  for (int j = 0; j < NB; ++j) {
    if (j > 0) row[j - 1] = readlane_f64(lprev, j);          // L(j, j-1): produced by the previous column
    double s0 = ar[j], s1 = 0.0, s2 = 0.0, s3 = 0.0;
#pragma unroll
    for (int k = 0; k + 3 < j; k += 4) { s0 -= u[k] * row[k]; s1 -= u[k + 1] * row[k + 1]; s2 -= u[k + 2] * row[k + 2]; s3 -= u[k + 3] * row[k + 3]; }
#pragma unroll
    for (int k = j & ~3; k < j; ++k) s0 -= u[k] * row[k];
    const double v = (s0 + s1) + (s2 + s3);
    // row j + 1 of L up to column j - 1 (written at earlier columns): requested now, arrives while the pivot chain runs
    if (j + 1 < NB) {
#pragma unroll
      for (int k = 0; k < j; ++k) row[k] = DL[k * NB + (j + 1)];
    }
    double d = readlane_f64(v, j);
    if (j < jb) { const double sg = ((sg_pos >> j) & 1ull) ? 1.0 : -1.0; if (!(d * sg > 1e-14)) { d = sg * 1e-10; ++bad; } }
    const double inv = rcp_f64(d);
    const double lj = v * inv;            // L(a, j) for a > j
    u[j] = v; lprev = lj;
    if (act && a > j) DL[j * NB + a] = lj;
    if (a == j) { dv[j] = d; dv[32 + j] = inv; }
  }
  if (threadIdx.x == 0) c.n_bad_pivots += bad;
}
template <int NB>
CHD_DEV void diag_block_g(LCtx& c, const GI* sign, LdsD* dv, LdsD* DL, const int c0, const int jb) {
  if (threadIdx.x < 64) {
    double ar[NB];
    diag_load<NB>(c, ar, c0, jb);
    diag_chain<NB>(c, sign, dv, DL, ar, c0, jb);
  }
}
#endif

// ---- active rows of a panel's window (sorted): u in [0, wr) whose first coupled column (efirst of a band row i0 + u,
//      bfirst of a border row u - nbelow) is <= last_col
#ifdef CHD_HOST_EMU
CHD_DEV void build_active_rows(LCtx& c, int* act, int* nact, int wr, int nbelow, int i0, int last_col) {
  int n = 0;
  for (int u = 0; u < wr; ++u) if (c.env[2 * (u >= nbelow ? c.Nb + u - nbelow : i0 + u)] <= last_col) act[n++] = u;
  *nact = n;
}
#else
CHD_DEV void build_active_rows(LCtx& c, int* act_, int* nact_, int wr, int nbelow, int i0, int last_col) {
  if (threadIdx.x >= 64 && threadIdx.x < 128) {       // one wavefront (the second): ballot + prefix count keeps the list sorted
    LdsI* act = (LdsI*)act_; LdsI* nact = (LdsI*)nact_;
    const int ln = threadIdx.x - 64;
    constexpr int MR = 10;                       // 64 MR >= w + b: the envelope starts are fetched together
    int ef[MR];
#pragma unroll
    for (int r = 0; r < MR; ++r) { const int u = 64 * r + ln; ef[r] = c.env[u < wr ? 2 * (u >= nbelow ? c.Nb + u - nbelow : i0 + u) : 0]; }
    int base = 0;
#pragma unroll
    for (int r = 0; r < MR; ++r) {
      const int u = 64 * r + ln;
      const bool on = u < wr && ef[r] <= last_col;
      const unsigned long long m = __ballot(on);
      if (on) act[base + __popcll(m & ((1ull << ln) - 1ull))] = u;
      base += __popcll(m);
    }
    for (int u0 = 64 * MR; u0 < wr; u0 += 64) {
      const int u = u0 + ln;
      const bool on = u < wr && c.env[2 * (u >= nbelow ? c.Nb + u - nbelow : i0 + u)] <= last_col;
      const unsigned long long m = __ballot(on);
      if (on) act[base + __popcll(m & ((1ull << ln) - 1ull))] = u;
      base += __popcll(m);
    }
    if (ln == 0) *nact = base;
  }
}
#endif

// ---- trailing update: window -= L_below D L_below^T (lower triangle), restricted to the active rows ----------
// window rows/cols u = 0..wr-1 live at PT rows NB+u; u < nbelow are band rows i0+u, the rest border rows
#ifdef CHD_HOST_EMU
template <int NB>
CHD_DEV void trailing_update(LCtx& c, const LdsD* dv, const LdsD* PT, const int ldp, const int wr, const int nbelow, const int i0, const int* act, const int nact, const bool) {
  const int W1 = c.w + 1, w = c.w, LD = c.LD, Nb = c.Nb;
  for (int tr = 0; tr < nact; ++tr)
    for (int tc = 0; tc <= tr; ++tc) {
      const int ur = act[tr], uc = act[tc];
      double v = 0;
      for (int j = 0; j < NB; ++j) v += PT[j * ldp + NB + ur] * (PT[j * ldp + NB + uc] * dv[j]);
      if (ur < nbelow) { const int i = i0 + ur, k = i0 + uc; c.Kfb[(long long)i * W1 + (k - i + w)] -= v; }
      else if (uc < nbelow) c.Kfx[(long long)(ur - nbelow) * LD + i0 + uc] -= v;
      else c.Kfx[(long long)(ur - nbelow) * LD + Nb + (uc - nbelow)] -= v;
    }
}
#else
typedef double chd_f64x4 __attribute__((ext_vector_type(4)));
// one 16x16 tile of the COMPACTED window per wavefront pass, K = NB in steps of 4 on the fp64 matrix core
// (v_mfma_f64_16x16x4_f64: A[row = lane & 15][k = lane >> 4], B[k = lane >> 4][col = lane & 15],
//  D[row = (lane >> 4) + 4 * reg][col = lane & 15])
// `split`: the first wavefront only takes the three tiles that cover the next panel's diagonal block (it goes on to
// factor that block), the other wavefronts share the rest; otherwise all wavefronts share all tiles
template <int NB>
CHD_DEV void trailing_update(LCtx& c, const LdsD* dv, const LdsD* PT, const int ldp, const int wr, const int nbelow, const int i0, const int* act_, const int nact, const bool split) {
  const int W1 = c.w + 1, w = c.w, LD = c.LD, Nb = c.Nb;
  const LdsI* act = (const LdsI*)act_;
  const int wave = threadIdx.x >> 6, nwv = blockDim.x >> 6, lane = threadIdx.x & 63;
  const int lr = lane & 15, lk = lane >> 4;
  const int nt = (nact + 15) >> 4;
  const int ntri = nt * (nt + 1) / 2;
  const int zrow = wr + 8;          // a zero (padding) row of the panel
  auto tile_of = [](int t, int& tr, int& tc) {
    tr = (int)((__fsqrt_rn(8.0f * (float)t + 1.0f) - 1.0f) * 0.5f);      // (single precision + the two corrections below: t < 10^4)
    while ((tr + 1) * (tr + 2) / 2 <= t) ++tr;
    while (tr * (tr + 1) / 2 > t) --tr;
    tc = t - tr * (tr + 1) / 2;
  };
  auto dest = [&](int ur, int uc) -> GD* {
    if (ur < nbelow) { const int i = i0 + ur, k = i0 + uc; return c.Kfb + (long long)i * W1 + (k - i + w); }
    if (uc < nbelow) return c.Kfx + (long long)(ur - nbelow) * LD + i0 + uc;
    return c.Kfx + (long long)(ur - nbelow) * LD + Nb + (uc - nbelow);
  };
  // TP independent tiles per pass: the old window values of all of them are requested before the MFMA
  // chains start (memory-level parallelism), and the chains interleave on the matrix pipe
  constexpr int TP = CHD_TRAIL_TP;
  const int t_first = !split ? wave : wave == 0 ? 0 : 3 + (wave - 1);
  const int t_stride = !split ? nwv : wave == 0 ? 1 : nwv - 1;                 // between the TP tiles of one pass
  const int t_limit = (split && wave == 0) ? (ntri < 3 ? ntri : 3) : ntri;
  // two passes in flight: the old window values of pass p + 1 are requested before the matrix-core chains of pass p start, so a
  // wavefront's passes cost one round trip to L2 / HBM together instead of one each (ping-pong between two register sets)
  struct TileSet { GD* pd[TP][4]; double old_[TP][4]; bool ok[TP][4]; const LdsD* pa[TP]; const LdsD* pb[TP]; };
  auto fetch = [&](TileSet& S, const int t0) {
#pragma unroll
    for (int u = 0; u < TP; ++u) {
      const int t = t0 + u * t_stride;
      const bool live = t < t_limit;
      int tr, tc;
      tile_of(live ? t : t0, tr, tc);
      const int ira = 16 * tr + lr, icb = 16 * tc + lr;           // compact indices of this lane's A row / B column
      const int ua = ira < nact ? act[ira] : zrow, ub = icb < nact ? act[icb] : zrow;
      S.pa[u] = PT + NB + ua; S.pb[u] = PT + NB + ub;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int irr = 16 * tr + lk + 4 * r;                     // compact row index of D register r
        S.ok[u][r] = live && irr < nact && icb < nact && icb <= irr;
        const int ur = S.ok[u][r] ? act[irr] : 0;
        S.pd[u][r] = dest(S.ok[u][r] ? ur : 0, S.ok[u][r] ? ub : 0);
        S.old_[u][r] = S.ok[u][r] ? *S.pd[u][r] : 0.0;
      }
    }
  };
  auto finish = [&](const TileSet& S) {
    chd_f64x4 acc[TP];
#pragma unroll
    for (int u = 0; u < TP; ++u) acc[u] = chd_f64x4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int kk = 0; kk < NB / 4; ++kk) {
      const int j = kk * 4 + lk;
      const double dj = dv[j];
#pragma unroll
      for (int u = 0; u < TP; ++u) acc[u] = __builtin_amdgcn_mfma_f64_16x16x4f64(S.pa[u][j * ldp], S.pb[u][j * ldp] * dj, acc[u], 0, 0, 0);
    }
#pragma unroll
    for (int u = 0; u < TP; ++u)
#pragma unroll
      for (int r = 0; r < 4; ++r)
        if (S.ok[u][r]) *S.pd[u][r] = S.old_[u][r] - acc[u][r];
  };
  const int t_step = TP * t_stride;
  TileSet A, B;
  int t0 = t_first;
  if (t0 < t_limit) fetch(A, t0);
  while (t0 < t_limit) {
    if (t0 + t_step < t_limit) fetch(B, t0 + t_step);
    finish(A);
    if (!(t0 + t_step < t_limit)) break;
    if (t0 + 2 * t_step < t_limit) fetch(A, t0 + 2 * t_step);
    finish(B);
    t0 += 2 * t_step;
  }
}
#endif

#ifndef CHD_HOST_EMU
// The first wavefront's share of a panel's trailing update when another panel follows (look-ahead): the three tiles of the compacted
// window that cover the next panel's diagonal block, then that block's L D L^T.  The block never goes through memory on the way:
// its old values are requested row-per-lane (the layout of the column chain) TOGETHER with the tiles' old values, the three
// accumulator tiles are handed from the matrix-core layout to row-per-lane through the LDS buffer that is about to receive the
// block's L (DL_n), and the chain starts from old - update in registers.  (Before: tiles, read-modify-write, fence, reload of the
// block, chain -- two dependent round trips to L2 and a store drain on the critical path of the whole workgroup, 5.9 + 5.5 us of the
// 13.8 us the phase took per panel.)
template <int NB>
CHD_NOINLINE CHD_DEV void lookahead_wave(LCtx& c, const GI* sign, const LdsD* dv, const LdsD* PT, const int ldp, const int wr, const int nbelow, const int i0,
                            const int* act_, const int nact, LdsD* dv_n, LdsD* DL_n, const int c0n, const int jbn) {
  const int W1 = c.w + 1, w = c.w, LD = c.LD, Nb = c.Nb;
  const LdsI* act = (const LdsI*)act_;
  const int lane = threadIdx.x & 63, lr = lane & 15, lk = lane >> 4;
  const int zrow = wr + 8;          // a zero (padding) row of the panel
#ifdef CHD_DIAG_TIMING
  const long long tl0_ = CHD_CLOCK();
#endif
  double ar[NB];
  diag_load<NB>(c, ar, c0n, jbn);                    // (requests only: nothing waits on them before the tiles' loads are out as well)
  auto dest = [&](int ur, int uc) -> GD* {
    if (ur < nbelow) { const int i = i0 + ur, k = i0 + uc; return c.Kfb + (long long)i * W1 + (k - i + w); }
    if (uc < nbelow) return c.Kfx + (long long)(ur - nbelow) * LD + i0 + uc;
    return c.Kfx + (long long)(ur - nbelow) * LD + Nb + (uc - nbelow);
  };
  constexpr int TT = 3;                              // tiles (0,0), (1,0), (1,1) of the compacted window
  const int ntl = nact > 16 ? 3 : nact > 0 ? 1 : 0;
  GD* pd[TT][4]; double old_[TT][4]; bool ok[TT][4]; int urr[TT][4]; int ubb[TT];
  const LdsD* pa[TT]; const LdsD* pb[TT];
#pragma unroll
  for (int u = 0; u < TT; ++u) {
    const int tr = u == 0 ? 0 : 1, tc = u == 2 ? 1 : 0;
    const bool live = u < ntl;
    const int ira = 16 * tr + lr, icb = 16 * tc + lr;
    const int ua = ira < nact ? act[ira] : zrow, ub = icb < nact ? act[icb] : zrow;
    pa[u] = PT + NB + ua; pb[u] = PT + NB + ub; ubb[u] = ub;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int irr = 16 * tr + lk + 4 * r;
      ok[u][r] = live && irr < nact && icb < nact && icb <= irr;
      const int ur = ok[u][r] ? act[irr] : 0;
      urr[u][r] = ur;
      pd[u][r] = dest(ok[u][r] ? ur : 0, ok[u][r] ? ub : 0);
      old_[u][r] = ok[u][r] ? *pd[u][r] : 0.0;
    }
  }
  // the hand-over buffer S[row * NB + col] (= DL_n, not yet in use): zero where no tile entry lands
#pragma unroll
  for (int q = 0; q < NB * NB / 64; ++q) DL_n[lane + 64 * q] = 0.0;
  chd_f64x4 acc[TT];
#pragma unroll
  for (int u = 0; u < TT; ++u) acc[u] = chd_f64x4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
  for (int kk = 0; kk < NB / 4; ++kk) {
    const int j = kk * 4 + lk;
    const double dj = dv[j];
#pragma unroll
    for (int u = 0; u < TT; ++u) acc[u] = __builtin_amdgcn_mfma_f64_16x16x4f64(pa[u][j * ldp], pb[u][j * ldp] * dj, acc[u], 0, 0, 0);
  }
#ifdef CHD_DIAG_TIMING
  asm volatile("s_nop 0" : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]));
  const long long tl1_ = CHD_CLOCK();
#endif
  // (one wavefront, and LDS operations complete in program order: zeros, entries and the reads below need no wait between them, only
  //  the compiler kept from reordering; columns rotated by the row index -- S[row][(col + row) mod NB] -- so that the 32 lanes of a
  //  column read hit 32 different banks)
  asm volatile("" ::: "memory");
#pragma unroll
  for (int u = 0; u < TT; ++u)
#pragma unroll
    for (int r = 0; r < 4; ++r)
      if (ok[u][r] && urr[u][r] < jbn) DL_n[urr[u][r] * NB + ((ubb[u] + urr[u][r]) & (NB - 1))] = acc[u][r];      // a pivot row of the next panel (then so is the column: ub <= ur)
  asm volatile("" ::: "memory");
  {
    const int a = lane < NB ? lane : 0;
#pragma unroll
    for (int j = 0; j < NB; ++j) ar[j] -= DL_n[a * NB + ((j + a) & (NB - 1))];      // (lanes >= NB: identity padding rows; they stay out of the block)
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // all of S is in registers before the chain writes L into the same buffer
#ifdef CHD_DIAG_TIMING
  const long long tl2_ = CHD_CLOCK();
#endif
  // the rest of the three tiles: window rows below the block that sit among the first 32 active ones (when some of the next pivot rows
  // are not active yet).  The block's own updated values are NOT written back: nothing reads them again -- the next panel stores L there.
#pragma unroll
  for (int u = 0; u < TT; ++u)
#pragma unroll
    for (int r = 0; r < 4; ++r)
      if (ok[u][r] && urr[u][r] >= jbn) *pd[u][r] = old_[u][r] - acc[u][r];
#ifdef CHD_DIAG_TIMING
  const long long tl3_ = CHD_CLOCK();
#endif
  diag_chain<NB>(c, sign, dv_n, DL_n, ar, c0n, jbn);
#ifdef CHD_DIAG_TIMING
  if (threadIdx.x == 0) { c.tacc[7] += tl1_ - tl0_; c.tacc[13] += tl2_ - tl1_; c.tacc[14] += tl3_ - tl2_; c.tacc[15] += CHD_CLOCK() - tl3_; }
#endif
}
#endif

template <int NB>
CHD_DEV void trailing_phase(LCtx& c, const GI* sign, const LdsD* dv, const LdsD* PT, const int ldp, const int wr, const int nbelow, const int i0,
                                         const int* act, const int nact, const bool more, LdsD* dv_n, LdsD* DL_n, const int c0n, const int jbn) {
#ifdef CHD_HOST_EMU
  trailing_update<NB>(c, dv, PT, ldp, wr, nbelow, i0, act, nact, more);
  if (more) diag_block_g<NB>(c, sign, dv_n, DL_n, c0n, jbn);
#else
  (void)sign; (void)dv_n; (void)DL_n; (void)c0n; (void)jbn;      // (the look-ahead wavefront is dispatched by the caller: lookahead_wave)
  trailing_update<NB>(c, dv, PT, ldp, wr, nbelow, i0, act, nact, more);
#endif
}

// Banded part of the factorisation, NB columns per panel.  Panel in LDS, column-major PT[j * ldp + a];
// local rows: [0, NB) diagonal block (rows >= jb of a short last block are identity padding),
// [NB, NB + nbelow) band rows below it, then the bc border rows.
// Every phase is its own function: each gets its own register allocation (the row solve and the MFMA update
// want ~200 VGPRs each, and inlined together they spill).
struct Panel {
  int c0, jb, nbelow, pr, ldp;
  LdsD* dv; LdsD* DL; LdsD* PT;
  int* act; int* nact_p;
};

// ---- load (zero padded; identity in the padding columns); one task = 8 consecutive columns of one row
// part 0: the sorted list of active window rows (second wavefront) and this panel's zero padding row;
// part 1: the active rows below the diagonal block (which was factored straight from HBM during the previous panel's
// trailing update).  Rows that are not active keep whatever an earlier panel left in the LDS panel: nothing reads them.
template <int NB>
CHD_NOINLINE CHD_DEV void panel_load(LCtx& c, const Panel P, const int part) {
  const int W1 = c.w + 1, w = c.w, LD = c.LD;
  const int c0 = P.c0, jb = P.jb, nbelow = P.nbelow, pr = P.pr, ldp = P.ldp;
  LdsD* PT = P.PT;
  if (part == 0) {
    // window rows that this panel can touch: rows whose envelope reaches the panel (second wavefront)
    build_active_rows(c, P.act, P.nact_p, pr - NB, nbelow, c0 + jb, c0 + jb - 1);
    return;
  }
  PAR_FOR(j, NB) PT[j * ldp + pr + 8] = 0.0;      // the zero padding row of this panel (an earlier, longer panel may have used it)
  const LdsI* act = (const LdsI*)P.act;
  const int nact = *(const LdsI*)P.nact_p;
  PAR_FOR(idx, nact * (NB / 8)) {
    const int u = act[idx / (NB / 8)], a = NB + u, j0 = (idx % (NB / 8)) * 8;
    const bool band = u < nbelow;
    const int i = c0 + jb + u;
    // left of a row's envelope the factor storage is zero (kreset + envelope-limited copy): only the band limit is checked
    const GD* src = band ? c.Kfb + (long long)i * W1 + (c0 + j0 - i + w) : c.Kfx + (long long)(u - nbelow) * LD + c0 + j0;
    double v[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const int j = j0 + q, k = c0 + j;
      const bool ok = j < jb && (!band || i - k <= w);
      v[q] = ok ? src[q] : 0.0;
    }
#pragma unroll
    for (int q = 0; q < 8; ++q) PT[(j0 + q) * ldp + a] = v[q];
  }
}

// ---- (B) rows below: y_j = A(a,j) - sum_{k<j} y_k L(j,k);  L(a,j) = y_j / d_j
//      One row per thread; column k + 1 of L (LDS broadcast reads from the dense copy of the diagonal block) is requested
//      before the multiply-adds of column k, so the LDS latency is paid once, not once per column (the first version
//      waited ~100 cycles per read pair: 9 us per panel for 2 us of arithmetic).
template <int NB>
CHD_DEV void panel_rows(LCtx& c, const Panel P, const int nact) {
  const int ldp = P.ldp;
  LdsD* PT = P.PT; const LdsD* DL = P.DL; const LdsD* dv = P.dv; const LdsI* act = (const LdsI*)P.act;
  PAR_FOR(t2, nact) {           // compacted: inactive rows are never read
    const int a0 = NB + act[t2];
    double y0[NB], cur[NB], nx[NB];
#pragma unroll
    for (int j = 0; j < NB; ++j) { y0[j] = PT[j * ldp + a0]; cur[j] = j > 0 ? DL[j] : 0.0; nx[j] = 0.0; }
#pragma unroll
    for (int k = 0; k < NB - 1; ++k) {
#pragma unroll
      for (int j = k + 2; j < NB; ++j) nx[j] = DL[(k + 1) * NB + j];          // next column: in flight during this one's FMAs
      CHD_SCHED_FENCE();
#pragma unroll
      for (int j = k + 1; j < NB; ++j) y0[j] -= y0[k] * cur[j];               // independent FMAs
      CHD_SCHED_FENCE();
#pragma unroll
      for (int j = k + 2; j < NB; ++j) cur[j] = nx[j];
    }
#pragma unroll
    for (int j = 0; j < NB; ++j) PT[j * ldp + a0] = y0[j] * dv[32 + j];
  }
}

// ---- write the panel back (8 consecutive columns of one row per task): the diagonal block rows and the active rows.
//      Entries left of a row's envelope are exact zeros (as is what the storage holds there): only the band limit is checked.
//      (`t0`: first participating thread -- the wavefront that factors the next diagonal block stays out of it)
template <int NB>
CHD_NOINLINE CHD_DEV void panel_store(LCtx& c, const Panel P, const int nact, const int t0) {
  const int W1 = c.w + 1, w = c.w, LD = c.LD;
  const int c0 = P.c0, jb = P.jb, nbelow = P.nbelow, ldp = P.ldp;
  const LdsD* PT = P.PT; const LdsD* dv = P.dv; const LdsD* DL = P.DL;
  const LdsI* act = (const LdsI*)P.act;
  PAR_FOR_FROM(idx, (jb + nact) * (NB / 8), t0) {
    const int r = idx / (NB / 8), j0 = (idx % (NB / 8)) * 8;
    if (r < jb) {                 // a row of the diagonal block (it only lives in its dense copy)
      const int a = r, i = c0 + a;
      GD* dst = c.Kfb + (long long)i * W1 + (c0 + j0 - i + w);
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const int j = j0 + q;
        if (j < a) dst[q] = DL[j * NB + a];
        else if (j == a) dst[q] = dv[j];
      }
    } else {
      const int u = act[r - jb], a = NB + u;
      const bool band = u < nbelow;
      const int i = c0 + jb + u;
      GD* dst = band ? c.Kfb + (long long)i * W1 + (c0 + j0 - i + w) : c.Kfx + (long long)(u - nbelow) * LD + c0 + j0;
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const int j = j0 + q, k = c0 + j;
        if (j < jb && (!band || i - k <= w)) dst[q] = PT[j * ldp + a];
      }
    }
  }
}

template <int NB>
CHD_DEV void panel_geometry(LCtx& c, Panel& P, const int c0, const int ldp) {
  const int Nb = c.Nb, w = c.w, bc = c.bc;
  P.c0 = c0; P.jb = Nb - c0 < NB ? Nb - c0 : NB;
  const int nbr = (Nb - c0 < P.jb + w) ? Nb - c0 : P.jb + w;     // band rows touched by this panel
  P.nbelow = nbr - P.jb;
  P.pr = NB + P.nbelow + bc; P.ldp = ldp;
}
template <int NB>
CHD_DEV void kfactor_band(LCtx& c, const GI* sign, LdsD* dv, LdsD* DL, LdsD* dv2, LdsD* DL2, LdsD* PT, const int ldp) {
  const int Nb = c.Nb, w = c.w, bc = c.bc;
  // look-ahead: the diagonal block of panel J + 1 is factored by the first wavefront during panel J's trailing update
  // (lookahead_wave: the three tiles of that update which touch the block stay in registers / LDS on the way), into the other
  // (dv, DL) pair; the second wavefront builds panel J + 1's list of active rows at the start of the same phase (second list
  // buffer); the panel's own columns are stored after the update (by the other seven wavefronts)
  const int lsz = w + bc + 64 + 2;                           // ints per list (+ its count)
  int* actA = (int*)(PT + (long long)ldp * NB); int* actB = actA + lsz;
  {
    Panel P0; panel_geometry<NB>(c, P0, 0, ldp);
    P0.dv = dv; P0.DL = DL; P0.PT = PT; P0.act = actA; P0.nact_p = actA + lsz - 2;
    panel_load<NB>(c, P0, 0);
  }
  diag_block_g<NB>(c, sign, dv, DL, 0, Nb < NB ? Nb : NB);
  CHD_SYNC();
  for (int c0 = 0; c0 < Nb; c0 += NB) {
#if CHD_INERTIA_RETRY && CHD_ABORT_BAD_FACTOR
    // A diagonal block met a pivot of unexpected sign: the factorisation is going to be discarded (inertia retry), so it stops here -- before the block's
    // columns (multipliers of order 1e10 after the pivot was replaced) are stored and can overflow into NaNs that would stay in the factor storage
    // outside the envelope.  (The count lives in the LDS context: uniform after the barrier.)
    if (c.n_bad_pivots > 0) return;
#endif
    Panel P; panel_geometry<NB>(c, P, c0, ldp);
    P.dv = dv; P.DL = DL; P.PT = PT;
    P.act = actA; P.nact_p = actA + lsz - 2;
    long long tp_ = CHD_CLOCK();
    const int nact = *(const LdsI*)P.nact_p;
    panel_load<NB>(c, P, 1);
    CHD_SYNC();
    TACC(c, 8, CHD_CLOCK() - tp_); tp_ = CHD_CLOCK();
    panel_rows<NB>(c, P, nact);
    CHD_SYNC();
    TACC(c, 9, CHD_CLOCK() - tp_); tp_ = CHD_CLOCK();
    // ---- trailing update of the window (+ the next diagonal block and the next list of active rows)
    const int c0n = c0 + NB;
    const bool more = c0n < Nb;
    if (more) {
      Panel Pn; panel_geometry<NB>(c, Pn, c0n, ldp);
      Pn.dv = dv2; Pn.DL = DL2; Pn.PT = PT; Pn.act = actB; Pn.nact_p = actB + lsz - 2;
#ifdef CHD_HOST_EMU
      panel_load<NB>(c, Pn, 0);
#else
      if (threadIdx.x >= 64 && threadIdx.x < 128) panel_load<NB>(c, Pn, 0);      // (the list is built by the second wavefront: the others skip the call)
#endif
    }
#ifndef CHD_HOST_EMU
#ifdef CHD_B_TIMING
    const long long tb0_ = CHD_CLOCK();
#endif
    if (more && threadIdx.x < 64) lookahead_wave<NB>(c, sign, dv, PT, ldp, P.pr - NB, P.nbelow, c0 + P.jb, P.act, nact, dv2, DL2, c0n, Nb - c0n < NB ? Nb - c0n : NB);
    else
#endif
    trailing_phase<NB>(c, sign, dv, PT, ldp, P.pr - NB, P.nbelow, c0 + P.jb, P.act, nact, more, dv2, DL2, c0n, Nb - c0n < NB ? Nb - c0n : NB);
    TACC(c, 11, CHD_CLOCK() - tp_); tp_ = CHD_CLOCK();
#if defined(CHD_B_TIMING) && !defined(CHD_HOST_EMU)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const long long tb1_ = CHD_CLOCK();
#endif
    // the panel's own columns go out last: nothing reads them before the barrier, and ahead of the update they would only stand
    // between its loads and the memory (the vector-memory counter retires in order)
#ifdef CHD_HOST_EMU
    panel_store<NB>(c, P, nact, 0);
#else
    panel_store<NB>(c, P, nact, more ? 64 : 0);
#endif
#if defined(CHD_B_TIMING) && !defined(CHD_HOST_EMU)
    const long long tb2_ = CHD_CLOCK();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const long long tb3_ = CHD_CLOCK();
    if (threadIdx.x == 64) { c.tacc[16] += tb1_ - tb0_; c.tacc[17] += tb2_ - tb1_; c.tacc[18] += tb3_ - tb2_; }       // second wavefront: list + tiles (drained), panel store issue, its drain
    if (threadIdx.x == 448) { c.tacc[19] += tb1_ - tb0_; c.tacc[20] += tb3_ - tb1_; }                                    // last wavefront: tiles (drained), store + drain
#endif
    CHD_SYNC();
    TACC(c, 10, CHD_CLOCK() - tp_);
    LdsD* t_ = dv; dv = dv2; dv2 = t_; t_ = DL; DL = DL2; DL2 = t_;
    int* ti_ = actA; actA = actB; actB = ti_;
  }
}

// in-place L D L^T of a dense symmetric n x n matrix (lower triangle, leading dimension ld).
// The columns stay unscaled (L D) while the elimination runs -- later columns never touch them -- and are divided
// by their pivots in one pass at the end: one workgroup barrier per column instead of three.
template <class P>
CHD_DEV void dense_ldlt(LCtx& c, P Sp, const int ld, const int n, const GI* sign) {
  int sg_next = n > 0 ? sign[0] : 1;
  for (int j = 0; j < n; ++j) {
    const int sg = sg_next;
    if (j + 1 < n) sg_next = sign[j + 1];            // (fetched a column ahead)
    const double d = pivot_fix(c, Sp[(long long)j * ld + j], sg);
    const double id = 1.0 / d;
    if (CHD_TID == 0) Sp[(long long)j * ld + j] = d;
    for (int r = j + 1 + CHD_TID / 8; r < n; r += CHD_NT / 8 > 0 ? CHD_NT / 8 : 1) {
      const double f = Sp[(long long)r * ld + j] * id;
      for (int k = j + 1 + CHD_TID % 8; k <= r; k += CHD_NT >= 8 ? 8 : 1) Sp[(long long)r * ld + k] -= f * Sp[(long long)k * ld + j];
    }
    CHD_SYNC();
  }
  PAR_FOR(idx, n * n) { const int r = idx / n, k = idx % n; if (k < r) Sp[(long long)r * ld + k] /= Sp[(long long)k * ld + k]; }
  CHD_SYNC();
}

// The same factorisation for a block that does not fit LDS (the 700-row border of a 600-frame sequence: 3.9 MB), blocked: PB columns at a time are
// factored in LDS (rows j0 .. n-1 of them: the only per-column barriers are LDS ones), written back, and the rest of the lower triangle is updated from
// the LDS copy, one thread per entry.  Same result format as dense_ldlt: D on the diagonal, unit-lower L below.  (The column-at-a-time version above
// spent 23 ms per factorisation in HBM round trips between 700 barriers: 29 % of a 600-frame solve.)
template <class SP>
CHD_DEV void dense_ldlt_blocked(LCtx& c, SP Sg, const int ld, const int n, const GI* sign, LdsD* P, const int PB) {
  const int LP = PB + 1;                           // padded leading dimension of the LDS panel (a lane per row walks a column: no bank conflicts)
  for (int j0 = 0; j0 < n; j0 += PB) {
    const int jb = n - j0 < PB ? n - j0 : PB, rows = n - j0;
    PAR_FOR(idx, rows * PB) { const int r = idx / PB, k = idx % PB; P[r * LP + k] = (k < jb && k <= r) ? Sg[(long long)(j0 + r) * ld + j0 + k] : 0.0; }
    CHD_SYNC();
    for (int k = 0; k < jb; ++k) {
      const double d = pivot_fix(c, P[k * LP + k], sign[j0 + k]);
      const double id = 1.0 / d;
      CHD_SYNC();                                  // (everybody has read the pivot before thread 0 replaces it)
      if (CHD_TID == 0) P[k * LP + k] = d;
      // rows below the pivot: l = a / d, and the row's remaining panel columns lose l d l_kk'
      PAR_FOR(r0, rows - k - 1) {
        const int r = k + 1 + r0;
        const double l = P[r * LP + k] * id;
        for (int kk = k + 1; kk < jb && kk <= r; ++kk) P[r * LP + kk] -= l * P[kk * LP + k];      // P[kk][k] still holds a_kk,k = l_kk,k d (the columns are scaled after the loop)
      }
      CHD_SYNC();
    }
    // scale the columns: L = A / d
    PAR_FOR(idx, rows * PB) { const int r = idx / PB, k = idx % PB; if (k < jb && k < r) P[r * LP + k] /= P[k * LP + k]; }
    CHD_SYNC();
    PAR_FOR(idx, rows * PB) { const int r = idx / PB, k = idx % PB; if (k < jb && k <= r) Sg[(long long)(j0 + r) * ld + j0 + k] = P[r * LP + k]; }
    // trailing update of the lower triangle below / right of the panel: a wavefront per row, its lanes along the row (consecutive addresses); the row's
    // own L d is the same for every lane (LDS broadcast), the other factor is a column walk through the padded panel.  Columns >= jb of a short last
    // panel hold zeros.
    const int nt = rows - jb;
    for (int r = CHD_WAVE_ID; r < nt; r += CHD_NWAVES) {
      const LdsD* lr = P + (long long)(jb + r) * LP;
      SP dst = Sg + (long long)(j0 + jb + r) * ld + j0 + jb;
      for (int cc = CHD_LANE; cc <= r; cc += CHD_WAVE_SZ) {
        const LdsD* lc = P + (long long)(jb + cc) * LP;
        double acc = 0;
        for (int k = 0; k < jb; ++k) acc += lr[k] * P[k * LP + k] * lc[k];
        dst[cc] -= acc;
      }
    }
    CHD_SYNC();
  }
}

CHD_NOINLINE CHD_DEV void kfactor_rl(LCtx& c, const GD* diag, const GI* sign) {
  TIC();
  const int Nb = c.Nb, w = c.w, W2 = c.W2, W1 = c.w + 1, LD = c.LD, bc = c.bc;
  if (CHD_TID == 0) c.n_bad_pivots = 0;
  // K0 + diag -> the factor storage, lower triangle: the factor's envelope [efirst_i, i] of each row is written in full (the factorisation fills it in), but
  // only the marked entries of K0 are read: zeros (the shift on the diagonal) first, then the list's entries on top
  {
    constexpr int RP = 8;                              // rows per wavefront pass: their envelope starts and shifts are fetched a pass ahead, the stores of a pass are independent
    int cnext[RP]; double dnext[RP];
#pragma unroll
    for (int r = 0; r < RP; ++r) { const int i = CHD_WAVE_ID * RP + r < Nb ? CHD_WAVE_ID * RP + r : Nb - 1; cnext[r] = c.env[2 * i] - i + w; dnext[r] = diag[i]; }
    for (int i0 = CHD_WAVE_ID * RP; i0 < Nb; i0 += CHD_NWAVES * RP) {
      int clo[RP]; double dg[RP];
#pragma unroll
      for (int r = 0; r < RP; ++r) {
        clo[r] = cnext[r]; dg[r] = dnext[r];
        const int in = i0 + CHD_NWAVES * RP + r < Nb ? i0 + CHD_NWAVES * RP + r : Nb - 1;
        cnext[r] = c.env[2 * in] - in + w; dnext[r] = diag[in];
      }
#ifdef CHD_HOST_EMU
      for (int r = 0; r < RP && i0 + r < Nb; ++r)
        for (int cc = 0; cc < clo[r]; ++cc)
          if (c.K0b[(long long)(i0 + r) * W2 + cc] != 0.0 || c.Kfb[(long long)(i0 + r) * W1 + cc] != 0.0) { c.err = 2; std::fprintf(stderr, "band envelope violated: row %d col offset %d first %d (K0 %g Kf %g)\n", i0 + r, cc, clo[r], c.K0b[(long long)(i0 + r) * W2 + cc], c.Kfb[(long long)(i0 + r) * W1 + cc]); break; }      // structural-envelope self check (host only)
#endif
#pragma unroll
      for (int r = 0; r < RP; ++r) {
        if (i0 + r >= Nb) break;
        GD* dst = c.Kfb + (long long)(i0 + r) * W1;
        for (int cc = clo[r] + CHD_LANE; cc < W1; cc += CHD_WAVE_SZ) dst[cc] = cc == w ? dg[r] : 0.0;
      }
    }
  }
  for (int r = CHD_WAVE_ID; r < bc; r += CHD_NWAVES) {
    GD* dst = c.Kfx + (long long)r * LD;
    const double dg = diag[Nb + r];
#ifdef CHD_HOST_EMU
    const GD* src = c.K0x + (long long)r * LD;
    for (int kk = 0; kk < c.env[2 * (Nb + r)]; ++kk) if (src[kk] != 0.0 || dst[kk] != 0.0) { c.err = 2; std::fprintf(stderr, "border structure violated: row %d col %d first %d\n", r, kk, c.env[2 * (Nb + r)]); break; }      // structural self check (host only)
#endif
    for (int k = c.env[2 * (Nb + r)] + CHD_LANE; k < LD; k += CHD_WAVE_SZ) dst[k] = k == Nb + r ? dg : 0.0;      // zero (in both) left of the first coupled band position
  }
  CHD_SYNC();
  PAR_FOR(e, c.csr_nnz) {
    const int row = c.csr_row[e], col = c.csr_col[e];
    if (row < Nb) {
      if (col <= row) c.Kfb[(long long)row * W1 + (col - row + w)] = c.K0b[(long long)row * W2 + (col - row + w)] + (col == row ? diag[row] : 0.0);
    } else {
      c.Kfx[(long long)(row - Nb) * LD + col] = c.K0x[(long long)(row - Nb) * LD + col] + (col == row ? diag[row] : 0.0);
    }
  }
  CHD_SYNC();
  TACC(c, 6, CHD_CLOCK() - tic_);
  // panel width from the LDS budget
  // the widest panel whose buffers fit: two (pivots, nb x nb diagonal block) pairs, the panel, the two active-row lists.  (The blocks are sized by the
  // panel width, not by its maximum of 32: with the 700-row border of a 600-frame sequence that is the difference between 16- and 8-column panels.)
  int nb = 32;
  while (nb > 8 && (long long)(nb + w + bc + 18) * nb > c.lds_cap - LDS_RED - 2 * (64 + nb * nb) - (w + bc + 66) - 8) nb >>= 1;
  LdsD* dv = c.lds + LDS_RED;            // pivots of the current panel (<= 32) and their reciprocals
  LdsD* DL = dv + 64;                    // dense copy of the panel's unit-lower diagonal block
  LdsD* dv2 = DL + nb * nb;              // the same pair for the next panel (look-ahead)
  LdsD* DL2 = dv2 + 64;
  LdsD* PT = DL2 + nb * nb;              // panel (the list of active window rows follows it)
  const int ldp = (nb + w + bc + 17) | 1;  // odd leading dimension (conflict-free column walks), >= 16 rows of zero padding
  PAR_FOR(i, ldp * nb) PT[i] = 0.0;          // rows a panel does not load (inactive, padding) must read as zero
  CHD_SYNC();
  if (nb == 32) kfactor_band<32>(c, sign, dv, DL, dv2, DL2, PT, ldp);
  else if (nb == 16) kfactor_band<16>(c, sign, dv, DL, dv2, DL2, PT, ldp);
  else kfactor_band<8>(c, sign, dv, DL, dv2, DL2, PT, ldp);
  // ---- dense L D L^T of the border Schur complement (rows/cols Nb..N-1)
  const long long td_ = CHD_CLOCK();
  if (bc > 0 && !(CHD_INERTIA_RETRY && CHD_ABORT_BAD_FACTOR && c.n_bad_pivots > 0)) {
    LdsD* SL = c.lds + LDS_RED;
#ifdef CHD_HOST_EMU
    const bool in_lds = (long long)bc * bc <= c.lds_cap - LDS_RED && !std::getenv("CHD_EMU_BORDER_IN_HBM");      // (tests: the blocked path for borders beyond LDS)
#else
    const bool in_lds = (long long)bc * bc <= c.lds_cap - LDS_RED;
#endif
    if (in_lds) {
      PAR_FOR(idx, bc * bc) { const int r = idx / bc, k = idx % bc; SL[idx] = k <= r ? c.Kfx[(long long)r * LD + Nb + k] : 0.0; }
      CHD_SYNC();
      dense_ldlt(c, SL, bc, bc, sign + Nb);
      PAR_FOR(idx, bc * bc) { const int r = idx / bc, k = idx % bc; if (k <= r) c.Kfx[(long long)r * LD + Nb + k] = SL[idx]; }
      CHD_SYNC();
    } else {
      // The blocked form is for the long sequences' borders (600 frames: 700 rows, 23 -> 5.6 ms per factorisation).  Borders just beyond LDS (100-frame clips:
      // ~160 rows) gain a few per cent at most, and the different summation order moves the chaotic clips of profiles/r04_curvature_study.md (section 7: one
      // clip's duration stage 501 -> 1 681 iterations): they keep the column-at-a-time form, and with it the arithmetic the round's studies were run with.
      int pb = 16;
      while (pb > 4 && (long long)bc * (pb + 1) > c.lds_cap - LDS_RED) pb >>= 1;
#ifdef CHD_HOST_EMU
      const bool long_border = bc > CHD_BLOCKED_BORDER_MIN || std::getenv("CHD_EMU_BORDER_IN_HBM") != nullptr;      // (the test forces the blocked form on a short border)
#else
      const bool long_border = bc > CHD_BLOCKED_BORDER_MIN;
#endif
      if (long_border && (long long)bc * (pb + 1) <= c.lds_cap - LDS_RED) dense_ldlt_blocked(c, c.Kfx + Nb, LD, bc, sign + Nb, SL, pb);
      else dense_ldlt(c, c.Kfx + Nb, LD, bc, sign + Nb);
    }
  }
  TACC(c, 12, CHD_CLOCK() - td_);
  TOC(c, 2);
}


// (Rounds 2-3 carried two alternative factorisations -- left-looking matrix-core tiles gathered from the factor storage, and a frontal one with the front in the
//  accumulator registers; both correct, both measured slower on the MI355X (482 and 409 against 629 sequences/s, profiles/r03a_merit_clip, r03b_register_front)
//  and removed in round 4.  chd_config.factorisation is kept in the ABI and ignored.)
CHD_NOINLINE CHD_DEV void kfactor(LCtx& c, const GD* diag, const GI* sign) { kfactor_rl(c, diag, sign); }

// in-block triangular solves for the substitution (wave-cooperative on the device)
#define CHD_SOLVE_NB 64
#define CHD_TILE_LD 65
#ifdef CHD_HOST_EMU
template <class YP>
CHD_DEV void tri_forward(LCtx& c, YP y, int c0, int jb, const LdsD*) {
  const int W1 = c.w + 1, w = c.w;
  for (int i = 1; i < jb; ++i) {
    double s = y[c0 + i];
    for (int j = 0; j < i; ++j) s -= c.Kfb[(long long)(c0 + i) * W1 + (j - i + w)] * y[c0 + j];
    y[c0 + i] = s;
  }
}
template <class YP>
CHD_DEV void tri_backward(LCtx& c, YP y, int c0, int jb, const LdsD*) {
  const int W1 = c.w + 1, w = c.w;
  for (int i = jb - 2; i >= 0; --i) {
    double s = y[c0 + i];
    for (int j = i + 1; j < jb; ++j) s -= c.Kfb[(long long)(c0 + j) * W1 + (i - j + w)] * y[c0 + j];
    y[c0 + i] = s;
  }
}
#else
// lane i owns row c0+i of the 64x64 diagonal block; its entries are fetched up front (independent loads) so that
// the dependent chain below runs out of registers
template <class YP>
CHD_NOINLINE CHD_DEV void tri_forward(LCtx& c, YP y, int c0, int jb, const LdsD* tile) {
  if (threadIdx.x < 64) {
    const int W1 = c.w + 1, w = c.w, i = threadIdx.x;
    const bool act = i < jb;
    double yi = act ? y[c0 + i] : 0.0;
    double l[CHD_SOLVE_NB];
    if (tile) {
#pragma unroll
      for (int j = 0; j < CHD_SOLVE_NB; ++j) l[j] = (act && j < i) ? tile[i * CHD_TILE_LD + j] : 0.0;
    } else {
      const GD* row = c.Kfb + (long long)(c0 + (act ? i : 0)) * W1 + (w - (act ? i : 0));
#pragma unroll
      for (int j = 0; j < CHD_SOLVE_NB; ++j) l[j] = (act && j < i) ? row[j] : 0.0;
    }
#pragma unroll
    for (int j = 0; j < CHD_SOLVE_NB - 1; ++j) {
      const double yj = readlane_f64(yi, j);      // j is a constant after unrolling: v_readlane, no LDS round trip
      yi -= l[j] * yj;
    }
    if (act) y[c0 + i] = yi;
  }
}
template <class YP>
CHD_NOINLINE CHD_DEV void tri_backward(LCtx& c, YP y, int c0, int jb, const LdsD* tile) {
  if (threadIdx.x < 64) {
    const int W1 = c.w + 1, w = c.w, i = threadIdx.x;
    const bool act = i < jb;
    double yi = act ? y[c0 + i] : 0.0;
    double l[CHD_SOLVE_NB];
    if (tile) {
#pragma unroll
      for (int j = 0; j < CHD_SOLVE_NB; ++j) l[j] = (act && j > i && j < jb) ? tile[j * CHD_TILE_LD + i] : 0.0;
    } else {
#pragma unroll
      for (int j = 0; j < CHD_SOLVE_NB; ++j) l[j] = (act && j > i && j < jb) ? c.Kfb[(long long)(c0 + j) * W1 + (i - j + w)] : 0.0;
    }
#pragma unroll
    for (int j = CHD_SOLVE_NB - 1; j > 0; --j) {
      const double yj = readlane_f64(yi, j);      // j is a constant after unrolling: v_readlane, no LDS round trip
      yi -= l[j] * yj;
    }
    if (act) y[c0 + i] = yi;
  }
}
#endif

// strictly-lower part of the diagonal block of `Kf` starting at c0 -> LDS tile (coalesced along the rows)
CHD_DEV void load_diag_tile(LCtx& c, LdsD* tile, int c0, int jb) {
  const int W1 = c.w + 1, w = c.w;
  PAR_FOR(t, CHD_SOLVE_NB * 8) {            // 8 tasks per row, 8 consecutive entries each, loads issued together
    const int a = t >> 3, j0 = (t & 7) << 3;
    if (a >= jb || j0 >= a) continue;
    const GD* src = c.Kfb + (long long)(c0 + a) * W1 + (j0 - a + w);
    double v[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) v[q] = (j0 + q < a) ? src[q] : 0.0;
#pragma unroll
    for (int q = 0; q < 8; ++q) if (j0 + q < a) tile[a * CHD_TILE_LD + j0 + q] = v[q];
  }
}

// ---- pieces of the substitution.  `wv0`/`nwv`: the wavefronts [wv0, wv0 + nwv) take part (the first wavefront is
// busy with a triangular block meanwhile when wv0 == 1)
#ifdef CHD_HOST_EMU
#define CHD_REST_W0 0
#define CHD_REST_NW 1
#else
#define CHD_REST_W0 1
#define CHD_REST_NW (CHD_NWAVES - 1)
#endif
// y[i] -= sum_{k in [kbeg, kend)} L(i, k) y[k] for the rows i = r0 .. r0 + nr - 1 (clipped to each row's envelope)
template <class YP>
CHD_DEV void fwd_rows_dot(LCtx& c, YP y, const int r0, const int nr, const int kbeg, const int kend, const int wv0, const int nwv) {
  const int W1 = c.w + 1, w = c.w;
  const int gpw = CHD_WAVE_SZ / CHD_GL;                    // lane groups per wavefront
  if (CHD_WAVE_ID < wv0 || CHD_WAVE_ID >= wv0 + nwv) return;
  const int lane_ = CHD_LANE % CHD_GL;
  for (int a = (CHD_WAVE_ID - wv0) * gpw + CHD_LANE / CHD_GL; a < nr; a += nwv * gpw) {
    const int i = r0 + a;
    const int lo = c.env[2 * i] > kbeg ? c.env[2 * i] : kbeg;          // L(i, k) = 0 left of the envelope
    const GD* row = c.Kfb + (long long)i * W1 + (w - i);
    const double acc = group_sum(dot_strided(row, y, lo + lane_, kend, CHD_GL));
    if (lane_ == 0) y[i] -= acc;
  }
}
// y[k] -= sum_a L(c0 + a, k) y[c0 + a] for the columns k in [kbeg, kend), rows of the block at c0:
// two half-waves per 32 columns (rows a < 32 and a >= 32 of the block), combined with one xor-32 shuffle
template <class YP>
CHD_DEV void bwd_cols_scatter(LCtx& c, YP y, const int c0, const int jb, const int kbeg, const int kend, const int wv0, const int nwv) {
  const int W1 = c.w + 1, w = c.w;
  if (CHD_WAVE_ID < wv0 || CHD_WAVE_ID >= wv0 + nwv) return;
  const int ncol = kend - kbeg, cpw = CHD_WAVE_SZ / CHD_PAIR;
  for (int base = (CHD_WAVE_ID - wv0) * cpw; base < ncol; base += nwv * cpw) {
    const int kk = base + CHD_LANE % cpw, half = CHD_LANE / cpw;
    const bool live = kk < ncol;
    const int k = kbeg + (live ? kk : 0);
    // rows i = c0 + a reach column k while i - k <= w
    int amax = (w - (c0 - k)) < jb - 1 ? (w - (c0 - k)) : jb - 1;
    if (c.env[2 * k + 1] - c0 < amax) amax = c.env[2 * k + 1] - c0;      // no row beyond this one reaches column k
    const GD* col = c.Kfb + (long long)c0 * W1 + (k - c0 + w);      // L(c0 + a, k) = col[a * (W1 - 1)]
    const int a0 = CHD_PAIR == 2 ? half * (CHD_SOLVE_NB / 2) : 0;
    const int aend = CHD_PAIR == 2 ? (half == 0 ? (amax < CHD_SOLVE_NB / 2 - 1 ? amax : CHD_SOLVE_NB / 2 - 1) : amax) : amax;
    const double acc = pair_sum(dot_column(col, W1 - 1, y + c0, a0, aend + 1));
    if (half == 0 && live) y[k] -= acc;
  }
}

// x = K^{-1} rhs using the factor.  y: work vector of N doubles, S: the dense border factor (both in LDS when they fit).
// The 64x64 triangular blocks are a one-wavefront dependent chain; the other wavefronts spend that time on the part
// of the next block's update that does not need the chain's result.
template <class YP, class SP>
CHD_DEV void ksolve_impl(LCtx& c, const GD* rhs, GD* x, YP y, SP Sp, const int lds_, const bool stage_s, LdsD* tile) {
  const int Nb = c.Nb, w = c.w, W1 = c.w + 1, LD = c.LD, bc = c.bc, N = c.N;
  PAR_FOR(i, N) y[i] = rhs[i];
  if (stage_s) PAR_FOR(idx, bc * bc) { const int r = idx / bc, k = idx % bc; Sp[idx] = c.Kfx[(long long)r * LD + Nb + k]; }
  const int nb = CHD_SOLVE_NB;
  const int nblk = (Nb + nb - 1) / nb;
  if (tile) load_diag_tile(c, tile, 0, Nb < nb ? Nb : nb);
  CHD_SYNC();
  long long ts_ = CHD_CLOCK();
  // forward, band
  for (int bk = 0; bk < nblk; ++bk) {
    const int c0 = bk * nb, c1 = c0 + nb;
    const int jb = Nb - c0 < nb ? Nb - c0 : nb;
    const int jb1 = bk + 1 < nblk ? (Nb - c1 < nb ? Nb - c1 : nb) : 0;
    ts_ = CHD_CLOCK();
    tri_forward(c, y, c0, jb, tile);                                                  // first wavefront
    if (jb1 > 0 && c0 > 0) fwd_rows_dot(c, y, c1, jb1, 0, c0, CHD_REST_W0, CHD_REST_NW);     // the others: columns left of this block
    CHD_SYNC();
    TACC(c, 16, CHD_CLOCK() - ts_); ts_ = CHD_CLOCK();
    if (jb1 > 0) {
      fwd_rows_dot(c, y, c1, jb1, c0, c1, 0, CHD_NWAVES);                             // columns of the block just solved
      if (tile) load_diag_tile(c, tile, c1, jb1);
    }
    CHD_SYNC();
    TACC(c, 17, CHD_CLOCK() - ts_);
  }
  ts_ = CHD_CLOCK();
  // forward, border rows: band part of L_border
  GROUP_FOR(r, bc) {
    const GD* row = c.Kfx + (long long)r * LD;
    const double acc = group_sum(dot_strided(row, y, c.env[2 * (Nb + r)] + lane_, Nb, CHD_GL));
    if (lane_ == 0) y[Nb + r] -= acc;
  }
  CHD_SYNC();
  // dense unit-lower part of the border: one wavefront, ordered through LDS (no workgroup barrier per column)
  if (CHD_WAVE0) {
    for (int j = 0; j + 1 < bc; ++j) {
      const double yj = y[Nb + j];
      for (int r = j + 1 + CHD_WLANE; r < bc; r += CHD_WSTEP) y[Nb + r] -= Sp[(long long)r * lds_ + j] * yj;
      CHD_WSYNC();
    }
  }
  CHD_SYNC();
  // diagonal
  PAR_FOR(i, N) y[i] /= (i < Nb ? c.Kfb[(long long)i * W1 + w] : c.Kfx[(long long)(i - Nb) * LD + i]);
  CHD_SYNC();
  // backward, border
  if (CHD_WAVE0) {
    for (int j = bc - 1; j > 0; --j) {
      const double yj = y[Nb + j];
      for (int r = CHD_WLANE; r < j; r += CHD_WSTEP) y[Nb + r] -= Sp[(long long)j * lds_ + r] * yj;
      CHD_WSYNC();
    }
  }
  CHD_SYNC();
  PAR_FOR(k, Nb) {
    y[k] -= dot_column(c.Kfx + k, LD, y + Nb, 0, c.rcnt[k]);
  }
  if (tile) load_diag_tile(c, tile, (nblk - 1) * nb, Nb - (nblk - 1) * nb);
  CHD_SYNC();
#ifndef CHD_EVAL_TIMING
  TACC(c, 18, CHD_CLOCK() - ts_);
#endif
  // backward, band
  tri_backward(c, y, (nblk - 1) * nb, Nb - (nblk - 1) * nb, tile);
  CHD_SYNC();
  for (int bk = nblk - 1; bk >= 1; --bk) {
    const int c0 = bk * nb, cp = c0 - nb;
    const int jb = Nb - c0 < nb ? Nb - c0 : nb;
    const int k0 = c0 - w < 0 ? 0 : c0 - w;
    ts_ = CHD_CLOCK();
    bwd_cols_scatter(c, y, c0, jb, k0 > cp ? k0 : cp, c0, 0, CHD_NWAVES);             // into the previous block
    if (tile) load_diag_tile(c, tile, cp, nb);
    CHD_SYNC();
    TACC(c, 19, CHD_CLOCK() - ts_); ts_ = CHD_CLOCK();
    tri_backward(c, y, cp, nb, tile);                                                 // first wavefront
    if (k0 < cp) bwd_cols_scatter(c, y, c0, jb, k0, cp, CHD_REST_W0, CHD_REST_NW);    // the others: columns further left
    CHD_SYNC();
    TACC(c, 20, CHD_CLOCK() - ts_);
  }
  PAR_FOR(i, N) x[i] = y[i];
  CHD_SYNC();
}

#ifndef CHD_HOST_EMU
// ---- device fast path of the substitution (y, the packed border factor and one 64x64 tile in LDS; 64 <= w <= 384).
// Every entry of L that a block step needs is requested one block step ahead and waits in registers, so a step is
// LDS traffic, FMAs and two barriers instead of two dependent HBM round trips.
//   forward, block B:  rows of B  x  columns left of block B-1   ("far",  lane groups of 16: rows g and g + 32)
//                      rows of B  x  columns of block B-1        ("near", 8 lanes per row, 8 columns each)
//   backward, block B: columns of block B-1  x  rows of B        ("near", 8 columns per wavefront, 8 rows per lane)
//                      columns left of B-1   x  rows of B        ("far",  16 columns x 4 row quarters per wavefront pass)
#define CHD_NQ 20
#define CHD_BP 3
CHD_DEV void tri_chain_fwd(LdsD* y, const LdsD* tile, const int c0, const int jb) {
  const int i = threadIdx.x;
  const bool act = i < jb;
  double yi = act ? y[c0 + i] : 0.0;
#pragma unroll
  for (int j0 = 0; j0 < CHD_SOLVE_NB; j0 += 8) {
    double l8[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) l8[q] = (act && j0 + q < i) ? tile[i * CHD_TILE_LD + j0 + q] : 0.0;
#pragma unroll
    for (int q = 0; q < 8; ++q) if (j0 + q < CHD_SOLVE_NB - 1) yi -= l8[q] * readlane_f64(yi, j0 + q);
    CHD_SCHED_FENCE();
  }
  if (act) y[c0 + i] = yi;
}
CHD_DEV void tri_chain_bwd(LdsD* y, const LdsD* tile, const int c0, const int jb) {
  const int i = threadIdx.x;
  const bool act = i < jb;
  double yi = act ? y[c0 + i] : 0.0;
#pragma unroll
  for (int j0 = CHD_SOLVE_NB - 8; j0 >= 0; j0 -= 8) {
    double l8[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) l8[q] = (act && j0 + q > i && j0 + q < jb) ? tile[(j0 + q) * CHD_TILE_LD + i] : 0.0;
#pragma unroll
    for (int q = 7; q >= 0; --q) if (j0 + q > 0) yi -= l8[q] * readlane_f64(yi, j0 + q);
    CHD_SCHED_FENCE();
  }
  if (act) y[c0 + i] = yi;
}
CHD_DEV void ksolve_fast(LCtx& c, const GD* rhs, GD* x, LdsD* y, LdsD* Sp, LdsD* tile) {
  const int Nb = c.Nb, w = c.w, W1 = c.w + 1, LD = c.LD, bc = c.bc, N = c.N;
  const int nb = CHD_SOLVE_NB, nblk = (Nb + nb - 1) / nb;
  const int tid = threadIdx.x, wv = tid >> 6, lane = tid & 63;
  const int g16 = (tid - 64) >> 4, l16 = tid & 15;      // far parts: the 28 lane groups of wavefronts 1..7 (wavefront 0 runs the triangular chain)
  const int r8 = tid >> 3, c8 = (tid & 7) << 3;
  const GD* Kfb = c.Kfb;
  const GD* safe = c.Kfb + w;            // a valid address for the predicated-off requests (loads are never branched around)
  PAR_FOR(i, N) y[i] = rhs[i];
  PAR_FOR(idx, bc * bc) { const int r = idx / bc, k = idx % bc; if (k <= r) Sp[r * (r + 1) / 2 + k] = c.Kfx[(long long)r * LD + Nb + k]; }
  double fr[3][CHD_NQ], nr[8], tl[8];
  int flo[3] = {0, 0, 0}, fmk[3] = {0, 0, 0};      // envelope starts / chunk masks the held far entries were requested with
  const GI* env = c.env;
  int elo[3] = {0, 0, 0};       // envelope starts of the rows whose far part is requested next (fetched one step ahead as well)
  // -- requests (block index B; all predicated, so B may run past the last block)
#define CHD_LOAD_TILE(B) do { const int c0_ = (B) * nb, jb_ = Nb - c0_ < nb ? Nb - c0_ : nb; \
    const bool ok_ = (B) >= 0 && (B) < nblk && r8 < jb_; \
    const GD* row_ = Kfb + (long long)(ok_ ? c0_ + r8 : 0) * W1 + (w - r8); \
    _Pragma("unroll") for (int q = 0; q < 8; ++q) { const bool v_ = ok_ && c8 + q < r8; tl[q] = *(v_ ? row_ + c8 + q : safe); } } while (0)
#define CHD_WRITE_TILE() do { _Pragma("unroll") for (int q = 0; q < 8; ++q) if (c8 + q < r8) tile[r8 * CHD_TILE_LD + c8 + q] = tl[q]; } while (0)
#define CHD_LOAD_FAR(B) do { const int c0_ = (B) * nb, jb_ = Nb - c0_ < nb ? Nb - c0_ : nb, kend_ = c0_ - nb; \
    _Pragma("unroll") for (int p = 0; p < 3; ++p) { const int a_ = g16 + 28 * p, i_ = c0_ + a_; \
      const bool ok_ = wv > 0 && (B) < nblk && a_ < jb_ && kend_ > 0; \
      const int lo_ = elo[p]; flo[p] = lo_; \
      const GD* row_ = Kfb + (long long)(ok_ ? i_ : 0) * W1 + (w - (ok_ ? i_ : 0)); \
      int m_ = 0; \
      _Pragma("unroll") for (int ch = 0; ch < CHD_NQ / 4; ++ch) {        /* 64-column chunks nobody in the wavefront needs are skipped */ \
        if (__any(ok_ && lo_ + l16 + 64 * ch < kend_)) { m_ |= 1 << ch; \
          _Pragma("unroll") for (int q = 4 * ch; q < 4 * ch + 4; ++q) { const int k_ = lo_ + l16 + 16 * q; const bool v_ = ok_ && k_ < kend_; \
            fr[p][q] = *(v_ ? row_ + k_ : safe); } } } \
      fmk[p] = m_; \
      const int in_ = i_ + nb; elo[p] = env[(wv > 0 && (B) + 1 < nblk && a_ < nb && in_ < Nb) ? 2 * in_ : 0]; } } while (0)
#define CHD_USE_FAR(B) do { const int c0_ = (B) * nb, jb_ = Nb - c0_ < nb ? Nb - c0_ : nb, kend_ = c0_ - nb; \
    _Pragma("unroll") for (int p = 0; p < 3; ++p) { const int a_ = g16 + 28 * p; \
      double s0_ = 0, s1_ = 0; \
      _Pragma("unroll") for (int ch = 0; ch < CHD_NQ / 4; ++ch) if ((fmk[p] >> ch) & 1) { \
        _Pragma("unroll") for (int q = 4 * ch; q < 4 * ch + 4; q += 2) { const int k_ = flo[p] + l16 + 16 * q; \
          s0_ += fr[p][q] * ((a_ < jb_ && k_ < kend_) ? y[k_] : 0.0); s1_ += fr[p][q + 1] * ((a_ < jb_ && k_ + 16 < kend_) ? y[k_ + 16] : 0.0); } } \
      const double acc_ = group_sum(s0_ + s1_); \
      if (l16 == 0 && wv > 0 && a_ < jb_ && kend_ > 0) y[c0_ + a_] -= acc_; } } while (0)
#define CHD_LOAD_NEAR(B) do { const int c0_ = (B) * nb, jb_ = Nb - c0_ < nb ? Nb - c0_ : nb; \
    const bool ok_ = (B) < nblk && r8 < jb_; const int i_ = c0_ + r8, kb_ = c0_ - nb + c8; \
    const GD* row_ = Kfb + (long long)(ok_ ? i_ : 0) * W1 + (w - (ok_ ? i_ : 0)); \
    _Pragma("unroll") for (int q = 0; q < 8; ++q) { const bool v_ = ok_ && i_ - (kb_ + q) <= w; nr[q] = *(v_ ? row_ + kb_ + q : safe); } } while (0)
#define CHD_USE_NEAR(B) do { const int c0_ = (B) * nb, jb_ = Nb - c0_ < nb ? Nb - c0_ : nb, kb_ = c0_ - nb + c8; \
    double s_ = 0; _Pragma("unroll") for (int q = 0; q < 8; ++q) s_ += nr[q] * ((r8 < jb_ && c0_ + r8 - (kb_ + q) <= w) ? y[kb_ + q] : 0.0); \
    s_ += __shfl_xor(s_, 4); s_ += __shfl_xor(s_, 2); s_ += __shfl_xor(s_, 1); \
    if ((tid & 7) == 0 && r8 < jb_) y[c0_ + r8] -= s_; } while (0)
  CHD_LOAD_TILE(0);
  CHD_WRITE_TILE();
  CHD_LOAD_FAR(0); CHD_LOAD_FAR(1); CHD_LOAD_NEAR(1); CHD_LOAD_TILE(1);      // (FAR(0), FAR(1) are empty: they only fetch the envelope starts)
  CHD_SYNC();
  long long ts_ = CHD_CLOCK(), t16_ = 0, t17_ = 0, t19_ = 0, t20_ = 0;     // (kept in registers: a store to the context would wait for the requests in flight)
  // forward, band
  for (int bk = 0; bk < nblk; ++bk) {
    const int c0 = bk * nb, jb = Nb - c0 < nb ? Nb - c0 : nb;
    ts_ = CHD_CLOCK();
    if (wv == 0) tri_chain_fwd(y, tile, c0, jb);
    else {
      if (bk + 1 < nblk) CHD_USE_FAR(bk + 1);
      CHD_LOAD_FAR(bk + 2);
    }
    CHD_SYNC();
    t16_ += CHD_CLOCK() - ts_; ts_ = CHD_CLOCK();
    if (bk + 1 < nblk) { CHD_USE_NEAR(bk + 1); CHD_WRITE_TILE(); }
    CHD_LOAD_NEAR(bk + 2); CHD_LOAD_TILE(bk + 2);
    CHD_SYNC();
    t17_ += CHD_CLOCK() - ts_;
  }
  ts_ = CHD_CLOCK();
  // requests for the first backward steps travel while the border is processed
  double bf[CHD_BP][16], bn[8];
  int bmk[CHD_BP] = {0, 0, 0};
  int ecl[CHD_BP], bcl[CHD_BP];        // last row reaching each far column (fetched one step ahead) / the value the held entries were masked with
  const int rg = lane >> 3, cl8 = lane & 7, qd = lane >> 4, cl16 = lane & 15;
#define CHD_LOAD_BNEAR(B) do { const int c0_ = (B) * nb, jb_ = Nb - c0_ < nb ? Nb - c0_ : nb; \
    const int k_ = c0_ - nb + 8 * wv + cl8; \
    _Pragma("unroll") for (int q = 0; q < 8; ++q) { const int a_ = rg * 8 + q, i_ = c0_ + a_; \
      const bool v_ = (B) >= 1 && a_ < jb_ && i_ - k_ <= w; bn[q] = *(v_ ? Kfb + (long long)i_ * W1 + (k_ - i_ + w) : safe); } } while (0)
#define CHD_USE_BNEAR(B) do { const int c0_ = (B) * nb, jb_ = Nb - c0_ < nb ? Nb - c0_ : nb; \
    const int kn_ = c0_ - nb + 8 * wv + cl8; \
    double s_ = 0; _Pragma("unroll") for (int q = 0; q < 8; ++q) { const int a_ = rg * 8 + q; s_ += bn[q] * ((a_ < jb_ && c0_ + a_ - kn_ <= w) ? y[c0_ + a_] : 0.0); } \
    s_ += __shfl_xor(s_, 8); s_ += __shfl_xor(s_, 16); s_ += __shfl_xor(s_, 32); \
    if (rg == 0) y[c0_ - nb + 8 * wv + cl8] -= s_; } while (0)
#define CHD_LOAD_BFAR(B) do { const int c0_ = (B) * nb, jb_ = Nb - c0_ < nb ? Nb - c0_ : nb; \
    const int k0_ = c0_ - w < 0 ? 0 : c0_ - w, kend_ = c0_ - nb; \
    const int c0n_ = c0_ - nb, k0n_ = c0n_ - w < 0 ? 0 : c0n_ - w; \
    _Pragma("unroll") for (int p = 0; p < CHD_BP; ++p) { const int k_ = k0_ + (p * 7 + wv - 1) * 16 + cl16; \
      const bool okc_ = wv > 0 && (B) >= 1 && k_ < kend_; \
      const int cl_ = ecl[p]; bcl[p] = cl_; \
      int m_ = 0; \
      _Pragma("unroll") for (int ch = 0; ch < 4; ++ch) {      /* validity falls with the row: a 4-row chunk is needed iff its first row is, for some lane */ \
        const int a0_ = qd * 16 + 4 * ch; \
        if (__any(okc_ && a0_ < jb_ && c0_ + a0_ - k_ <= w && c0_ + a0_ <= cl_)) { m_ |= 1 << ch; \
          _Pragma("unroll") for (int r = 4 * ch; r < 4 * ch + 4; ++r) { const int a_ = qd * 16 + r, i_ = c0_ + a_; \
            const bool v_ = okc_ && a_ < jb_ && i_ - k_ <= w && i_ <= cl_; bf[p][r] = *(v_ ? Kfb + (long long)i_ * W1 + (k_ - i_ + w) : safe); } } } \
      bmk[p] = m_; \
      const int kn_ = k0n_ + (p * 7 + wv - 1) * 16 + cl16; ecl[p] = env[(wv > 0 && (B) >= 2 && kn_ < c0n_ - nb) ? 2 * kn_ + 1 : 1]; } } while (0)
#define CHD_USE_BFAR(B) do { const int c0_ = (B) * nb, jb_ = Nb - c0_ < nb ? Nb - c0_ : nb; \
    const int k0_ = c0_ - w < 0 ? 0 : c0_ - w, kend_ = c0_ - nb; \
    _Pragma("unroll") for (int p = 0; p < CHD_BP; ++p) { const int k_ = k0_ + (p * 7 + wv - 1) * 16 + cl16; \
      double s0_ = 0, s1_ = 0; \
      _Pragma("unroll") for (int ch = 0; ch < 4; ++ch) if ((bmk[p] >> ch) & 1) { \
        _Pragma("unroll") for (int r = 4 * ch; r < 4 * ch + 4; r += 2) { const int a_ = qd * 16 + r; \
          s0_ += bf[p][r] * ((k_ < kend_ && a_ < jb_ && c0_ + a_ - k_ <= w && c0_ + a_ <= bcl[p]) ? y[c0_ + a_] : 0.0); \
          s1_ += bf[p][r + 1] * ((k_ < kend_ && a_ + 1 < jb_ && c0_ + a_ + 1 - k_ <= w && c0_ + a_ + 1 <= bcl[p]) ? y[c0_ + a_ + 1] : 0.0); } } \
      double s_ = s0_ + s1_; s_ += __shfl_xor(s_, 16); s_ += __shfl_xor(s_, 32); \
      if (qd == 0 && wv > 0 && k_ < kend_) y[k_] -= s_; } } while (0)
  {
    const int c0_ = (nblk - 1) * nb, k0_ = c0_ - w < 0 ? 0 : c0_ - w;
#pragma unroll
    for (int p = 0; p < CHD_BP; ++p) { const int k_ = k0_ + (p * 7 + wv - 1) * 16 + cl16; ecl[p] = env[(wv > 0 && nblk >= 2 && k_ < c0_ - nb) ? 2 * k_ + 1 : 1]; bcl[p] = 0; }
  }
  CHD_LOAD_TILE(nblk - 1); CHD_LOAD_BNEAR(nblk - 1); CHD_LOAD_BFAR(nblk - 1);
  // forward, border rows: band part of L_border
  GROUP_FOR(r, bc) {
    const GD* row = c.Kfx + (long long)r * LD;
    const double acc = group_sum(dot_strided(row, y, c.env[2 * (Nb + r)] + lane_, Nb, CHD_GL));
    if (lane_ == 0) y[Nb + r] -= acc;
  }
  CHD_SYNC();
  // dense unit-lower part of the border: one wavefront, ordered through LDS (no workgroup barrier per column)
  if (wv == 0) {
    for (int j = 0; j + 1 < bc; ++j) {
      const double yj = y[Nb + j];
      for (int r = j + 1 + lane; r < bc; r += 64) y[Nb + r] -= Sp[r * (r + 1) / 2 + j] * yj;
      CHD_WSYNC();
    }
  }
  CHD_SYNC();
  // diagonal
  PAR_FOR(i, N) y[i] /= (i < Nb ? c.Kfb[(long long)i * W1 + w] : c.Kfx[(long long)(i - Nb) * LD + i]);
  CHD_SYNC();
  // backward, border
  if (wv == 0) {
    for (int j = bc - 1; j > 0; --j) {
      const double yj = y[Nb + j];
      for (int r = lane; r < j; r += 64) y[Nb + r] -= Sp[j * (j + 1) / 2 + r] * yj;
      CHD_WSYNC();
    }
  }
  CHD_SYNC();
  PAR_FOR(k, Nb) {
    y[k] -= dot_column(c.Kfx + k, LD, y + Nb, 0, c.rcnt[k]);
  }
  CHD_WRITE_TILE();
  CHD_SYNC();
#ifndef CHD_EVAL_TIMING
  TACC(c, 18, CHD_CLOCK() - ts_);
#endif
  // backward, band
  CHD_LOAD_TILE(nblk - 2);
  if (wv == 0) tri_chain_bwd(y, tile, (nblk - 1) * nb, Nb - (nblk - 1) * nb);
  CHD_SYNC();
  for (int bk = nblk - 1; bk >= 1; --bk) {
    ts_ = CHD_CLOCK();
    CHD_USE_BNEAR(bk);
    CHD_WRITE_TILE();
    CHD_LOAD_BNEAR(bk - 1); CHD_LOAD_TILE(bk - 2);
    CHD_SYNC();
    t19_ += CHD_CLOCK() - ts_; ts_ = CHD_CLOCK();
    if (wv == 0) tri_chain_bwd(y, tile, (bk - 1) * nb, nb);
    else {
      CHD_USE_BFAR(bk);
      CHD_LOAD_BFAR(bk - 1);
    }
    CHD_SYNC();
    t20_ += CHD_CLOCK() - ts_;
  }
  PAR_FOR(i, N) x[i] = y[i];
  CHD_SYNC();
#ifndef CHD_EVAL_TIMING
  TACC(c, 16, t16_); TACC(c, 17, t17_); TACC(c, 19, t19_); TACC(c, 20, t20_);
#endif
#undef CHD_LOAD_TILE
#undef CHD_WRITE_TILE
#undef CHD_LOAD_FAR
#undef CHD_USE_FAR
#undef CHD_LOAD_NEAR
#undef CHD_USE_NEAR
#undef CHD_LOAD_BNEAR
#undef CHD_USE_BNEAR
#undef CHD_LOAD_BFAR
#undef CHD_USE_BFAR
}
#endif

CHD_NOINLINE CHD_DEV void ksolve_once(LCtx& c, const GD* rhs, GD* x) {
  TIC();
  const int N = c.N, bc = c.bc, Npad = (N + 1) & ~1;
  const int room = c.lds_cap - LDS_RED;
  const int tsz = CHD_SOLVE_NB * CHD_TILE_LD + 1;
  LdsD* base = c.lds + LDS_RED;
#ifndef CHD_HOST_EMU
  const int spk = (bc * (bc + 1) / 2 + 1) & ~1;
  if (CHD_NT == 512 && room >= Npad + spk + tsz && c.w >= CHD_SOLVE_NB && c.w - CHD_SOLVE_NB <= 16 * CHD_NQ && c.w - CHD_SOLVE_NB <= CHD_BP * 112) {      // (lane-group layout of ksolve_fast: eight wavefronts)
    ksolve_fast(c, rhs, x, base, base + Npad, base + Npad + spk);
    TOC(c, 3);
    return;
  }
#endif
  if (room >= Npad + bc * bc + tsz) ksolve_impl(c, rhs, x, base, base + Npad, bc, true, base + Npad + bc * bc);
  else if (room >= Npad + bc * bc) ksolve_impl(c, rhs, x, base, base + Npad, bc, true, (LdsD*)nullptr);
  else if (room >= Npad + tsz) ksolve_impl(c, rhs, x, base, c.Kfx + c.Nb, c.LD, false, base + Npad);
  else if (room >= N) ksolve_impl(c, rhs, x, base, c.Kfx + c.Nb, c.LD, false, (LdsD*)nullptr);
  else ksolve_impl(c, rhs, x, VK(c, VK_Y), c.Kfx + c.Nb, c.LD, false, (LdsD*)nullptr);
  TOC(c, 3);
}

// K x = rhs with up to `refine` steps of iterative refinement; a step is taken only while the residual is above
// CHD_REFINE_SKIP x |rhs| (max norms): the L D L^T of the regularised KKT matrix leaves 1e-11 .. 2e-9 on the bench sequences
// (up to 1e-6 on the hard ones, which do get the step), and below the threshold the correction -- a second pair of substitutions,
// an eighth of the kernel's time -- does not change a single iteration count on the 201 fixture sequences
// (oracle/ipm_solver.hpp BorderedBandLDL::solve has the same rule).
#define CHD_REFINE_SKIP 1e-8
CHD_DEV void ksolve(LCtx& c, const GD* rhs, GD* x, const GD* diag, int refine) {
  ksolve_once(c, rhs, x);
  GD* t1 = VK(c, VK_T1); GD* t2 = VK(c, VK_T2);
  for (int it = 0; it < refine; ++it) {
    kmatvec(c, x, t1, diag, nullptr);
    double rn = 0.0, bn = 0.0;
    PAR_FOR(i, c.N) { const double b = rhs[i], r = b - t1[i]; t1[i] = r; rn = fmax(rn, fabs(r)); bn = fmax(bn, fabs(b)); }
    rn = block_max(c, rn); bn = block_max(c, bn);
    CHD_SYNC();
    if (rn <= CHD_REFINE_SKIP * bn) break;
    ksolve_once(c, t1, t2);
    PAR_FOR(i, c.N) x[i] += t2[i];
    CHD_SYNC();
  }
}

// ------------------------------------------------------------------------------------------
// NLP state <-> x
// ------------------------------------------------------------------------------------------
CHD_DEV void refresh_durations(QP q) {     // polynomial durations + cumulative times from the phase durations
  GD* wd = q->wd;
  PAR_FOR(idx, q->tot_polys) {
    int s = 0;
    while (s + 1 < N_SPLINES && idx >= q->sp[s + 1].poly_off) ++s;
    const auto& sp = q->sp[s];
    if (sp.phase_based) {
      const GI* pi = q->ci + q->o_pinfo + idx * 4;
      wd[q->o_poly_dur + idx] = wd[q->o_phase_dur + q->phase_off[sp.ee] + pi[0]] / pi[2];
    }
  }
  CHD_SYNC();
  PAR_FOR(s, N_SPLINES + N_EE) {
    if (s < N_SPLINES) {
      const auto& sp = q->sp[s];
      double t = 0;
      for (int p = 0; p < sp.n_polys; ++p) { t += wd[q->o_poly_dur + sp.poly_off + p]; wd[q->o_pend + sp.poly_off + p] = t; }
      wd[q->o_ttot + s] = t;
    } else {
      const int e = s - N_SPLINES;
      double t = 0;
      for (int p = 0; p < q->n_phase[e]; ++p) { t += wd[q->o_phase_dur + q->phase_off[e] + p]; wd[q->o_phend + q->phase_off[e] + p] = t; }
    }
  }
  CHD_SYNC();
  const int nph = q->phase_off[N_EE - 1] + q->n_phase[N_EE - 1];
  const bool fits = q->tot_polys <= CHD_PEND_CAP && nph <= CHD_PHEND_CAP;
  if (fits) {
    PAR_FOR(i, q->tot_polys) ((LdsD*)chd_pend_l)[i] = wd[q->o_pend + i];
    PAR_FOR(i, nph) ((LdsD*)chd_phend_l)[i] = wd[q->o_phend + i];
  }
  if (CHD_TID == 0) chd_tab_ok = fits ? 1 : 0;
  CHD_SYNC();
}

CHD_DEV void state_from_x(LCtx& c, const GD* x) {
  QP q = c.q;
  PAR_FOR(k, q->tot_entries) {
    const int v = q->ci[q->o_varof + k];
    if (v >= 0) {
      int s = 0;
      while (s + 1 < N_SPLINES && k >= q->sp[s + 1].node_off) ++s;
      q->wd[q->o_node + k] = x[q->sp[s].var_off + v];
    }
  }
  if (c.S->opt_dur) {
    PAR_FOR(e, N_EE) {      // TOWR PhaseDurations::SetVariables: last duration = T - sum
      double sum = 0;
      const int np = q->n_phase[e];
      GD* ph = q->wd + q->o_phase_dur + q->phase_off[e];
      for (int k = 0; k + 1 < np; ++k) { ph[k] = x[c.S->dur_off[e] + k]; sum += ph[k]; }
      ph[np - 1] = q->T - sum;
    }
  }
  CHD_SYNC();
  if (c.S->opt_dur) refresh_durations(q);
}

CHD_DEV void x_from_state(LCtx& c, GD* x) {
  QP q = c.q;
  PAR_FOR(k, q->tot_entries) {
    const int v = q->ci[q->o_varof + k];
    if (v >= 0) {
      int s = 0;
      while (s + 1 < N_SPLINES && k >= q->sp[s + 1].node_off) ++s;
      x[q->sp[s].var_off + v] = q->wd[q->o_node + k];
    }
  }
  if (c.S->opt_dur)
    PAR_FOR(e, N_EE)
      for (int k = 0; k + 1 < q->n_phase[e]; ++k) x[c.S->dur_off[e] + k] = q->wd[q->o_phase_dur + q->phase_off[e] + k];
  CHD_SYNC();
}

// ------------------------------------------------------------------------------------------
// Constraint rows
// ------------------------------------------------------------------------------------------
struct RowW {           // where one row's Jacobian entries go
  LCtx* c; int pr; double sc; bool on;
};
// The (up to) 12 Jacobian entries of one row w.r.t. the four Hermite coefficients x three dimensions of the active
// polynomial.  They are distinct KKT entries (after folding the two nodes of a stance pair, which share a variable), so
// their read-modify-writes are issued together: all index look-ups, then all loads, then all stores -- one HBM round
// trip per call instead of one (dependent) per entry.
// (its own function, arguments by value: inlined at its ~25 call sites the twelve-entry bodies made the row phases 70 k
//  instructions long -- more than the instruction cache -- and kept the spline samples they read in scratch memory)
CHD_NOINLINE CHD_DEV void row_nodes_nv(LCtx& c, const int pr, const double sc, const int s, const int poly, const double w0, const double w1, const double w2, const double w3,
                                       const double c0, const double c1, const double c2, const int dimmask) {
  QP q = c.q;
  const auto& sp = q->sp[s];
  const GI* vo = q->ci + q->o_varof + sp.node_off + poly * 6;
  int v[12]; double val[12];
#pragma unroll
  for (int i = 0; i < 12; ++i) {
    const int side = i / 6, dq = (i % 6) / 3, k = i % 3;
    const double wk = side == 0 ? (dq == 0 ? w0 : w1) : (dq == 0 ? w2 : w3);
    v[i] = ((dimmask >> k) & 1) ? vo[i] : -1;
    val[i] = sc * (k == 0 ? c0 : k == 1 ? c1 : c2) * wk;
  }
#pragma unroll
  for (int i = 0; i < 6; ++i)
    if (v[i] >= 0 && v[i] == v[6 + i]) { val[i] += val[6 + i]; v[6 + i] = -1; }      // stance pair: one variable, two nodes
  int pv[12];
#pragma unroll
  for (int i = 0; i < 12; ++i) pv[i] = c.pos_var[sp.var_off + (v[i] >= 0 ? v[i] : 0)];
  KSlot sl[12];
#pragma unroll
  for (int i = 0; i < 12; ++i) { sl[i].a = nullptr; sl[i].b = nullptr; if (v[i] >= 0) sl[i] = kslot(c, pr, pv[i]); }
  double oa[12], ob[12];
#pragma unroll
  for (int i = 0; i < 12; ++i) { oa[i] = *(sl[i].a ? sl[i].a : c.K0b); ob[i] = *(sl[i].b ? sl[i].b : c.K0b); }
#pragma unroll
  for (int i = 0; i < 12; ++i) { if (sl[i].a) K0_ADD(c, sl[i].a, oa[i], val[i]); if (sl[i].b) K0_ADD(c, sl[i].b, ob[i], val[i]); }
}
CHD_DEV void row_nodes(const RowW& r, int s, const PE& e, int which, const double coef[3], int dimmask) {
  if (!r.on) return;
  row_nodes_nv(*r.c, r.pr, r.sc, s, e.poly, e.w[which][0], e.w[which][1], e.w[which][2], e.w[which][3], coef[0], coef[1], coef[2], dimmask);
}
CHD_DEV void row_durs(const RowW& r, int s, double t, const PE& e, const double coef[3]) {
  if (!r.on || !r.c->S->opt_dur) return;
  QP q = r.c->q;
  DurJac dj;
  dur_jac(q, s, t, e, dj);
  const int ee = q->sp[s].ee;
  const int base = r.c->S->dur_off[ee];
  const double ve = coef[0] * dj.early[0] + coef[1] * dj.early[1] + coef[2] * dj.early[2];
  const double vo = coef[0] * dj.own[0] + coef[1] * dj.own[1] + coef[2] * dj.own[2];
  // entries (row, T_k) for k < cur (value ve) and k == cur (value vo, unless it is the dependent last duration):
  // distinct KKT entries, eight at a time with their look-ups / loads / stores issued together
  LCtx& c = *r.c;
  const int n_early = dj.cur < dj.nvar ? dj.cur : dj.nvar;
  const int n_ent = n_early + (dj.last ? 0 : 1);
  for (int k0 = 0; k0 < n_ent; k0 += 8) {
    int pv[8]; KSlot sl[8]; double oa[8], ob[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) { const int k = k0 + i; pv[i] = c.pos_var[base + (k < n_ent ? (k < n_early ? k : dj.cur) : 0)]; }
#pragma unroll
    for (int i = 0; i < 8; ++i) { sl[i].a = nullptr; sl[i].b = nullptr; if (k0 + i < n_ent) sl[i] = kslot(c, r.pr, pv[i]); }
#pragma unroll
    for (int i = 0; i < 8; ++i) { oa[i] = *(sl[i].a ? sl[i].a : c.K0b); ob[i] = *(sl[i].b ? sl[i].b : c.K0b); }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const double val = r.sc * (k0 + i < n_early ? ve : vo);
      if (sl[i].a) K0_ADD(c, sl[i].a, oa[i], val);
      if (sl[i].b) K0_ADD(c, sl[i].b, ob[i], val);
    }
  }
}

CHD_DEV int frame_index(QP q, double t) {       // humanoid_rigid_body_dynamics.cpp:81-87, leg_length_constraint.cpp:40-42
  int idx = (int)((t / q->T) * q->F);
  if (idx >= q->F) idx = q->F - 1;
  if (idx < 0) idx = 0;
  return idx;
}

// One dynamics sample (6 rows): humanoid_dynamic_constraint.cpp:63-143, humanoid_rigid_body_dynamics.cpp:89-206.
// The Jacobian of a sample is ~500 read-modify-writes of K0; done by one thread they form one dependent chain
// (~0.6 ms).  The sample is therefore split into 16 units that touch disjoint entries, one thread each:
//   unit 0: the six row values          units 1-3: base (linear + angular) columns of rows i = unit - 1
//   units 4-15: end-effector e = (unit - 4) / 3, rows i = (unit - 4) % 3 (force nodes, position nodes, durations;
//               the unit with i = 0 also stores the second-order duration terms of e)
// Every unit evaluates only the splines it needs.
CHD_ALWAYS_INLINE CHD_DEV void dyn_unit(LCtx& c, const int ti, const int unit, const bool D2, GD* cout_, const GD* lam, const GD* sc) {
  QP q = c.q; SDP S = c.S;
  const GI* tk = q->ci + S->o_task + 4 * ti;
  const int B = tk[2], row0 = tk[3];
  const double t = q->cd[S->o_task_t + ti];
  PE pl;
  spline_eval(q, 0, t, pl);
  if (unit < 4) {
    PE pa;
    spline_eval(q, 1, t, pa);
    double tau[3] = {0, 0, 0}, fsum[3] = {0, 0, 0};
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      PE pm, pf;
      spline_eval(q, 2 + e, t, pm); spline_eval(q, 6 + e, t, pf);
      double rr[3] = {pl.p[0] - pm.p[0], pl.p[1] - pm.p[1], pl.p[2] - pm.p[2]}, tq[3];
      cross3(pf.p, rr, tq);
      for (int k = 0; k < 3; ++k) { tau[k] += tq[k]; fsum[k] += pf.p[k]; }
    }
    const GD* I6 = q->cd + q->o_inertia + frame_index(q, t) * 6;
    const double Ib[3][3] = {{I6[0], I6[3], I6[4]}, {I6[3], I6[1], I6[5]}, {I6[4], I6[5], I6[2]}};   // humanoid_rigid_body_dynamics.cpp:47-56
    double ang[3], d0[3][3], d1[3][3], d2[3][3];
    angular_term(pa.p, pa.v, pa.a, Ib, unit > 0, ang, d0, d1, d2);
    if (unit == 0) {
      for (int k = 0; k < 3; ++k) {
        cout_[row0 + k] = sc[row0 + k] * (ang[k] - tau[k]);
        cout_[row0 + 3 + k] = sc[row0 + 3 + k] * (q->mass * pl.a[k] - fsum[k] - q->mass * CHD_G * q->gdir[k]);
      }
      return;
    }
    const int i = unit - 1;
    RowW ra{&c, c.pos_row[row0 + i], sc[row0 + i], true};
    RowW rl{&c, c.pos_row[row0 + 3 + i], sc[row0 + 3 + i], true};
    // (v x u)_i = v_{i1} u_{i2} - v_{i2} u_{i1}  ->  coefficients on u: [i2] = v_{i1}, [i1] = -v_{i2}   (cross_row)
    double cr_[3];
    cross_row(fsum, i, cr_);
    const double cf[3] = {-cr_[0], -cr_[1], -cr_[2]};           // -sum_e (f_e x dc)_i
    row_nodes(ra, 0, pl, 0, cf, 7 & ~(1 << i));
    const double cm[3] = {i == 0 ? q->mass : 0.0, i == 1 ? q->mass : 0.0, i == 2 ? q->mass : 0.0};
    row_nodes(rl, 0, pl, 2, cm, 1 << i);
    const double d0i[3] = {pick3(d0[0][0], d0[1][0], d0[2][0], i), pick3(d0[0][1], d0[1][1], d0[2][1], i), pick3(d0[0][2], d0[1][2], d0[2][2], i)};
    const double d1i[3] = {pick3(d1[0][0], d1[1][0], d1[2][0], i), pick3(d1[0][1], d1[1][1], d1[2][1], i), pick3(d1[0][2], d1[1][2], d1[2][2], i)};
    const double d2i[3] = {pick3(d2[0][0], d2[1][0], d2[2][0], i), pick3(d2[0][1], d2[1][1], d2[2][1], i), pick3(d2[0][2], d2[1][2], d2[2][2], i)};
    row_nodes(ra, 1, pa, 0, d0i, 7);
    row_nodes(ra, 1, pa, 1, d1i, 7);
    row_nodes(ra, 1, pa, 2, d2i, 7);
    return;
  }
  const int e = (unit - 4) / 3, i = (unit - 4) % 3;
  PE pme, pfe;
  spline_eval(q, 2 + e, t, pme); spline_eval(q, 6 + e, t, pfe);
  const double rr[3] = {pl.p[0] - pme.p[0], pl.p[1] - pme.p[1], pl.p[2] - pme.p[2]};
  {
    RowW ra{&c, c.pos_row[row0 + i], sc[row0 + i], true};
    RowW rl{&c, c.pos_row[row0 + 3 + i], sc[row0 + 3 + i], true};
    double xr[3], xf[3];
    cross_row(rr, i, xr);                           // +(r x df)_i
    cross_row(pfe.p, i, xf);                        // +(f x dp)_i
    const double ml[3] = {i == 0 ? -1.0 : 0.0, i == 1 ? -1.0 : 0.0, i == 2 ? -1.0 : 0.0};
    row_nodes(ra, 6 + e, pfe, 0, xr, 7 & ~(1 << i));
    row_nodes(rl, 6 + e, pfe, 0, ml, 1 << i);
    row_nodes(ra, 2 + e, pme, 0, xf, 7 & ~(1 << i));
    row_durs(ra, 6 + e, t, pfe, xr);                // humanoid_dynamic_constraint.cpp:112-118
    row_durs(rl, 6 + e, t, pfe, ml);
    row_durs(ra, 2 + e, t, pme, xf);
  }
  if (D2 && i == 0) {
    // rows: ang_i - sum_e (F_e x r_e)_i with r_e = c - p_e, and m a_i - sum_e F_e,i.  With La / Ll the multipliers of the
    // angular / linear rows:  d2L = -La . [Q^F x r - (G^F_k x G^p_l + G^F_l x G^p_k) - F x Q^p] - Ll . Q^F
    double La[3], Ll[3];
    for (int k = 0; k < 3; ++k) { La[k] = lam[row0 + k] * sc[row0 + k]; Ll[k] = lam[row0 + 3 + k] * sc[row0 + 3 + k]; }
    DurJac2 dF, dP; dur_jac2(q, 6 + e, t, pfe, dF); dur_jac2(q, 2 + e, t, pme, dP);
    double S3[3];
    for (int cls = 0; cls < 3; ++cls) {
      const double* QF = cls == 0 ? dF.Qee : cls == 1 ? dF.Qec : dF.Qcc;
      const double* QP = cls == 0 ? dP.Qee : cls == 1 ? dP.Qec : dP.Qcc;
      const double* GFx = cls == 2 ? dF.Gc : dF.Ge; const double* GFy = cls == 0 ? dF.Ge : dF.Gc;
      const double* GPx = cls == 2 ? dP.Gc : dP.Ge; const double* GPy = cls == 0 ? dP.Ge : dP.Gc;
      double a1[3], a2[3], a3[3], a4[3];
      cross3(QF, rr, a1); cross3(GFx, GPy, a2); cross3(GFy, GPx, a3); cross3(pfe.p, QP, a4);
      double v = 0;
      for (int k = 0; k < 3; ++k) v += -La[k] * (a1[k] - a2[k] - a3[k] - a4[k]) - Ll[k] * QF[k];
      S3[cls] = v;
    }
    d2_store(q, e, B, dF.cur, S3[0], S3[1], S3[2]);
    {
      // node x duration block.  L = La . [ang - sum_e f_e x r_e] + Ll . [m a - sum_e f_e]:  dL/df_e = -(r_e x La) - Ll,  dL/dp_e = La x f_e,  dL/dc = -sum_e La x f_e
      double AFe[3], AFc[3], BF[3], APe[3], APc[3], BP[3], ACe[3], ACc[3], rxl[3];
      cross3(La, dP.Ge, AFe); cross3(La, dP.Gc, AFc); cross3(rr, La, rxl);
      cross3(La, dF.Ge, APe); cross3(La, dF.Gc, APc); cross3(La, pfe.p, BP);
      for (int k = 0; k < 3; ++k) { AFe[k] = -AFe[k]; AFc[k] = -AFc[k]; BF[k] = -rxl[k] - Ll[k]; ACe[k] = -APe[k]; ACc[k] = -APc[k]; }
      xrec_store(xrec(q, e, XB_DYN_F, B), pfe, dF, AFe, AFc, BF);
      xrec_store(xrec(q, e, XB_DYN_P, B), pme, dP, APe, APc, BP);
      xrec_store(xrec(q, e, XB_DYN_C, B), pl, dP, ACe, ACc, nullptr);
    }
  }
}

// all dynamics samples of the stage.  A function of its own around the inlined unit: as a called function the unit saved and
// restored ~120 callee-saved registers per CALL (five or six calls per thread per evaluation); here that happens once
CHD_NOINLINE CHD_DEV void dyn_rows(LCtx& c, const bool J, const bool D2, GD* cout_, const GD* lam, const GD* sc) {
  SDP S = c.S;
  if (J) { PAR_FOR(u, S->n_dyn * 16) dyn_unit(c, S->dyn_first + u / 16, u % 16, D2, cout_, lam, sc); }
  else { PAR_FOR(u, S->n_dyn) dyn_unit(c, S->dyn_first + u, 0, false, cout_, lam, sc); }
}

CHD_NOINLINE CHD_DEV void eval_rows(LCtx& c, int mode, GD* cout_, const GD* lam) {
  QP q = c.q; SDP S = c.S;
  const GD* sc = VM(c, VM_SC);
  const bool J = mode == EV_FULL;
  const bool D2 = J && S->opt_dur && lam != nullptr && !c.second_model;      // exact duration blocks of the Lagrangian Hessian (first model of an iteration)
  const int slot_height = q->n_tdyn, slot_rom = 2 * q->n_tdyn, slot_heel = 2 * q->n_tdyn + q->n_trom;
  dyn_rows(c, J, D2, cout_, lam, sc);
  PAR_FOR(ti, S->n_tasks) {
    const GI* tk = q->ci + S->o_task + 4 * ti;
    const int type = tk[0], A = tk[1], B = tk[2], row0 = tk[3];
    const double t = q->cd[S->o_task_t + ti];
    switch (type) {
      case T_BASEACC: {      // TOWR SplineAccConstraint: a_j(T_j) - a_{j+1}(0) = 0
        PE e1, e2;
        hermite_eval(q, A, B, q->wd[q->o_poly_dur + q->sp[A].poly_off + B], e1);
        hermite_eval(q, A, B + 1, 0.0, e2);
        for (int k = 0; k < 3; ++k) {
          const int row = row0 + k;
          cout_[row] = sc[row] * (e1.a[k] - e2.a[k]);
          RowW r{&c, c.pos_row[row], sc[row], J};
          double cf[3] = {0, 0, 0};
          cf[k] = 1.0; row_nodes(r, A, e1, 2, cf, 1 << k);
          cf[k] = -1.0; row_nodes(r, A, e2, 2, cf, 1 << k);
        }
      } break;
      case T_TERRAIN: {      // TOWR TerrainConstraint: z - h(x, y)
        const auto& sp = q->sp[2 + A];
        const GD* nv = q->wd + q->o_node + sp.node_off + B * 6;
        const double h = (-q->normal[1] * (nv[1] - q->point[1]) - q->normal[0] * (nv[0] - q->point[0])) / q->normal[2] + q->point[2];   // ground_plane.cpp:18-27
        cout_[row0] = sc[row0] * (nv[2] - h);
        if (J) {
          const GI* vo = q->ci + q->o_varof + sp.node_off + B * 6;
          const int pr = c.pos_row[row0];
          if (vo[2] >= 0) kadd(c, pr, c.pos_var[sp.var_off + vo[2]], sc[row0]);
          if (vo[0] >= 0 && q->hx != 0.0) kadd(c, pr, c.pos_var[sp.var_off + vo[0]], -sc[row0] * q->hx);
          if (vo[1] >= 0 && q->hy != 0.0) kadd(c, pr, c.pos_var[sp.var_off + vo[1]], -sc[row0] * q->hy);
        }
      } break;
      case T_ROM: {          // leg_length_constraint.cpp:36-111: 1/2 |p_ee - (R hip + c)|^2
        const int e = A;
        PE pl, pa, pm;
        spline_eval(q, 0, t, pl); spline_eval(q, 1, t, pa); spline_eval(q, 2 + e, t, pm);
        const GD* hip = q->cd + q->o_hip[(e == 0 || e == 2) ? 0 : 1] + frame_index(q, t) * 3;   // humanoid.h:45-48
        double R[3][3], dR[3][3][3], Rh[3], dvec[3];
        rot_and_derivs(pa.p, R, dR);
        matvec3(R, hip, Rh);
        for (int k = 0; k < 3; ++k) dvec[k] = pm.p[k] - (Rh[k] + pl.p[k]);
        cout_[row0] = sc[row0] * 0.5 * (dvec[0] * dvec[0] + dvec[1] * dvec[1] + dvec[2] * dvec[2]);
        if (J) {
          RowW r{&c, c.pos_row[row0], sc[row0], true};
          double cf[3] = {-dvec[0], -dvec[1], -dvec[2]}, ca[3];
          row_nodes(r, 0, pl, 0, cf, 7);
          for (int k = 0; k < 3; ++k) { double dRh[3]; matvec3(dR[k], hip, dRh); ca[k] = -(dvec[0] * dRh[0] + dvec[1] * dRh[1] + dvec[2] * dRh[2]); }
          row_nodes(r, 1, pa, 0, ca, 7);
          row_nodes(r, 2 + e, pm, 0, dvec, 7);
          row_durs(r, 2 + e, t, pm, dvec);
          if (D2) {      // d2/dT2 of 1/2 |d|^2 = G_k . G_l + d . Q_kl
            DurJac2 dj; dur_jac2(q, 2 + e, t, pm, dj);
            const double ls = lam[row0] * sc[row0];
            d2_store(q, e, slot_rom + B, dj.cur, ls * (dot3(dj.Ge, dj.Ge) + dot3(dvec, dj.Qee)), ls * (dot3(dj.Ge, dj.Gc) + dot3(dvec, dj.Qec)),
                     ls * (dot3(dj.Gc, dj.Gc) + dot3(dvec, dj.Qcc)));
            {      // node x duration block: d = p_ee - R h - c_com
              double Ae[3], Ac[3], Bv[3], nAe[3], nAc[3], tAe[3], tAc[3];
              for (int k = 0; k < 3; ++k) { Ae[k] = ls * dj.Ge[k]; Ac[k] = ls * dj.Gc[k]; Bv[k] = ls * dvec[k]; nAe[k] = -Ae[k]; nAc[k] = -Ac[k]; }
              for (int k = 0; k < 3; ++k) { double dRh[3]; matvec3(dR[k], hip, dRh); tAe[k] = -dot3(dRh, Ae); tAc[k] = -dot3(dRh, Ac); }
              xrec_store(xrec(q, e, XB_ROM_M, B), pm, dj, Ae, Ac, Bv);
              xrec_store(xrec(q, e, XB_ROM_C, B), pl, dj, nAe, nAc, nullptr);
              xrec_store(xrec(q, e, XB_ROM_A, B), pa, dj, tAe, tAc, nullptr);
            }
          }
        }
      } break;
      case T_HEELDIST: {     // ee_dist_constraint.cpp:29-94: 1/2 |p_toe - p_heel|^2
        PE p1, p2;
        spline_eval(q, 2 + A, t, p1); spline_eval(q, 4 + A, t, p2);
        double dvec[3], md[3];
        for (int k = 0; k < 3; ++k) { dvec[k] = p1.p[k] - p2.p[k]; md[k] = -dvec[k]; }
        cout_[row0] = sc[row0] * 0.5 * (dvec[0] * dvec[0] + dvec[1] * dvec[1] + dvec[2] * dvec[2]);
        if (J) {
          RowW r{&c, c.pos_row[row0], sc[row0], true};
          row_nodes(r, 2 + A, p1, 0, dvec, 7); row_nodes(r, 4 + A, p2, 0, md, 7);
          row_durs(r, 2 + A, t, p1, dvec); row_durs(r, 4 + A, t, p2, md);
          if (D2) {      // toe block, heel block and the toe x heel cross block
            DurJac2 da, db; dur_jac2(q, 2 + A, t, p1, da); dur_jac2(q, 4 + A, t, p2, db);
            const double ls = lam[row0] * sc[row0];
            d2_store(q, A, slot_heel + B, da.cur, ls * (dot3(da.Ge, da.Ge) + dot3(dvec, da.Qee)), ls * (dot3(da.Ge, da.Gc) + dot3(dvec, da.Qec)),
                     ls * (dot3(da.Gc, da.Gc) + dot3(dvec, da.Qcc)));
            d2_store(q, A + 2, slot_heel + B, db.cur, ls * (dot3(db.Ge, db.Ge) - dot3(dvec, db.Qee)), ls * (dot3(db.Ge, db.Gc) - dot3(dvec, db.Qec)),
                     ls * (dot3(db.Gc, db.Gc) - dot3(dvec, db.Qcc)));
            GD* x2 = q->wd + q->o_x2tab + ((long long)A * q->n_trom + B) * X2_STRIDE;
            x2[0] = da.cur; x2[1] = db.cur;
            x2[2] = -ls * dot3(da.Ge, db.Ge); x2[3] = -ls * dot3(da.Ge, db.Gc); x2[4] = -ls * dot3(da.Gc, db.Ge); x2[5] = -ls * dot3(da.Gc, db.Gc);
            {      // node x duration block: toe / heel nodes x toe / heel durations
              double Aae[3], Aac[3], Abe[3], Abc[3], nAae[3], nAac[3], nAbe[3], nAbc[3], Ba[3], Bb[3];
              for (int k = 0; k < 3; ++k) {
                Aae[k] = ls * da.Ge[k]; Aac[k] = ls * da.Gc[k]; Abe[k] = ls * db.Ge[k]; Abc[k] = ls * db.Gc[k];
                nAae[k] = -Aae[k]; nAac[k] = -Aac[k]; nAbe[k] = -Abe[k]; nAbc[k] = -Abc[k]; Ba[k] = ls * dvec[k]; Bb[k] = -ls * dvec[k];
              }
              xrec_store(xrec(q, A, XB_HEEL_OWN, B), p1, da, Aae, Aac, Ba);
              xrec_store(xrec(q, A, XB_HEEL_X, B), p2, da, nAae, nAac, nullptr);
              xrec_store(xrec(q, A + 2, XB_HEEL_X, B), p1, db, nAbe, nAbc, nullptr);
              xrec_store(xrec(q, A + 2, XB_HEEL_OWN, B), p2, db, Abe, Abc, Bb);
            }
          }
        }
      } break;
      case T_DYN: break;     // dyn_unit above
      case T_FORCE: {        // TOWR ForceConstraint: normal force range + friction pyramid
        const auto& sp = q->sp[6 + A];
        const GD* nv = q->wd + q->o_node + sp.node_off + B * 6;
        const GI* vo = q->ci + q->o_varof + sp.node_off + B * 6;
        for (int r5 = 0; r5 < 5; ++r5) {
          double dir[3];
          for (int k = 0; k < 3; ++k) {
            const double tk_ = (r5 == 1 || r5 == 2) ? q->bt1[k] : q->bt2[k];
            dir[k] = r5 == 0 ? q->bn[k] : ((r5 & 1) ? tk_ - CHD_MU_FRICTION * q->bn[k] : tk_ + CHD_MU_FRICTION * q->bn[k]);
          }
          const int row = row0 + r5;
          cout_[row] = sc[row] * (nv[0] * dir[0] + nv[1] * dir[1] + nv[2] * dir[2]);
          if (J) for (int k = 0; k < 3; ++k) if (vo[k] >= 0 && dir[k] != 0.0) kadd(c, c.pos_row[row], c.pos_var[sp.var_off + vo[k]], sc[row] * dir[k]);
        }
      } break;
      case T_HEIGHT: {       // height_constraint.cpp:24-58: n . (p - p0) >= 0 with the file normal
        PE pm;
        spline_eval(q, 2 + A, t, pm);
        cout_[row0] = sc[row0] * (q->normal[0] * (pm.p[0] - q->point[0]) + q->normal[1] * (pm.p[1] - q->point[1]) + q->normal[2] * (pm.p[2] - q->point[2]));
        if (J) {
          RowW r{&c, c.pos_row[row0], sc[row0], true};
          const int mask = (q->normal[0] != 0.0 ? 1 : 0) | (q->normal[1] != 0.0 ? 2 : 0) | (q->normal[2] != 0.0 ? 4 : 0);
          const double nrm[3] = {q->normal[0], q->normal[1], q->normal[2]};
          row_nodes(r, 2 + A, pm, 0, nrm, mask);
          row_durs(r, 2 + A, t, pm, nrm);
          if (D2) {
            DurJac2 dj; dur_jac2(q, 2 + A, t, pm, dj);
            const double ls = lam[row0] * sc[row0];
            d2_store(q, A, slot_height + B, dj.cur, ls * dot3(nrm, dj.Qee), ls * dot3(nrm, dj.Qec), ls * dot3(nrm, dj.Qcc));
            { const double Bv[3] = {ls * nrm[0], ls * nrm[1], ls * nrm[2]}; xrec_store(xrec(q, A, XB_HEIGHT, B), pm, dj, nullptr, nullptr, Bv); }
          }
        }
      } break;
      case T_TOTALTIME: {    // total_duration_constraint.cpp:60-82
        const int nv = q->n_phase[A] - 1;
        double sum = 0;
        for (int k = 0; k < nv; ++k) sum += q->wd[q->o_phase_dur + q->phase_off[A] + k];
        cout_[row0] = sc[row0] * sum;
        if (J) for (int k = 0; k < nv; ++k) kadd(c, c.pos_row[row0], c.pos_var[S->dur_off[A] + k], sc[row0]);
      } break;
      case T_DURBOUND: {     // TOWR PhaseDurations::GetBounds
        cout_[row0] = sc[row0] * q->wd[q->o_phase_dur + q->phase_off[A] + B];
        if (J) kadd(c, c.pos_row[row0], c.pos_var[S->dur_off[A] + B], sc[row0]);
      } break;
    }
  }
}

// ------------------------------------------------------------------------------------------
// Cost terms: sum 1/2 w r^2 with gradient and Gauss-Newton Hessian
// ------------------------------------------------------------------------------------------
CHD_DEV const GD* scache(QP q, int s, int i) { return q->wd + q->o_scache + ((long long)s * (q->F + 2) + i) * SC_STRIDE; }

CHD_NOINLINE CHD_DEV void fill_sample_cache(LCtx& c, const bool with_dur) {       // with_dur: also the duration derivatives (full evaluations only)
  QP q = c.q;
  const int F1 = q->F + 1;
  PAR_FOR(idx, 6 * F1) {
    const int s = idx / F1, i = idx % F1;
    const double t = q->cd[q->o_tcost + i];
    PE e;
    spline_eval(q, s, t, e);
    GD* sc_ = q->wd + q->o_scache + ((long long)s * (q->F + 2) + i) * SC_STRIDE;
    for (int k = 0; k < 4; ++k) { sc_[SC_WP + k] = e.w[0][k]; sc_[SC_WV + k] = e.w[1][k]; }
    for (int k = 0; k < 3; ++k) { sc_[SC_P + k] = e.p[k]; sc_[SC_V + k] = e.v[k]; sc_[SC_DXDT + k] = 0.0; }
    sc_[SC_POLY] = e.poly; sc_[SC_PHASE] = 0; sc_[SC_LAST] = 0;
    for (int j = 0; j < 4; ++j) { sc_[SC_OME + j] = 0.0; sc_[SC_OMC + j] = 0.0; }
    if (with_dur && c.S->opt_dur && s >= 2) {
      DurJac dj;
      dur_jac(q, s, t, e, dj);
      // store own / early in a form usable per k: own[] (k == cur, not last) and early[] (k < cur)
      for (int k = 0; k < 3; ++k) sc_[SC_DXDT + k] = dj.last ? -(dj.early[k] + e.v[k]) : dj.own[k];   // = dx_dT
      sc_[SC_PHASE] = dj.cur; sc_[SC_LAST] = dj.last;
      DurJac2 d2;
      dur_jac2(q, s, t, e, d2);
      for (int k = 0; k < 3; ++k) { sc_[SC_QEE + k] = d2.Qee[k]; sc_[SC_QEC + k] = d2.Qec[k]; sc_[SC_QCC + k] = d2.Qcc[k]; }
      for (int j = 0; j < 4; ++j) { sc_[SC_OME + j] = d2.ome[j]; sc_[SC_OMC + j] = d2.omc[j]; }
    }
  }
  CHD_SYNC();
}
// d p_i[dim] / d T_k from the cache
template <class SP>
CHD_DEV double cache_djac(SP sc_, int dim, int k) {
  const int cur = (int)sc_[SC_PHASE], last = (int)sc_[SC_LAST];
  if (k > cur) return 0.0;
  if (k == cur) return last ? 0.0 : sc_[SC_DXDT + dim];
  return -sc_[SC_V + dim] - (last ? sc_[SC_DXDT + dim] : 0.0);
}

// number of smoothing residuals of spline s: loop `for (t = 0; t < T_s - dt; t += dt)` (vel_smooth_cost.cpp:41)
CHD_DEV int n_smooth(QP q, int s) {
  const double lim = q->wd[q->o_ttot + s] - q->dt - 1e-9;   // the last sample sits exactly on the limit: tolerance, not rounding, decides
  int n = q->F;
  while (n > 0 && !(q->cd[q->o_tcost + n - 1] < lim)) --n;
  return n;
}

CHD_NOINLINE CHD_DEV double eval_cost_value(LCtx& c) {
  QP q = c.q; SDP S = c.S;
  const int F = q->F;
  double part = 0.0;
  PAR_FOR(idx, 6 * F) {
    const int s = idx / F, i = idx % F;
    const double wdat = S->w_data[s < 2 ? s : 2], wvel = S->w_vel[s < 2 ? s : 2], wacc = S->w_acc[s < 2 ? s : 2];
    const GD* a = scache(q, s, i);
    const GD* dat = q->cd + q->o_data[s] + i * 3;
    double acc = 0;
    for (int k = 0; k < 3; ++k) { const double r = dat[k] - a[SC_P + k]; acc += 0.5 * wdat * r * r; }
    if (i < n_smooth(q, s)) {
      const GD* b = scache(q, s, i + 1);
      if (wvel >= 0) for (int k = 0; k < 3; ++k) { const double r = b[SC_P + k] - a[SC_P + k]; acc += 0.5 * wvel * r * r; }
      if (wacc >= 0) for (int k = 0; k < 3; ++k) { const double r = b[SC_V + k] - a[SC_V + k]; acc += 0.5 * wacc * r * r; }
    }
    part += acc;
  }
  if (S->opt_dur && S->w_dur >= 0) {
    PAR_FOR(idx, q->tot_phases) {
      int e = 0; while (e + 1 < N_EE && idx >= q->phase_off[e + 1]) ++e;
      const int k = idx - q->phase_off[e];
      if (k + 1 < q->n_phase[e]) { const double r = q->cd[q->o_phase_dur0 + idx] - q->wd[q->o_phase_dur + idx]; part += 0.5 * S->w_dur * r * r; }
    }
  }
  return block_sum(c, part);
}

// one residual's node support: up to 8 (node, deriv, weight) entries
struct Supp { int n; int node[8]; int dq[8]; double g[8]; };
CHD_DEV void supp_add_sample(Supp& sp, const double* sc_, int which, double sign) {
  const int poly = (int)sc_[SC_POLY];
  const double* wv = sc_ + (which == 0 ? SC_WP : SC_WV);
  for (int side = 0; side < 2; ++side)
    for (int dq = 0; dq < 2; ++dq) { sp.node[sp.n] = poly + side; sp.dq[sp.n] = dq; sp.g[sp.n] = sign * wv[side * 2 + dq]; ++sp.n; }
}

// Exact curvature of the heel-distance rows (ee_dist_constraint.cpp:29-94): c = 1/2 |p_toe(t) - p_heel(t)|^2 is quadratic in the
// node values, and its multipliers reach 10^2 .. 10^3, so lam * grad^2 c is a large part of the Lagrangian Hessian that the
// Gauss-Newton model lacks (without it the optimality error hovers just above tol for 100+ iterations on the slow
// sequences).  grad^2 c = sum_dim g g^T with g = Hermite weights of the toe polynomial (+) and of the heel polynomial (-) at
// the sample.  This cache holds, per (end-effector, range-of-motion sample): the four position weights, the polynomial and
// the row's multiplier times its scaling; the entries are then gathered owner-computes like the cost terms.  (First model of an iteration only:
// the second model is Gauss-Newton, see solve_stage.)
CHD_DEV const GD* rcache(QP q, int ee, int k) { return q->wd + q->o_rcache + ((long long)ee * q->n_trom + k) * RC_STRIDE; }
CHD_DEV void fill_rom_cache(LCtx& c, const GD* lam) {
  QP q = c.q; SDP S = c.S;
  const GD* sc = VM(c, VM_SC);
  PAR_FOR(idx, 4 * q->n_trom) {
    const int ee = idx / q->n_trom, k = idx % q->n_trom;
    PE e;
    spline_eval(q, 2 + ee, q->cd[q->o_trom + k], e);
    GD* r = q->wd + q->o_rcache + (long long)idx * RC_STRIDE;
    for (int j = 0; j < 4; ++j) r[SC_WP + j] = e.w[0][j];
    const int row = S->heel_row0 + (ee % 2) * q->n_trom + k;          // pairs (0, 2) and (1, 3): nlp_formulation.cpp:249-257
    r[RC_MU] = lam[row] * sc[row];
    r[SC_POLY] = e.poly;
  }
  CHD_SYNC();
}

CHD_NOINLINE CHD_DEV void eval_cost_grad_hess(LCtx& c, GD* g, const GD* lam) {
  QP q = c.q; SDP S = c.S;
  const int F = q->F;
  GI* first = q->wi + q->o_first;
  const int fstride = q->max_polys + 2;
  // first data sample of every polynomial
  PAR_FOR(idx, 6 * fstride) first[idx] = F;
  PAR_FOR(j, c.n) g[j] = 0.0;
  CHD_SYNC();
  int gap = 1;              // largest polynomial-index step between consecutive samples
  PAR_FOR(idx, 6 * F) {
    const int s = idx / F, i = idx % F;
    const int pi = (int)scache(q, s, i)[SC_POLY];
    const int pp = i > 0 ? (int)scache(q, s, i - 1)[SC_POLY] : -1;
    for (int p = pp + 1; p <= pi; ++p) first[s * fstride + p] = i;
    if (i > 0 && pi - pp > gap) gap = pi - pp;
  }
  gap = (int)block_max(c, (double)gap);       // (ends with a barrier)
  const bool HC = lam != nullptr && S->heel_row0 >= 0 && !c.second_model;      // exact curvature of the heel-distance rows: first model of an iteration only
  if (HC) fill_rom_cache(c, lam);
  long long tg_ = CHD_CLOCK();
  // ---- node variables.  Every cost residual is linear in the node values for fixed durations,
  //   r = sum_v G(v) x_v + const, with G the Hermite weights of the sample(s) the residual touches, so the
  //   Gauss-Newton block is the Gram matrix of the G's.  One thread per entry (row coefficient, column coefficient
  //   at most `reach` nodes back) sums its products in a register; a stance pair (two nodes, one variable) is one
  //   coefficient whose weight is the sum over both nodes.  The weights do not depend on the dimension.
  int tot_nodes = 0;
  for (int s = 0; s < 6; ++s) tot_nodes += q->sp[s].n_nodes;
  const int reach = gap + 2, ncol = (reach + 1) * 2;
  auto second_of_pair = [&](const auto& sp, const GI* pinfo, int n) { return sp.phase_based && n > 0 && pinfo[(n - 1) * 4 + 3] != 0; };
  auto group_end = [&](const auto& sp, const GI* pinfo, int n) { return (sp.phase_based && n < sp.n_polys && pinfo[n * 4 + 3]) ? n + 1 : n; };
  // weight of the coefficient (nodes n..nh, derivative dq) in the position (which = 0) / velocity (1) of a sample
  auto wgt = [](auto a, int which, int n, int nh, int dq) {
    const int p = (int)a[SC_POLY];
    auto W = a + (which ? SC_WV : SC_WP);
    double v = 0.0;
    if (p >= n && p <= nh) v += W[dq];
    if (p + 1 >= n && p + 1 <= nh) v += W[2 + dq];
    return v;
  };
  auto node_terms = [&](auto sample) {
    EVT_BEGIN();
    PAR_FOR(idx, tot_nodes * 2 * ncol) {
      const int col = idx % ncol, row = idx / ncol;
      const int dq1 = row % 2, back = col / 2, dq2 = col % 2;
      int n1 = row / 2, s = 0;
      while (n1 >= q->sp[s].n_nodes) { n1 -= q->sp[s].n_nodes; ++s; }
      const auto& sp = q->sp[s];
      const GI* pinfo = q->ci + q->o_pinfo + sp.poly_off * 4;
      const GI* vo = q->ci + q->o_varof + sp.node_off;
      const int n2 = n1 - back;
      if (n2 < 0 || (back == 0 && dq2 > dq1)) continue;
      if (second_of_pair(sp, pinfo, n1) || second_of_pair(sp, pinfo, n2)) continue;       // folded into the pair's first node
      const int h1 = group_end(sp, pinfo, n1), h2 = group_end(sp, pinfo, n2);
      bool any = false;
      for (int dim = 0; dim < 3; ++dim) any = any || (vo[n1 * 6 + dq1 * 3 + dim] >= 0 && vo[n2 * 6 + dq2 * 3 + dim] >= 0);
      if (!any) continue;
      const int pa = n1 - 1 < 0 ? 0 : n1 - 1, pb = h1 > sp.n_polys - 1 ? sp.n_polys - 1 : h1;
      const int i_lo = first[s * fstride + pa], i_hi = first[s * fstride + pb + 1] - 1;
      const int nsm = n_smooth(q, s);
      const int ti = s < 2 ? s : 2;
      const double wdat = S->w_data[ti], wvel = S->w_vel[ti], wacc = S->w_acc[ti];
      double acc = 0.0;
      for (int i = i_lo; i <= i_hi && i < F; ++i) {
        auto a = sample(s, i);
        acc += wdat * wgt(a, 0, n1, h1, dq1) * wgt(a, 0, n2, h2, dq2);
      }
      if (wvel >= 0 || wacc >= 0)
        for (int i = i_lo - 1 < 0 ? 0 : i_lo - 1; i <= i_hi && i < nsm; ++i) {
          auto a = sample(s, i); auto b = sample(s, i + 1);
          if (wvel >= 0) acc += wvel * (wgt(b, 0, n1, h1, dq1) - wgt(a, 0, n1, h1, dq1)) * (wgt(b, 0, n2, h2, dq2) - wgt(a, 0, n2, h2, dq2));
          if (wacc >= 0) acc += wacc * (wgt(b, 1, n1, h1, dq1) - wgt(a, 1, n1, h1, dq1)) * (wgt(b, 1, n2, h2, dq2) - wgt(a, 1, n2, h2, dq2));
        }
      double acch = 0.0;                   // + sum over the range-of-motion samples in these polynomials of lam sc w w (heel-distance rows)
      if (HC && s >= 2) {
        int k_lo, k_hi;
        sample_range(q, s, pa, pb, q->cd + q->o_trom, q->n_trom, k_lo, k_hi);
        for (int k = k_lo; k <= k_hi; ++k) {
          const GD* a = rcache(q, s - 2, k);
          const int p = (int)a[SC_POLY];
          if (p < pa || p > pb) continue;
          acch += a[RC_MU] * wgt(a, 0, n1, h1, dq1) * wgt(a, 0, n2, h2, dq2);
        }
      }
      if (acc == 0.0 && acch == 0.0) continue;
      {
        int pp[3], qv[3]; double vv[3];
#pragma unroll
        for (int dim = 0; dim < 3; ++dim) {
          const int v1 = vo[n1 * 6 + dq1 * 3 + dim], v2 = vo[n2 * 6 + dq2 * 3 + dim];
          const bool on = v1 >= 0 && v2 >= 0;
          pp[dim] = on ? c.pos_var[sp.var_off + v1] : -1; qv[dim] = on ? c.pos_var[sp.var_off + v2] : 0; vv[dim] = c.sf * acc + acch;
        }
        kadd_batch<3>(c, pp, qv, vv);      // the three dimensions are three different entries
        if (acch != 0.0) {                 // the second model's entry: the Gauss-Newton sum alone (model_switch)
          const double gn = acc != 0.0 ? side_gn(c.sf * acc) : -0.0;
#pragma unroll
          for (int dim = 0; dim < 3; ++dim) if (pp[dim] >= 0) { if (vv[dim] != 0.0) side_put(c, pp[dim], qv[dim], gn); else c.side_ok = 0; }
        }
      }
    }
    EVT(c, 16);
    // gradient: one thread per (coefficient, dimension)
    PAR_FOR(idx, tot_nodes * 2 * 3) {
      const int dim = idx % 3, row = idx / 3, dq1 = row % 2;
      int n1 = row / 2, s = 0;
      while (n1 >= q->sp[s].n_nodes) { n1 -= q->sp[s].n_nodes; ++s; }
      const auto& sp = q->sp[s];
      const GI* pinfo = q->ci + q->o_pinfo + sp.poly_off * 4;
      const GI* vo = q->ci + q->o_varof + sp.node_off;
      const int v1 = vo[n1 * 6 + dq1 * 3 + dim];
      if (v1 < 0 || second_of_pair(sp, pinfo, n1)) continue;
      const int h1 = group_end(sp, pinfo, n1);
      const int pa = n1 - 1 < 0 ? 0 : n1 - 1, pb = h1 > sp.n_polys - 1 ? sp.n_polys - 1 : h1;
      const int i_lo = first[s * fstride + pa], i_hi = first[s * fstride + pb + 1] - 1;
      const int nsm = n_smooth(q, s);
      const int ti = s < 2 ? s : 2;
      const double wdat = S->w_data[ti], wvel = S->w_vel[ti], wacc = S->w_acc[ti];
      const GD* dat = q->cd + q->o_data[s];
      double acc = 0.0;
      for (int i = i_lo; i <= i_hi && i < F; ++i) {
        auto a = sample(s, i);
        acc -= wdat * (dat[i * 3 + dim] - a[SC_P + dim]) * wgt(a, 0, n1, h1, dq1);
      }
      if (wvel >= 0 || wacc >= 0)
        for (int i = i_lo - 1 < 0 ? 0 : i_lo - 1; i <= i_hi && i < nsm; ++i) {
          auto a = sample(s, i); auto b = sample(s, i + 1);
          if (wvel >= 0) acc += wvel * (b[SC_P + dim] - a[SC_P + dim]) * (wgt(b, 0, n1, h1, dq1) - wgt(a, 0, n1, h1, dq1));
          if (wacc >= 0) acc += wacc * (b[SC_V + dim] - a[SC_V + dim]) * (wgt(b, 1, n1, h1, dq1) - wgt(a, 1, n1, h1, dq1));
        }
      g[sp.var_off + v1] += c.sf * acc;
    }
    EVT(c, 17);
  };
  // the per-sample fields are read ~10^5 times by the entry tasks: keep the first SC_LDS fields of the cache in LDS
  const int sstride = SC_LDS;
  const bool in_lds = 6 * (F + 2) * sstride <= c.lds_cap - LDS_RED;
  LdsD* ws = c.lds + LDS_RED;
  auto lds_sample = [&](int s, int i) { return (const LdsD*)(ws + ((long long)s * (F + 2) + i) * sstride); };
  auto hbm_sample = [&](int s, int i) { return scache(q, s, i); };
  if (in_lds) {
    PAR_FOR(idx, 6 * (F + 2) * sstride) { const int si = idx / sstride, fld = idx % sstride; ws[idx] = (q->wd + q->o_scache + (long long)si * SC_STRIDE)[fld]; }
    CHD_SYNC();
    node_terms(lds_sample);
  } else node_terms(hbm_sample);
  EVT_BEGIN();
  if (HC) {
    // toe x heel blocks of the heel-distance curvature: one thread per (pair, toe coefficient, heel coefficient); the entry is
    // - sum over the samples of lam sc w_toe w_heel, the same for the three dimensions
    const int nA0 = q->sp[2].n_nodes, nB0 = q->sp[4].n_nodes, nA1 = q->sp[3].n_nodes, nB1 = q->sp[5].n_nodes;
    const int cnt0 = 4 * nA0 * nB0, cnt1 = 4 * nA1 * nB1;
    PAR_FOR(idx0, cnt0 + cnt1) {
      const int pr = idx0 < cnt0 ? 0 : 1;
      const int idx = pr ? idx0 - cnt0 : idx0;
      const int nB_ = pr ? nB1 : nB0;
      const int dq1 = idx & 1, dq2 = (idx >> 1) & 1, n1 = (idx >> 2) / nB_, n2 = (idx >> 2) % nB_;
      const auto& sa = q->sp[2 + pr]; const auto& sb = q->sp[4 + pr];
      const GI* pia = q->ci + q->o_pinfo + sa.poly_off * 4; const GI* pib = q->ci + q->o_pinfo + sb.poly_off * 4;
      if (second_of_pair(sa, pia, n1) || second_of_pair(sb, pib, n2)) continue;
      const GI* voa = q->ci + q->o_varof + sa.node_off; const GI* vob = q->ci + q->o_varof + sb.node_off;
      bool any = false;
      for (int dim = 0; dim < 3; ++dim) any = any || (voa[n1 * 6 + dq1 * 3 + dim] >= 0 && vob[n2 * 6 + dq2 * 3 + dim] >= 0);
      if (!any) continue;
      const int h1 = group_end(sa, pia, n1), h2 = group_end(sb, pib, n2);
      double acc = 0.0;
      int k_lo, k_hi;
      sample_range(q, 2 + pr, n1 - 1 < 0 ? 0 : n1 - 1, h1 > sa.n_polys - 1 ? sa.n_polys - 1 : h1, q->cd + q->o_trom, q->n_trom, k_lo, k_hi);
      for (int k = k_lo; k <= k_hi; ++k) {
        const GD* a = rcache(q, pr, k);
        const int p = (int)a[SC_POLY];
        if (p + 1 < n1 || p > h1) continue;                 // the toe polynomial of this sample does not touch the coefficient
        const GD* b = rcache(q, 2 + pr, k);
        acc -= a[RC_MU] * wgt(a, 0, n1, h1, dq1) * wgt(b, 0, n2, h2, dq2);
      }
      if (acc == 0.0) continue;
      int pp[3], qv[3]; double vv[3];
#pragma unroll
      for (int dim = 0; dim < 3; ++dim) {
        const int v1 = voa[n1 * 6 + dq1 * 3 + dim], v2 = vob[n2 * 6 + dq2 * 3 + dim];
        const bool on = v1 >= 0 && v2 >= 0;
        pp[dim] = on ? c.pos_var[sa.var_off + v1] : -1; qv[dim] = on ? c.pos_var[sb.var_off + v2] : 0; vv[dim] = acc;
      }
      kadd_batch<3>(c, pp, qv, vv);
#pragma unroll
      for (int dim = 0; dim < 3; ++dim) if (pp[dim] >= 0) side_put(c, pp[dim], qv[dim], -0.0);      // (the second model has no toe x heel block)
    }
  }
  EVT(c, 18);
  CHD_SYNC();
  TACC(c, 13, CHD_CLOCK() - tg_); tg_ = CHD_CLOCK();
  // ---- duration variables (stage 3 only)
  if (S->opt_dur && lam && !c.second_model) {
    // residual-weighted curvature of the cost terms, one table slot per data sample:
    //   data  1/2 w |data_i - p_i|^2            -> -w r_i . Q(i)
    //   smooth 1/2 w |p_{i+1} - p_i|^2          -> +w r_i . Q(i+1)  and  -w r_i . Q(i)
    const int slot_cost = 2 * q->n_tdyn + 2 * q->n_trom;
    PAR_FOR(idx, 4 * F) {
      const int e = idx / F, i = idx % F, s = 2 + e;
      const GD* a = scache(q, s, i);
      const GD* dat = q->cd + q->o_data[s] + i * 3;
      const int nsm = n_smooth(q, s);
      double cf[3];
      for (int k = 0; k < 3; ++k) {
        double v = -S->w_data[2] * (dat[k] - a[SC_P + k]);
        if (S->w_vel[2] >= 0) {
          if (i < nsm) v -= S->w_vel[2] * (scache(q, s, i + 1)[SC_P + k] - a[SC_P + k]);
          if (i > 0 && i - 1 < nsm) v += S->w_vel[2] * (a[SC_P + k] - scache(q, s, i - 1)[SC_P + k]);
        }
        cf[k] = c.sf * v;
      }
      d2_store(q, e, slot_cost + i, (int)a[SC_PHASE], dot3(cf, a + SC_QEE), dot3(cf, a + SC_QEC), dot3(cf, a + SC_QCC));
      { GD* aw = q->wd + q->o_scache + ((long long)s * (q->F + 2) + i) * SC_STRIDE; for (int k = 0; k < 3; ++k) aw[SC_CF + k] = cf[k]; }      // (for the node x duration entries below)
    }
    CHD_SYNC();
    // toe x heel cross blocks of the heel-distance rows: one thread per (pair, k, l)
    if (S->families & FAM_HEELDIST) {
      const int na0 = q->n_phase[0] - 1, nb0 = q->n_phase[2] - 1, na1 = q->n_phase[1] - 1, nb1 = q->n_phase[3] - 1;
      PAR_FOR(idx0, na0 * nb0 + na1 * nb1) {
        const int pr_ = idx0 < na0 * nb0 ? 0 : 1;
        const int idx = pr_ ? idx0 - na0 * nb0 : idx0;
        const int nb_ = pr_ ? nb1 : nb0;
        const int k = idx / nb_, l = idx % nb_;
        double acc = 0;
        for (int smp = 0; smp < q->n_trom; ++smp) {
          const GD* x2 = q->wd + q->o_x2tab + ((long long)pr_ * q->n_trom + smp) * X2_STRIDE;
          const int ca = (int)x2[0], cb = (int)x2[1];
          if (k > ca || l > cb) continue;
          acc += x2[2 + (k == ca ? 2 : 0) + (l == cb ? 1 : 0)];
        }
        if (acc != 0.0) { kadd(c, c.pos_var[S->dur_off[pr_] + k], c.pos_var[S->dur_off[pr_ + 2] + l], acc); side_put(c, c.pos_var[S->dur_off[pr_] + k], c.pos_var[S->dur_off[pr_ + 2] + l], -0.0); }
      }
    }
  }
  CHD_SYNC();
  TACC(c, 14, CHD_CLOCK() - tg_); tg_ = CHD_CLOCK();
  if (S->opt_dur) {
    auto dur_terms = [&](auto sample) {
    EVT_BEGIN();
    int tot = 0;
    for (int e = 0; e < 4; ++e) tot += (q->n_phase[e] - 1) * (q->n_phase[e] - 1);
    GROUP_FOR(idx0, tot) {     // (T_k, T_k2) entries: a lane group per entry, the samples and the table slots dealt out over its lanes (fixed-tree sums)
      int idx = idx0, e = 0;
      while (idx >= (q->n_phase[e] - 1) * (q->n_phase[e] - 1)) { idx -= (q->n_phase[e] - 1) * (q->n_phase[e] - 1); ++e; }
      const int s = 2 + e;
      const int ntar = q->n_phase[e] - 1;
      const int k = idx / ntar, k2 = idx % ntar;
      if (k2 > k) continue;            // (uniform over the group)
      const int Pk = c.pos_var[S->dur_off[e] + k];
      const double wdat = S->w_data[2], wvel = S->w_vel[2];
      const int nsm = n_smooth(q, s);
      const GD* dat = q->cd + q->o_data[s];
      // ---- (T_k, T_k2), k2 <= k : all residuals of this end-effector
      double hacc = 0, gacc = 0;
      for (int i = lane_; i < F; i += CHD_GL) {
        auto a = sample(s, i);
        for (int dm = 0; dm < 3; ++dm) {
          const double gk = -cache_djac(a, dm, k), gk2 = -cache_djac(a, dm, k2);
          hacc += wdat * gk * gk2;
          if (k2 == k) gacc += wdat * (dat[i * 3 + dm] - a[SC_P + dm]) * gk;
        }
        if (i < nsm && wvel >= 0) {
          auto b = sample(s, i + 1);
          for (int dm = 0; dm < 3; ++dm) {
            const double gk = cache_djac(b, dm, k) - cache_djac(a, dm, k), gk2 = cache_djac(b, dm, k2) - cache_djac(a, dm, k2);
            hacc += wvel * gk * gk2;
            if (k2 == k) gacc += wvel * (b[SC_P + dm] - a[SC_P + dm]) * gk;
          }
        }
      }
      double h2 = 0;
      if (lam && !c.second_model) {     // exact second-order terms collected by the row tasks and the cost pass above (first model of an iteration)
        const GD* tb = q->wd + q->o_d2tab + (long long)e * q->d2_slots * D2_STRIDE;
        for (int sl = lane_; sl < q->d2_slots; sl += CHD_GL) h2 += d2_select(tb + sl * D2_STRIDE, k, k2);
      }
      hacc = group_sum(hacc); gacc = group_sum(gacc); h2 = group_sum(h2);
      if (lane_ != 0) continue;
      double hgn = hacc;                  // the second model's sum: without the exact terms
      hacc += h2 / c.sf;
      if (k2 == k && S->w_dur >= 0) {     // DurationCost (duration_cost.cpp:25-50): 1/2 w (T0 - T)^2
        hacc += S->w_dur; hgn += S->w_dur;
        gacc += S->w_dur * (q->wd[q->o_phase_dur + q->phase_off[e] + k] - q->cd[q->o_phase_dur0 + q->phase_off[e] + k]);
      }
      kadd(c, Pk, c.pos_var[S->dur_off[e] + k2], c.sf * hacc);
      if (h2 != 0.0) { if (c.sf * hacc != 0.0) side_put(c, Pk, c.pos_var[S->dur_off[e] + k2], side_gn(c.sf * hgn)); else c.side_ok = 0; }
      if (k2 == k) g[S->dur_off[e] + k] = c.sf * gacc;
    }
    // ---- (T_k, node variable) entries: one thread per node variable.  Gauss-Newton part (cost terms of the ee-motion splines): the thread gathers the
    // residuals that touch the variable's node(s) once and accumulates the entries of up to 8 durations at a time in registers.  Exact part (first model
    // of an iteration, multipliers given): + the residual curvature of the cost terms, sum_i df/dp_i . d2p_i/dx dT, and the records the row tasks left in
    // the node x duration table (chd_device.hpp, XR_* / XB_*) -- which also couple the durations with force, centre-of-mass and base-angle nodes.
    EVT(c, 19);
    const bool DX = lam != nullptr && !c.second_model;
    const GI* varspl = q->ci + q->o_varspl; const GI* varnode = q->ci + q->o_varnode;
    PAR_FOR(var, q->n_nodesvars) {
      const int s = varspl[var];
      if (!DX && (s < 2 || s > 5)) continue;
      const auto& sp = q->sp[s];
      const double wdat = S->w_data[2], wvel = S->w_vel[2];
      const int ent = varnode[var];
      const int nd = ent / 6, dq0 = (ent % 6) / 3, dm = ent % 3;
      const GI* pinfo = q->ci + q->o_pinfo + sp.poly_off * 4;
      const int nd_hi = (dq0 == 0 && nd < sp.n_polys && pinfo[nd * 4 + 3]) ? nd + 1 : nd;
      const int pa = nd - 1 < 0 ? 0 : nd - 1, pb = nd_hi > sp.n_polys - 1 ? sp.n_polys - 1 : nd_hi;
      const int Pv = c.pos_var[var];
      // weight of this variable in a quantity sum_j W[j] x_j over the coefficients of polynomial `poly` (which touches node poly: side 0, and poly + 1: side 1)
      auto wsel = [&](const int poly, auto W) -> double {
        double g_ = 0;
        if (poly >= nd && poly <= nd_hi) g_ += W[dq0];
        if (poly + 1 >= nd && poly + 1 <= nd_hi) g_ += W[2 + dq0];
        return g_;
      };
      for (int te = 0; te < N_EE; ++te) {      // end-effector whose durations the entries differentiate by
        int blks[4], nblk = 0;
        bool own_cost = false;
        if (s >= 2 && s <= 5) {
          if (te == s - 2) { own_cost = true; if (DX) { blks[0] = XB_HEIGHT; blks[1] = XB_ROM_M; blks[2] = XB_HEEL_OWN; blks[3] = XB_DYN_P; nblk = 4; } }
          else if (DX && te == (s - 2 + 2) % 4) { blks[0] = XB_HEEL_X; nblk = 1; }
          else continue;
        } else if (s >= 6) {
          if (te != s - 6) continue;
          blks[0] = XB_DYN_F; nblk = 1;
        } else if (s == 0) { blks[0] = XB_ROM_C; blks[1] = XB_DYN_C; nblk = 2; }
        else { blks[0] = XB_ROM_A; nblk = 1; }
        const int nv = q->n_phase[te] - 1;
        const int nsm = own_cost ? n_smooth(q, s) : 0;
        const int i_lo = own_cost ? first[s * fstride + pa] : 0;
        const int i_hi = own_cost ? first[s * fstride + pb + 1] - 1 : -1;
        for (int k0 = 0; k0 < nv; k0 += 8) {
          double hk[8], hx[8];
#pragma unroll
          for (int kk = 0; kk < 8; ++kk) { hk[kk] = 0.0; hx[kk] = 0.0; }
          if (own_cost && wdat >= 0) {
            const int r_hi = i_hi > F - 1 ? F - 1 : i_hi;
            for (int i = i_lo; i <= r_hi; ++i) {
              auto a = sample(s, i);
              const double gt = -wdat * wsel((int)a[SC_POLY], a + SC_WP);             // r = data - p
#pragma unroll
              for (int kk = 0; kk < 8; ++kk) if (k0 + kk < nv) hk[kk] += gt * (-cache_djac(a, dm, k0 + kk));
            }
          }
          if (own_cost && wvel >= 0) {
            const int r_lo = i_lo - 1 < 0 ? 0 : i_lo - 1, r_hi = i_hi > nsm - 1 ? nsm - 1 : i_hi;
            for (int i = r_lo; i <= r_hi; ++i) {
              auto a = sample(s, i); auto b = sample(s, i + 1);
              const double gt = wvel * (wsel((int)b[SC_POLY], b + SC_WP) - wsel((int)a[SC_POLY], a + SC_WP));      // r = p_{i+1} - p_i
#pragma unroll
              for (int kk = 0; kk < 8; ++kk) if (k0 + kk < nv) hk[kk] += gt * (cache_djac(b, dm, k0 + kk) - cache_djac(a, dm, k0 + kk));
            }
          }
          if (own_cost && DX) {
            const int r_hi = i_hi > F - 1 ? F - 1 : i_hi;
            for (int i = i_lo; i <= r_hi; ++i) {
              auto a = sample(s, i);
              const double cfd = scache(q, s, i)[SC_CF + dm];
              const int poly = (int)a[SC_POLY], cur = (int)a[SC_PHASE], last = (int)a[SC_LAST];
              const double oe = cfd * wsel(poly, a + SC_OME), oc = last ? 0.0 : cfd * wsel(poly, a + SC_OMC);
#pragma unroll
              for (int kk = 0; kk < 8; ++kk) { const int k = k0 + kk; if (k < nv) hx[kk] += k < cur ? oe : (k == cur ? oc : 0.0); }
            }
          }
          for (int bi = 0; bi < nblk; ++bi) {
            const int len = xblock_len(q, blks[bi]);
            int k_lo, k_hi;
            sample_range(q, s, pa, pb, q->cd + xblock_times(q, blks[bi]), len, k_lo, k_hi);
            const GD* r = xrec(q, te, blks[bi], k_lo);
            for (int smp = k_lo; smp <= k_hi; ++smp, r += XR_STRIDE) {
              const int poly = (int)r[XR_POLY];
              if (poly < pa) continue;
              if (poly > pb) break;            // (the samples of a block are ordered in time; an unused slot reads as polynomial 0 with zero coefficients)
              const int cur = (int)r[XR_CUR], last = (int)r[XR_LAST];
              const double g_ = wsel(poly, r + XR_W), bd = r[XR_B + dm];
              const double ge = g_ * r[XR_AE + dm] + wsel(poly, r + XR_OME) * bd;
              const double gc = last ? 0.0 : g_ * r[XR_AC + dm] + wsel(poly, r + XR_OMC) * bd;
#pragma unroll
              for (int kk = 0; kk < 8; ++kk) { const int k = k0 + kk; if (k < nv) hx[kk] += k < cur ? ge : (k == cur ? gc : 0.0); }
            }
          }
          {
            int pp[8], qv[8]; double vv[8];
#pragma unroll
            for (int kk = 0; kk < 8; ++kk) {
              vv[kk] = c.sf * hk[kk] + hx[kk];
              const bool on = k0 + kk < nv && vv[kk] != 0.0;
              pp[kk] = on ? c.pos_var[S->dur_off[te] + k0 + kk] : -1; qv[kk] = Pv;
            }
            kadd_batch<8>(c, pp, qv, vv);
            if (DX) {
#pragma unroll
              for (int kk = 0; kk < 8; ++kk)
                if (k0 + kk < nv && hx[kk] != 0.0) {      // the exact part changed this entry: the second model holds the cost's Gauss-Newton term alone (own end-effector) or nothing
                  if (vv[kk] != 0.0) side_put(c, pp[kk], Pv, own_cost ? side_gn(c.sf * hk[kk]) : -0.0); else c.side_ok = 0;
                }
            }
          }
        }
      }
    }
    EVT(c, 20);
    };
    if (in_lds) dur_terms(lds_sample); else dur_terms(hbm_sample);
  }
  CHD_SYNC();
  TACC(c, 15, CHD_CLOCK() - tg_);
}

// Full evaluation at x.  Returns the (scaled) objective; fills c_out (scaled rows), and in
// EV_FULL mode the scaled gradient g and the unfactored KKT matrix K0 = [sf H, (sc J)^T; sc J, 0].
CHD_DEV double eval_nlp(LCtx& c, const GD* x, int mode, GD* c_out, GD* g, const GD* lam = nullptr) {
  TIC();
  state_from_x(c, x);
  if (mode == EV_FULL) {
    kzero(c);
    if (CHD_TID == 0) { c.side_n = 0; c.side_ok = (lam != nullptr && !c.second_model) ? 1 : 0; }      // (ordered before the entry tasks by the barriers of the cache fill below)
    if (c.S->opt_dur && lam) {
      PAR_FOR(i, 4 * c.q->d2_slots * D2_STRIDE) c.q->wd[c.q->o_d2tab + i] = 0.0;
      PAR_FOR(i, 2 * c.q->n_trom * X2_STRIDE) c.q->wd[c.q->o_x2tab + i] = 0.0;
    }
  }
  fill_sample_cache(c, mode == EV_FULL);      // ends with a sync (also orders kzero before the kadd's below)
  if (mode == EV_FULL) TACC(c, 21, CHD_CLOCK() - tic_);
  long long te_ = CHD_CLOCK();
  eval_rows(c, mode, c_out, lam);
  const double f = c.sf * eval_cost_value(c);
  if (mode == EV_FULL) {
    CHD_SYNC();
    TACC(c, 22, CHD_CLOCK() - te_); te_ = CHD_CLOCK();
    eval_cost_grad_hess(c, g, lam);
    c.err = block_max(c, (double)c.err) > 0.5 ? 1 : 0;     // a band overflow seen by any thread
    if (c.pm_dirty) csr_rebuild(c);                        // (uniform: the flag is read behind the reduction's barriers) the evaluation wrote entries the readers' list does not hold yet
    if (CHD_TID == 0 && c.side_n > c.q->side_cap) c.side_ok = 0;      // (the side list overflowed: the second model of this point is evaluated)
    TACC(c, 23, CHD_CLOCK() - te_);
  }
  CHD_SYNC();
  TOC(c, mode == EV_FULL ? 0 : 1);
  return f;
}

// ------------------------------------------------------------------------------------------
// Interior-point solve of one stage (restates ifopt::IpoptSolver::Solve, phys_optim.cpp:567-580;
// algorithm: primal-dual log-barrier on the slack formulation, Gauss-Newton Hessian with
// Levenberg damping, l1 merit line search with second-order correction).
// ------------------------------------------------------------------------------------------
struct StageResult { int status, iters, n_factor, stalled; double kkt, viol, obj, mu; };

CHD_DEV double compl_error(LCtx& c, double mu) {
  const GI* fl = c.q->wi + c.q->o_flags;
  const GD* s = VM(c, VM_S), *l = VM(c, VM_L), *u = VM(c, VM_U), *zL = VM(c, VM_ZL), *zU = VM(c, VM_ZU);
  double e = 0;
  PAR_FOR(i, c.m) {
    if (fl[i] & RF_L) e = fmax(e, fabs((s[i] - l[i]) * zL[i] - mu));
    if (fl[i] & RF_U) e = fmax(e, fabs((u[i] - s[i]) * zU[i] - mu));
  }
  return block_max(c, e);
}

CHD_DEV double barrier_val(LCtx& c, const GD* ss, double mu) {
  const GI* fl = c.q->wi + c.q->o_flags;
  const GD* l = VM(c, VM_L), *u = VM(c, VM_U);
  double b = 0;
  PAR_FOR(i, c.m) {
    if (fl[i] & RF_L) b -= mu * log(ss[i] - l[i]);
    if (fl[i] & RF_U) b -= mu * log(u[i] - ss[i]);
  }
  return block_sum(c, b);
}

CHD_DEV void residual(LCtx& c, const GD* cc, const GD* ss, GD* r) {
  const GI* fl = c.q->wi + c.q->o_flags;
  const GD* l = VM(c, VM_L);
  PAR_FOR(i, c.m) r[i] = (fl[i] & RF_EQ) ? cc[i] - l[i] : cc[i] - ss[i];
  CHD_SYNC();
}

#ifdef CHD_HOST_EMU
// CHD_EMU_SWITCH_CHECK=1 (tests): after model_switch the second model is ALSO evaluated the old way, and every listed entry must hold the same bits
#define CHD_SWITCH_CHECK_OFF && !std::getenv("CHD_EMU_SWITCH_OFF")
#define CHD_SWITCH_SELF_CHECK() do { if (std::getenv("CHD_EMU_SWITCH_CHECK")) { \
    std::vector<unsigned long long> sw_(c.csr_nnz); for (int e_ = 0; e_ < c.csr_nnz; ++e_) sw_[e_] = dbits(*k0_at(c, c.csr_row[e_], c.csr_col[e_])); \
    const int nn_ = c.csr_nnz, ns_ = c.side_n; c.second_model = 1; const double f2_ = eval_nlp(c, x, EV_FULL, cc, g, lam); c.second_model = 0; \
    long long bad_ = nn_ != c.csr_nnz; for (int e_ = 0; e_ < c.csr_nnz && e_ < nn_; ++e_) if (sw_[e_] != dbits(*k0_at(c, c.csr_row[e_], c.csr_col[e_]))) ++bad_; \
    if (bad_ || dbits(f2_) != dbits(f)) { std::fprintf(stderr, "MODEL SWITCH MISMATCH stage %d it %d: %lld entries (f %.17g vs %.17g)\n", S->stage, it, bad_, f, f2_); c.err = 3; } \
    else if (std::getenv("CHD_EMU_SWITCH_CHECK")[0] == '2') std::fprintf(stderr, "model switch ok: stage %d it %d, %d entries of %d switched\n", S->stage, it, ns_, nn_); } } while (0)
#else
#define CHD_SWITCH_CHECK_OFF
#define CHD_SWITCH_SELF_CHECK() ((void)0)
#endif
CHD_NOINLINE CHD_DEV void solve_stage(LCtx& c, StageResult& res) {
  QP q = c.q; SDP S = c.S;
  const int n = c.n, m = c.m, N = c.N;
  GD* x = VN(c, VN_X), *g = VN(c, VN_G), *dualx = VN(c, VN_DUALX), *dx = VN(c, VN_DX), *xt = VN(c, VN_XT), *xs = VN(c, VN_XS);
  GD* cc = VM(c, VM_C), *s = VM(c, VM_S), *zL = VM(c, VM_ZL), *zU = VM(c, VM_ZU), *lam = VM(c, VM_LAM), *l = VM(c, VM_L), *u = VM(c, VM_U),
         *sc = VM(c, VM_SC), *Sig = VM(c, VM_SIGMA), *rs = VM(c, VM_RS), *D = VM(c, VM_D), *r = VM(c, VM_R), *dlam = VM(c, VM_DLAM),
         *ds = VM(c, VM_DS), *dzL = VM(c, VM_DZL), *dzU = VM(c, VM_DZU), *st = VM(c, VM_ST), *ct = VM(c, VM_CT), *rt = VM(c, VM_RT), *ss2 = VM(c, VM_SS2);
  GD* rhs = VK(c, VK_RHS), *sol = VK(c, VK_SOL), *rhs2 = VK(c, VK_RHS2), *sol2 = VK(c, VK_SOL2), *diag = VK(c, VK_DIAG), *t1 = VK(c, VK_T1);
  GI* fl = q->wi + q->o_flags; GI* sign = q->wi + q->o_sign;
  const GD* Dw = q->cd + S->o_Dw; const GD* cl = q->cd + S->o_cl; const GD* cu = q->cd + S->o_cu;
  const GI* pos_var = c.pos_var; const GI* pos_row = c.pos_row;

  x_from_state(c, x);
  // ---- unscaled evaluation -> gradient-based scaling (nlp_scaling_max_gradient = 100)
  c.sf = 1.0;
  PAR_FOR(i, m) sc[i] = 1.0;
  PAR_FOR(i, N) sign[i] = 1;
  CHD_SYNC();
  PAR_FOR(i, m) sign[pos_row[i]] = -1;
  CHD_SYNC();
  eval_nlp(c, x, EV_FULL, cc, g);
  double gm = 0;
  PAR_FOR(j, n) gm = fmax(gm, fabs(g[j]));
  gm = block_max(c, gm);
  const double sf = gm > 100.0 ? 100.0 / gm : 1.0;
  PAR_FOR(i, m) {          // largest |J_ij| of the row = largest entry of row/column pos_row[i] of K0
    const int p = pos_row[i];
    double rm = 0;
    if (p < c.Nb) {
      const int lo = p - c.w < 0 ? 0 : p - c.w, hi = p + c.w >= c.Nb ? c.Nb - 1 : p + c.w;
      const GD* row = c.K0b + (long long)p * c.W2 + (c.w - p);
      for (int k = lo; k <= hi; ++k) rm = fmax(rm, fabs(row[k]));
      for (int rr = 0; rr < c.bc; ++rr) rm = fmax(rm, fabs(c.K0x[(long long)rr * c.LD + p]));
    } else {
      const GD* row = c.K0x + (long long)(p - c.Nb) * c.LD;
      for (int k = 0; k < c.LD; ++k) rm = fmax(rm, fabs(row[k]));
    }
    t1[i] = rm > 100.0 ? fmax(100.0 / rm, 1e-8) : 1.0;
  }
  CHD_SYNC();
  PAR_FOR(i, m) {
    sc[i] = t1[i];
    const bool eq = (cu[i] - cl[i]) <= 0.0;
    const bool hl = !eq && cl[i] > -CHD_INF, hu = !eq && cu[i] < CHD_INF;
    fl[i] = (eq ? RF_EQ : 0) | (hl ? RF_L : 0) | (hu ? RF_U : 0);
    double li = cl[i] > -CHD_INF ? cl[i] * sc[i] : -HUGE_VAL;
    double ui = cu[i] < CHD_INF ? cu[i] * sc[i] : HUGE_VAL;
    if (hl) li -= 1e-8 * fmax(1.0, fabs(li));      // bound_relax_factor
    if (hu) ui += 1e-8 * fmax(1.0, fabs(ui));
    l[i] = li; u[i] = ui;
  }
  CHD_SYNC();
  c.sf = sf;
  double f = eval_nlp(c, x, EV_FULL, cc, g);     // multipliers are initialised below: the first model has no curvature terms

  // ---- slack / multiplier initialisation
  double mu = (S->stage == 0) ? CHD_MU_INIT_COLD : CHD_MU_INIT_WARM;
  PAR_FOR(i, m) {
    double si = 0, zl = 0, zu = 0;
    if (!(fl[i] & RF_EQ)) {
      si = cc[i];
      const bool hl = fl[i] & RF_L, hu = fl[i] & RF_U;
      if (hl) { double pl = 1e-2 * fmax(1.0, fabs(l[i])); if (hu) pl = fmin(pl, 1e-2 * (u[i] - l[i])); si = fmax(si, l[i] + pl); }
      if (hu) { double pu = 1e-2 * fmax(1.0, fabs(u[i])); if (hl) pu = fmin(pu, 1e-2 * (u[i] - l[i])); si = fmin(si, u[i] - pu); }
      if (hl) zl = mu / (si - l[i]);
      if (hu) zu = mu / (u[i] - si);
    }
    s[i] = si; zL[i] = zl; zU[i] = zu; lam[i] = zu - zl;
  }
  CHD_SYNC();
  double nu = 1.0, dw = CHD_DELTA_W0;
  const double kappa_eps = 10.0, kappa_mu = 0.2, theta_mu = 1.5, smax = 100.0;
  const double tol_ = c.tol;

  int status = -1, it = 0, n_factor = 0, stalled_out = 0;
  double E0 = 0, e_d = 0, e_p = 0, e_pu = 0;
  // damping safeguard (oracle/ipm_solver.hpp, dual_rise_k): the dual infeasibility rising in six consecutive iterations raises delta_w
  double ed_prev = -1.0; int ed_rise = 0;
  for (it = 0; it < S->max_iter; ++it) {
    // ---- optimality error (IPOPT eq. (5)/(6))
    PAR_FOR(i, N) t1[i] = 0.0;
    CHD_SYNC();
    PAR_FOR(i, m) t1[pos_row[i]] = lam[i];
    CHD_SYNC();
    kmatvec(c, t1, sol2, nullptr, sign);           // J^T lam, evaluated at the variable positions only
    double d1 = 0, sumlam = 0, sumz = 0, cnt = 0, ep = 0, epu = 0;
    PAR_FOR(j, n) { const double v = g[j] + sol2[pos_var[j]]; dualx[j] = v; d1 = fmax(d1, fabs(v)); }
    residual(c, cc, s, r);
    PAR_FOR(i, m) {
      sumlam += fabs(lam[i]);
      if (!(fl[i] & RF_EQ)) d1 = fmax(d1, fabs(-lam[i] - zL[i] + zU[i]));
      if (fl[i] & RF_L) { sumz += fabs(zL[i]); cnt += 1; }
      if (fl[i] & RF_U) { sumz += fabs(zU[i]); cnt += 1; }
      ep = fmax(ep, fabs(r[i])); epu = fmax(epu, fabs(r[i]) / sc[i]);
    }
    d1 = block_max(c, d1); sumlam = block_sum(c, sumlam); sumz = block_sum(c, sumz); cnt = block_sum(c, cnt);
    e_p = block_max(c, ep); e_pu = block_max(c, epu);
    const double s_d = fmax(smax, (sumlam + sumz) / fmax(1.0, m + cnt)) / smax;
    const double s_c = fmax(smax, sumz / fmax(1.0, cnt)) / smax;
    e_d = d1 / s_d;
    E0 = fmax(e_d, fmax(e_p, compl_error(c, 0.0) / s_c));
    if (ed_prev >= 0 && e_d > ed_prev) ++ed_rise; else ed_rise = 0;
    ed_prev = e_d;
    if (ed_rise >= CHD_DUAL_RISE_K) { dw = fmin(CHD_DELTA_W_MAX, fmax(dw, 1e-6) * 4.0); ed_rise = 0; }
    if (E0 <= tol_ && e_pu <= CHD_CONSTR_VIOL_TOL) { status = 0; break; }
    // stall guard (chd_config.stall_window, 0 = off): no factor-2 reduction of the optimality error over the last `window`
    // iterations -> status -2 instead of running to the iteration cap (stage 3 then takes the reference's stage-4
    // fallback, phys_optim.cpp:714).  Not an IPOPT rule: the per-stage statistics report it (RS_AUX2 = 1).
    if (c.stall_window > 0) {
      GD* hist = VK(c, VK_HIST);
      const int win = c.stall_window < q->max_N ? c.stall_window : q->max_N;
      const int slot = it % win;
      const bool stalled = it >= win && E0 > 0.5 * hist[slot];
      CHD_SYNC();
      if (stalled) { status = -2; stalled_out = 1; break; }
      if (CHD_TID == 0) hist[slot] = E0;
      CHD_SYNC();
    }
    // ---- monotone barrier update
    while (true) {
      const double Emu = fmax(e_d, fmax(e_p, compl_error(c, mu) / s_c));
      if (Emu <= kappa_eps * mu && mu > tol_ / 10) mu = fmax(tol_ / 10, fmin(kappa_mu * mu, pow(mu, theta_mu)));
      else break;
    }
    const double tau = fmax(0.99, 1 - mu);
    PAR_FOR(i, m) {
      if (fl[i] & RF_EQ) { Sig[i] = 0; rs[i] = 0; D[i] = CHD_DELTA_C; continue; }
      double sg = 0, qq = -lam[i];
      if (fl[i] & RF_L) { sg += zL[i] / (s[i] - l[i]); qq -= mu / (s[i] - l[i]); }
      if (fl[i] & RF_U) { sg += zU[i] / (u[i] - s[i]); qq += mu / (u[i] - s[i]); }
      Sig[i] = fmax(sg, 1e-300); rs[i] = qq; D[i] = 1.0 / Sig[i];
    }
    PAR_FOR(j, n) rhs[pos_var[j]] = -dualx[j];
    CHD_SYNC();
    PAR_FOR(i, m) rhs[pos_row[i]] = (fl[i] & RF_EQ) ? -r[i] : -(r[i] + rs[i] / Sig[i]);
    double cn = 0, gdx;
    PAR_FOR(i, m) cn += fabs(r[i]);
    cn = block_sum(c, cn);

    bool ok = false, used_soc = false;
    double ratio_num = 0.0, ratio_den = 0.0;      // actual and predicted reduction of the merit function by the accepted step (chd_config.damping_rule = 1, below)
#ifdef CHD_HOST_EMU
    double trace_dphi = 0.0;      // (CHD_EMU_TRACE: the directional derivative of the merit function the accepted step was judged with)
#endif
    double alpha = 0, a_du = 1.0;
    int nls = 0, attempt = 0;
    // Second model of an iteration: when an attempt with the exact blocks (heel-distance curvature, duration-duration and node x duration blocks)
    // fails -- wrong inertia, or the line search runs out of backtracks -- the model is rebuilt ONCE without them (plain Gauss-Newton, positive
    // semi-definite by construction) and the attempt is repeated with the same damping; only if that fails too does the damping grow.  The exact
    // blocks make the easy sequences converge in few iterations; where they are indefinite, the positive semi-definite model is the fallback.
    // (Rounds 2-3 kept the positive part of the heel-distance multipliers instead: near-redundant rows -- a sample milliseconds before a
    //  touch-down next to one in the stance that follows -- carry multipliers of +-1e4 that cancel in the exact block, and the clipped copy
    //  kept the +1e4: a spurious stiffness under which the kinematic optimisation's clips crawled to the iteration cap.  profiles/r04_curvature_study.md)
    bool second_used = false;
    auto second_model = [&]() -> bool {
      if (second_used || it == 0) return false;          // (the first model of a stage has no curvature terms)
      second_used = true;
      CHD_SYNC();
      if (c.side_ok CHD_SWITCH_CHECK_OFF) {          // K0 <- the second model of this point from the side list of the evaluation that built the first (same c, g, f)
        model_switch(c);
        if (CHD_TID == 0) c.side_ok = 0;
        CHD_SYNC();
        CHD_SWITCH_SELF_CHECK();
        return true;
      }
      if (CHD_TID == 0) c.second_model = 1;
      CHD_SYNC();
      f = eval_nlp(c, x, EV_FULL, cc, g, lam);
      if (CHD_TID == 0) c.second_model = 0;
      CHD_SYNC();
      return true;
    };
    for (attempt = 0; attempt < CHD_MAX_ATTEMPTS; ++attempt) {
      PAR_FOR(j, n) diag[pos_var[j]] = dw * Dw[j];
      PAR_FOR(i, m) diag[pos_row[i]] = -D[i];
      CHD_SYNC();
      kfactor(c, diag, sign); ++n_factor;
#if CHD_INERTIA_RETRY
      // a pivot of unexpected sign was replaced: second model / more damping instead of a step from the modified matrix
      if (factor_failed(c)) { if (second_model()) continue; dw *= 10.0; if (dw > CHD_DELTA_W_MAX) break; continue; }
#endif
      ksolve(c, rhs, sol, diag, 1);
      PAR_FOR(j, n) dx[j] = sol[pos_var[j]];
      PAR_FOR(i, m) dlam[i] = sol[pos_row[i]];
      CHD_SYNC();
      double a_pr = 1.0, adu = 1.0, dbar = 0, sSds = 0;
      PAR_FOR(i, m) {
        double dsi = 0, dl = 0, du = 0;
        if (!(fl[i] & RF_EQ)) {
          dsi = (dlam[i] - rs[i]) / Sig[i];
          if (fl[i] & RF_L) {
            const double sl = s[i] - l[i];
            dl = mu / sl - zL[i] - zL[i] / sl * dsi;
            if (dsi < 0) a_pr = fmin(a_pr, -tau * sl / dsi);
            if (dl < 0) adu = fmin(adu, -tau * zL[i] / dl);
            dbar -= mu / sl * dsi;
          }
          if (fl[i] & RF_U) {
            const double su = u[i] - s[i];
            du = mu / su - zU[i] + zU[i] / su * dsi;
            if (dsi > 0) a_pr = fmin(a_pr, tau * su / dsi);
            if (du < 0) adu = fmin(adu, -tau * zU[i] / du);
            dbar += mu / su * dsi;
          }
          sSds += Sig[i] * dsi * dsi;
        }
        ds[i] = dsi; dzL[i] = dl; dzU[i] = du;
      }
      a_pr = block_min(c, a_pr); a_du = block_min(c, adu); dbar = block_sum(c, dbar); sSds = block_sum(c, sSds);
      // dx^T (H + dw Dw) dx without a mat-vec, from the two block rows of the KKT system just solved:
      //   (H + dw Dw) dx + J^T dlam = -dualx ,   J dx - D dlam = rhs_row
      double dHd = 0; gdx = 0;
      PAR_FOR(j, n) { dHd -= dx[j] * dualx[j]; gdx += g[j] * dx[j]; }
      PAR_FOR(i, m) dHd -= (rhs[pos_row[i]] + D[i] * dlam[i]) * dlam[i];
      dHd = block_sum(c, dHd) + sSds; gdx = block_sum(c, gdx);
      const double dphi_bar = gdx + dbar;
      // penalty parameter of the merit function, recomputed for every step (not monotone): a value that was needed once -- typically a
      // quotient by a constraint violation at noise level -- otherwise stays for the rest of the stage, and every later step is then
      // judged by second-order changes of a violation of 1e-7 times nu = 1e3 (the duration-stage stragglers of round 2)
      {
        const double nut = cn >= 1e-6 ? (dphi_bar + 0.5 * fmax(dHd, 0.0)) / ((1 - 0.1) * cn) : 0.0;
        nu = fmax(1.0, nut * 1.1 + 1e-8);
      }
      const double Dphi = dphi_bar - nu * cn;
#ifdef CHD_HOST_EMU
      trace_dphi = Dphi;
#endif
      const double phi0 = f + barrier_val(c, s, mu) + nu * cn;
      alpha = a_pr; ok = false; nls = 0; used_soc = false;
      while (nls <= CHD_MAX_BACKTRACK) {
        PAR_FOR(j, n) xt[j] = x[j] + alpha * dx[j];
        PAR_FOR(i, m) st[i] = s[i] + alpha * ds[i];
        CHD_SYNC();
        const double ft = eval_nlp(c, xt, EV_VALUES, ct, nullptr);
        residual(c, ct, st, rt);
        double cnt_ = 0;
        PAR_FOR(i, m) cnt_ += fabs(rt[i]);
        cnt_ = block_sum(c, cnt_);
        const double phit = ft + barrier_val(c, st, mu) + nu * cnt_;
        if (phit <= phi0 + 1e-4 * alpha * Dphi + 1e-12 * fabs(phi0)) { ok = true; ratio_num = phi0 - phit; ratio_den = -(alpha * Dphi + 0.5 * alpha * alpha * fmax(dHd, 0.0)); break; }
        if (nls == 0 && cnt_ > 1e-12) {
          // second-order correction: same factorisation, constraint residual of the trial point
          PAR_FOR(j, n) rhs2[pos_var[j]] = 0.0;
          CHD_SYNC();
          PAR_FOR(i, m) rhs2[pos_row[i]] = -rt[i];
          CHD_SYNC();
          ksolve(c, rhs2, sol2, diag, 1);
          double inside = 1.0;
          PAR_FOR(j, n) xs[j] = xt[j] + sol2[pos_var[j]];
          PAR_FOR(i, m) {
            double v = st[i];
            if (!(fl[i] & RF_EQ)) {
              v = st[i] + sol2[pos_row[i]] / Sig[i];
              // (1 - 1e-8): a slack that limited the step (alpha = a_pr) sits exactly on this boundary and the correction leaves it
              // there to rounding, which made the test a coin flip between implementations (seed 219)
              if ((fl[i] & RF_L) && v - l[i] < (1 - 1e-8) * (1 - tau) * (s[i] - l[i])) inside = 0.0;
              if ((fl[i] & RF_U) && u[i] - v < (1 - 1e-8) * (1 - tau) * (u[i] - s[i])) inside = 0.0;
            }
            ss2[i] = v;
          }
          inside = block_min(c, inside);
          if (inside > 0.5) {
            const double fs = eval_nlp(c, xs, EV_VALUES, ct, nullptr);
            residual(c, ct, ss2, rt);
            double cns = 0;
            PAR_FOR(i, m) cns += fabs(rt[i]);
            cns = block_sum(c, cns);
            const double phis = fs + barrier_val(c, ss2, mu) + nu * cns;
            if (phis <= phi0 + 1e-4 * alpha * Dphi + 1e-12 * fabs(phi0)) { ok = true; used_soc = true; ratio_num = phi0 - phis; ratio_den = -(alpha * Dphi + 0.5 * alpha * alpha * fmax(dHd, 0.0)); break; }
          }
        }
        alpha *= 0.5; ++nls;
      }
      if (ok) break;
      if (second_model()) continue;
      dw *= 10.0;
      if (dw > CHD_DELTA_W_MAX) break;
    }
#ifdef CHD_HOST_EMU
    if (std::getenv("CHD_EMU_TRACE")) std::fprintf(stderr, "TRACE stage %d it %d E0 %.17g ed %.17g f %.17g mu %g dw %g att %d nls %d soc %d alpha %.17g nfact %d nu %g cn %g ep %g Dphi %g\n", S->stage, it, E0, e_d, f, mu, dw, attempt, nls, (int)used_soc, alpha, n_factor, nu, cn, e_p, ok ? trace_dphi : 0.0);
#endif
    if (!ok) { status = -2; break; }
    // the damping follows the exact model: halved after a clean step of the first model, raised by half when the iteration had to fall back to the second
    // model -- so that it settles where the exact model just passes the pivot test, as IPOPT's inertia correction does, instead of staying at a level where
    // every iteration pays a failed factorisation and takes a Gauss-Newton step (the two-cycles of the kinematic optimisation's clips: 1 700 iterations in
    // stage 2.1 with the dual infeasibility alternating between 0.024 and 0.031; profiles/r04_curvature_study.md 6)
    // chd_config.damping_rule = 1 (round 6; off by default): the damping also grows, like after a backtrack, when the ACCEPTED step delivered less than a quarter of the reduction
    // the quadratic model of the merit function promised (Levenberg-Marquardt's ratio test; a model that predicts no reduction counts as ratio 0).  With the rule off only rejections
    // move the damping up.  Few or long sequences per call gain from it, a launch of thousands of walks loses 7 %: include/chd_phys.h, profiles/r06_globalisation_study.md.
    const bool poor_ratio = q->ratio_low > 0.0 && !(ratio_num >= q->ratio_low * ratio_den);
    if (nls >= 1 || poor_ratio) dw *= 4.0;
    else if (attempt == 0) dw = fmax(CHD_DELTA_W_MIN, dw / 2.0);
    else if (second_used) dw *= CHD_DW_GROW_SECOND;
    PAR_FOR(j, n) x[j] = used_soc ? xs[j] : x[j] + alpha * dx[j];
    PAR_FOR(i, m) {
      s[i] = used_soc ? ss2[i] : s[i] + alpha * ds[i];
      lam[i] += alpha * dlam[i];
      double zl = zL[i] + a_du * dzL[i], zu = zU[i] + a_du * dzU[i];
      const double ks = 1e10;
      if (fl[i] & RF_L) { const double sl = s[i] - l[i]; zl = fmin(fmax(zl, mu / (ks * sl)), ks * mu / sl); }
      if (fl[i] & RF_U) { const double su = u[i] - s[i]; zu = fmin(fmax(zu, mu / (ks * su)), ks * mu / su); }
      zL[i] = zl; zU[i] = zu;
    }
    CHD_SYNC();
    f = eval_nlp(c, x, EV_FULL, cc, g, lam);
    if (c.err) { status = -3; ++it; break; }
  }
  state_from_x(c, x);
  double cv = 0;
  PAR_FOR(i, m) { const double v = cc[i] / sc[i]; cv = fmax(cv, fmax(cl[i] - v, v - cu[i])); }
  cv = block_max(c, cv);
  res.status = c.err ? -3 : status; res.iters = it; res.n_factor = n_factor; res.kkt = E0; res.viol = fmax(cv, 0.0); res.obj = f / c.sf; res.mu = mu; res.stalled = stalled_out;
}

// ------------------------------------------------------------------------------------------
// SaveSolution (phys_optim.cpp:63-143): resample the splines at the data rate
// ------------------------------------------------------------------------------------------
CHD_NOINLINE CHD_DEV void sample_solution(QP q, int snap) {
  const int cap = q->cap;
  GD* od = q->out_d + N_STAGES * RS_STRIDE + (long long)snap * 10 * cap * 3;
  GI* oi = q->out_i;
  const double tot = q->wd[q->o_ttot + 0];                 // solution.base_linear_->GetTotalTime() (:69)
  // number of samples of `while (t <= tot + 1e-5)` with t accumulated by += dt
  int ns = 0;
  { double t = 0; while (t <= tot + 1e-5 && ns < cap) { ++ns; t += q->dt; } }
  if (CHD_TID == 0) { oi[snap * 2] = ns; oi[snap * 2 + 1] = (int)((tot + 1e-5) / q->dt) + 1; }
  const GD* tc = q->cd + q->o_tcost;
  PAR_FOR(idx, ns * 10) {
    const int i = idx / 10, b = idx % 10;
    double t = 0;
    if (i <= q->F + 1) t = tc[i]; else { for (int k = 0; k < i; ++k) t += q->dt; }
    PE e;
    const int s = b < 2 ? b : (b < 6 ? b : b);          // blocks: 0 base_lin, 1 base_ang, 2..5 ee_pos, 6..9 ee_force
    spline_eval(q, s, t, e);
    GD* o = od + ((long long)b * cap + i) * 3;
    if (b == 1) for (int k = 0; k < 3; ++k) o[k] = e.p[k] / M_PI * 180;     // :97
    else for (int k = 0; k < 3; ++k) o[k] = e.p[k];
    if (b >= 2 && b < 6) {
      const int ee = b - 2;
      const int ph = phase_lookup(q, ee, t);                                  // TOWR PhaseDurations::IsContactPhase
      const int cflag = (ph % 2 == 0) ? q->start_contact[ee] : !q->start_contact[ee];
      oi[8 + ((long long)snap * 4 + ee) * cap + i] = cflag ? 1 : 0;
    }
  }
  CHD_SYNC();
}

// ------------------------------------------------------------------------------------------
// One sequence, stages stage_first..stage_last (the whole of main() after the readers)
// ------------------------------------------------------------------------------------------
CHD_DEV void bind_stage(LCtx& c, QP q, int stage) {
  if (CHD_TID == 0) {          // the context is shared by the workgroup
    c.q = q; c.S = &q->st[stage];
    c.n = c.S->n; c.m = c.S->m; c.N = c.n + c.m; c.Nb = c.S->Nb; c.bc = c.S->bc; c.w = c.S->w;
    c.W2 = 2 * c.w + 1; c.LD = c.N;
    c.MW = (c.W2 + 63) >> 6; c.LW = (c.LD + 63) >> 6; c.CW = (c.bc + 63) >> 6;
    c.pmb = (GU*)(q->wd + q->o_pmb); c.pmx = (GU*)(q->wd + q->o_pmx); c.pmt = (GU*)(q->wd + q->o_pmt);
    c.csr_rp = q->wi + q->o_csr_rp; c.csr_col = q->wi + q->o_csr_col; c.csr_row = q->wi + q->o_csr_row; c.csr_cap = q->csr_cap; c.csr_nnz = 0; c.pm_dirty = 0; c.side_n = 0; c.side_ok = 0;
    c.K0b = q->wd + q->o_K0b; c.K0x = q->wd + q->o_K0x; c.Kfb = q->wd + q->o_Kfb; c.Kfx = q->wd + q->o_Kfx;
    c.pos_var = q->ci + c.S->o_pos_var; c.pos_row = q->ci + c.S->o_pos_row; c.env = q->wi + q->o_envw; c.rcnt = q->wi + q->o_rcntw;
    c.sf = 1.0; c.err = 0; c.n_bad_pivots = 0; c.second_model = 0;
  }
  CHD_SYNC();
}

CHD_DEV void init_state(QP q) {
  PAR_FOR(k, q->tot_entries) q->wd[q->o_node + k] = q->cd[q->o_node0 + k];
  PAR_FOR(k, q->tot_phases) q->wd[q->o_phase_dur + k] = q->cd[q->o_phase_dur_in + k];
  // base splines: fixed 0.1 s polynomials (parameters.cpp:109-125)
  for (int b = 0; b < 2; ++b) {
    const auto& sp = q->sp[b];
    PAR_FOR(p, sp.n_polys) {
      double left = q->T;
      for (int k = 0; k < p; ++k) left -= 0.1;
      q->wd[q->o_poly_dur + sp.poly_off + p] = left > 0.1 ? 0.1 : left;
    }
  }
  CHD_SYNC();
  refresh_durations(q);
}

// The workspace belongs to the workgroup, not to the sequence: what a later launch needs of a sequence's state (the
// stage-4 fallback starts from the node values and durations stage 3 left, phys_optim.cpp:714-749) is kept with the
// sequence's results.
// Three slots, one per output snapshot (node values + phase durations as the stage that produced the snapshot left them): slot 2 is what the fallback
// launch starts from; all three are what chd_debug_get_state hands to the tests' independent evaluator (the oracle's model functions at the returned point).
CHD_DEV GD* saved_state(QP q, int slot) { return q->out_d + out_d_state_off(q->cap) + (long long)slot * (q->tot_entries + q->tot_phases); }
CHD_DEV void save_state(QP q, int slot) {
  GD* st = saved_state(q, slot);
  PAR_FOR(k, q->tot_entries) st[k] = q->wd[q->o_node + k];
  PAR_FOR(k, q->tot_phases) st[q->tot_entries + k] = q->wd[q->o_phase_dur + k];
  CHD_SYNC();
}
CHD_DEV void load_state(QP q) {
  const GD* st = saved_state(q, 2);
  PAR_FOR(k, q->tot_entries) q->wd[q->o_node + k] = st[k];
  PAR_FOR(k, q->tot_phases) q->wd[q->o_phase_dur + k] = st[q->tot_entries + k];
  CHD_SYNC();
  refresh_durations(q);
}

CHD_DEV void reset_context(LCtx& c, LdsD* lds, int lds_cap, double tol, int stall_window) {
  if (CHD_TID == 0) {
    c.lds = lds; c.lds_cap = lds_cap; c.tol = tol; c.stall_window = stall_window;
    for (int k = 0; k < 24; ++k) c.tacc[k] = 0;
  }
  CHD_SYNC();
}

CHD_DEV void run_sequence(QP q, LCtx& c, LdsD* lds, int lds_cap, double tol, int stall_window, int stage_first, int stage_last) {
  reset_context(c, lds, lds_cap, tol, stall_window);
  const long long t_begin = CHD_CLOCK();
  // the workgroup's workspace still holds the previous sequence: everything below the KKT storage (state, solver
  // vectors, tables) starts from zero; the KKT storage is cleared stage by stage (kreset)
  for (long long i = CHD_TID; i < q->o_K0b; i += CHD_NT) q->wd[i] = 0.0;
  CHD_SYNC();
  init_state(q);
  if (stage_first != 0) load_state(q);
  for (int stage = stage_first; stage <= stage_last; ++stage) {
    GD* rs = q->out_d + stage * RS_STRIDE;
    if (!q->st[stage].valid) { if (CHD_TID == 0) { rs[RS_STATUS] = -3; rs[RS_ITERS] = 0; } continue; }
    bind_stage(c, q, stage);
    kreset(c);
    StageResult r;
    solve_stage(c, r);
    if (CHD_TID == 0) {
      rs[RS_STATUS] = r.status; rs[RS_ITERS] = r.iters; rs[RS_KKT] = r.kkt; rs[RS_VIOL] = r.viol; rs[RS_OBJ] = r.obj;
      rs[RS_MU] = r.mu; rs[RS_NFACT] = r.n_factor; rs[RS_AUX] = r.stalled;
    }
    if (stage == 1) { sample_solution(q, 0); save_state(q, 0); }       // sol_out_no_dynamics.txt  (phys_optim.cpp:603)
    if (stage == 3) { sample_solution(q, 1); save_state(q, 1); }       // sol_out_dynamics.txt     (:661)
    if (stage == 4 || stage == 5) { sample_solution(q, 2); save_state(q, 2); }   // sol_out_durations.txt (:757); stage 4's is also what the fallback launch starts from
    CHD_SYNC();
  }
  if (CHD_TID == 0) {      // phase timers (100 MHz wall clock ticks), accumulated over launches
    c.tacc[5] = CHD_CLOCK() - t_begin;
    GD* tm = q->out_d + N_STAGES * RS_STRIDE + 3LL * 10 * q->cap * 3;
    for (int k = 0; k < 24; ++k) tm[k] += (double)c.tacc[k];
  }
  CHD_SYNC();
}

// Debug entry: evaluate stage `stage` at the initial state (or at xin), with the exact duration block for the multipliers lamin if given.
CHD_DEV void debug_eval(QP q, LCtx& c, int stage, int use_x, LdsD* lds, int lds_cap, const GD* xin, const GD* lamin, double* f_out) {
  reset_context(c, lds, lds_cap, 1e-3, 0);
  for (long long i = CHD_TID; i < q->o_K0b; i += CHD_NT) q->wd[i] = 0.0;
  CHD_SYNC();
  init_state(q);
  bind_stage(c, q, stage);
  kreset(c);
  GD* x = VN(c, VN_X);
  if (use_x) { PAR_FOR(j, c.n) x[j] = xin[j]; CHD_SYNC(); state_from_x(c, x); }
  x_from_state(c, x);
  PAR_FOR(i, c.m) VM(c, VM_SC)[i] = 1.0;
  if (lamin) PAR_FOR(i, c.m) VM(c, VM_LAM)[i] = lamin[i];
  CHD_SYNC();
  const double f = eval_nlp(c, x, EV_FULL, VM(c, VM_C), VN(c, VN_G), lamin ? VM(c, VM_LAM) : nullptr);
  if (CHD_TID == 0) { f_out[0] = f; f_out[1] = c.err; }
  CHD_SYNC();
}

// Debug entry: K0 of `stage` at the initial state, + (dw Dw, -dval) on the diagonal, factored `reps` times and solved for `rhs` (one refinement step).
// out: [0] replaced pivots, [1] clock ticks (100 MHz) of the factorisations, [2] of the solve, [3] 1, [4..11] the factorisation's phase timers 6..13.
CHD_DEV void debug_linsolve(QP q, LCtx& c, int stage, LdsD* lds, int lds_cap, double dw, double dval, int which, int reps, const GD* rhs_in, GD* x_out, double* out) {
  double fo[2];
  debug_eval(q, c, stage, 0, lds, lds_cap, nullptr, nullptr, fo);
  GD* diag = VK(c, VK_DIAG); GI* sign = q->wi + q->o_sign;
  const GD* Dw = q->cd + c.S->o_Dw;
  PAR_FOR(j, c.n) { diag[c.pos_var[j]] = dw * Dw[j]; sign[c.pos_var[j]] = 1; }
  PAR_FOR(i, c.m) { diag[c.pos_row[i]] = -dval; sign[c.pos_row[i]] = -1; }
  CHD_SYNC();
  int ran = 1; (void)which;
  const long long t0 = CHD_CLOCK();
  for (int r = 0; r < reps; ++r) { kfactor_rl(c, diag, sign); CHD_SYNC(); }
  const long long t1 = CHD_CLOCK();
  GD* rhs = VK(c, VK_RHS); GD* sol = VK(c, VK_SOL);
  PAR_FOR(i, c.N) rhs[i] = rhs_in[i];
  CHD_SYNC();
  ksolve(c, rhs, sol, diag, 1);
  const long long t2 = CHD_CLOCK();
  PAR_FOR(i, c.N) x_out[i] = sol[i];
  if (CHD_TID == 0) { out[0] = c.n_bad_pivots; out[1] = (double)(t1 - t0); out[2] = (double)(t2 - t1); out[3] = ran; for (int k = 0; k < 10; ++k) out[4 + k] = (double)c.tacc[6 + k]; }
  CHD_SYNC();
}

}  // namespace chd

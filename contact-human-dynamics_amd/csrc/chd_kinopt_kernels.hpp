// chd_kinopt_kernels.hpp -- one `least_squares` solve of the kinematic optimisation for one video, one workgroup.
//
// Reference: optimize_trajectory.py:660-670 / :779-789 --
//     least_squares(fun_anim_for_projection, x0, jac=jac_anim_for_projection_sparse, max_nfev=50, gtol=1e-12, tr_solver='lsmr')
// i.e. SciPy's trust-region-reflective method without bounds in its 2-D subspace form (`trf_no_bounds`: Gauss-Newton direction
// from LSMR with the Cauchy-step regularisation, trust-region problem in span{g, gn}) on the residual of :324-483 with the
// Jacobian of :51-322.  The same source is compiled by hipcc for gfx950 (one workgroup of KO_NT threads per video) and by g++
// with -DCHD_HOST_EMU (one emulated thread) for the CPU tests.
//
// What is different from the reference, all exact in real arithmetic:
//  * the Jacobian is never formed (the reference allocates rows x (84 F) and rows x (87 F) dense arrays: 3.4 GB + 3.5 GB for
//    100 frames).  J = dE/dP * dP/dx: dE/dP is a handful of stencil coefficients per row, dP/dx per frame is
//    cross(axis_{j,a}, p_t - p_j) for joint j an ancestor of t (InverseKinematics.py:192-230), so
//        J v  : omega_j = sum_a v_{j,a} axis_{j,a};  dp_t = sum_{j anc t} omega_j x (p_t - p_j);  rows from dp
//        J^T u: lambda_t from the rows;  (J^T u)_{j,a} = axis_{j,a} . sum_{t desc j} (p_t - p_j) x lambda_t
//    with the linearisation (positions, axes, projection coefficients: 420 doubles per frame) cached per accepted point;
//    both products run frame tile by frame tile through the workgroup's LDS block (kin_jv / kin_jtu), and LSMR's two half steps
//    are fused into them (kin_lsmr): per iteration HBM sees U read + written, V / H / H-bar / x read + written and the
//    linearisation read twice -- measured 0.8 .. 1.3 x that (profiles/r02k_final/kinopt_pmc.md);
//  * every norm is a tree sum over the workgroup's 512 per-thread partial sums (KoAcc): with a single running sum per norm LSMR's
//    iterate at its iteration limit drifts 5 % away from SciPy's and the solves end 1e-3 .. 3e-3 from the reference's
//    instead of 1e-4 (tests/test_kinopt_emu.py);
//  * forward kinematics with rotation matrices instead of quaternions;
//  * span{g, gn} is orthonormalised by Gram-Schmidt instead of Householder QR (same subspace, so the same step);
//  * the boundary solution of the 2-D trust-region problem is found on the angle parametrisation (scan + bisection of the
//    derivative) instead of the roots of the tangent-half-angle quartic (numpy.roots).
// Kept on purpose: the reference Jacobian's misplaced root column in the projection rows (`varIndex + 0`, :103-137) -- it
// decides which steps get rejected, i.e. where the solve stops.
#pragma once
#include <cmath>
#ifdef CHD_HOST_EMU
#include <cstdio>
#include <cstdlib>
#define KO_TRACE(...) do { if (std::getenv("KIN_TRACE")) std::fprintf(stderr, __VA_ARGS__); } while (0)
#else
#define KO_TRACE(...) ((void)0)
#endif

#ifdef CHD_HOST_EMU
#define KO_DEV static inline
#define KO_HD inline
#define KO_TID 0
#define KO_NT 1
#define KO_SYNC() ((void)0)
#define KO_CONST static const
#else
#include <hip/hip_runtime.h>
#define KO_DEV __device__ inline
#define KO_HD __host__ __device__ inline
#define KO_TID ((int)threadIdx.x)
#define KO_NT ((int)blockDim.x)
#define KO_SYNC() __syncthreads()
#define KO_CONST __constant__ const
#endif
#define KO_FOR(i, n) for (int i = KO_TID; i < (n); i += KO_NT)
#ifdef CHD_HOST_EMU
#define KO_CLOCK() 0LL
#else
#define KO_CLOCK() ((long long)wall_clock64())
#endif

namespace chd_kin {

enum { NJ = 28, NV = 87, ROOT = 8 };
// SkeletonDefinitions.py:64-137 (combined skeleton = body-25 + three spine joints)
KO_CONST int FWD[NJ] = {8, 12, 13, 14, 21, 19, 20, 9, 10, 11, 24, 22, 23, 25, 26, 27, 1, 0, 16, 18, 15, 17, 5, 6, 7, 2, 3, 4};      // skeleton joint -> data joint
KO_CONST int BWD[NJ] = {17, 16, 25, 26, 27, 22, 23, 24, 0, 7, 8, 9, 1, 2, 3, 20, 18, 21, 19, 5, 6, 4, 11, 12, 10, 13, 14, 15};       // data joint -> skeleton joint
KO_CONST double SMOOTH_W[NJ] = {2.5, 2.5, 2.5, 1.5, 1.0, 2.5, 1.5, 1.0, 1.0, 2.5, 1.5, 1.0, 2.5, 1.5, 1.0, 1.0, 1.0, 1.0, 1.0, 1.0, 1.0, 1.0, 1.0, 1.0, 1.0, 1.5, 1.5, 1.5};
KO_CONST double SMOOTH_VEL[3] = {1.0, 1.0, 2.0};          // optimize_trajectory.py:43-45
#define KO_SMOOTH_EULER 10.0                               // :46-48 (the same for the three angles)

struct KinParams {
  int parents[NJ];
  unsigned desc[NJ];             // bit t: joint t is a strict descendant of joint j (AnimationStructure.descendants_mask)
  int max_nfev;                  // 50
  double ftol, xtol, gtol;       // 1e-8, 1e-8, 1e-12
  double atol, btol, conlim;     // LSMR: 1e-6, 1e-6, 1e8
  int lsmr_maxiter;              // 0: min(m, n) as SciPy
};

// one video inside the batch pools
struct KinSeq {
  int F, n, m;
  long long o_const;             // double pool: offsets[84] | pose3d[F*84] | root_trans[F*3] | pose2d[F*56] | proj_w[F*28] | data_w[F*28]
  long long o_contact;           // int pool: contact[F*28]
  long long o_work;              // workspace pool (doubles), work_doubles(F)
  long long o_x;                 // state pool: x[F*87] (start point in, solution out)
  double floor_n[3], floor_p[3];
  double w[6];                   // projWeight, smoothWeightVel, smoothWeightAcc, dataWeight, velWeight, floorWeight
};

KO_HD int rows_of(int F) { return 507 * F - 423; }      // 56F + 84(F-1) + 84(F-2) + 84F + 84(F-1) + 28F + 87(F-1)
KO_HD long long work_doubles(int F) { return 9LL * NV * F + 5LL * rows_of(F) + 1008LL * F + 64; }

// views into a video's workspace
struct KinWork {
  double *X, *XN, *G, *GN, *V, *H, *HB, *S0, *S1;      // n
  double *Fv, *FN, *U, *T1, *T2;                         // m
  double *P, *E, *C, *RG, *PN, *RGN;                     // per frame: 84, 252, 84, 252, 84, 252
  KO_HD void carve(double* b, int F) {
    const long long n = (long long)NV * F, m = rows_of(F);
    X = b; b += n; XN = b; b += n; G = b; b += n; GN = b; b += n; V = b; b += n; H = b; b += n; HB = b; b += n; S0 = b; b += n; S1 = b; b += n;
    Fv = b; b += m; FN = b; b += m; U = b; b += m; T1 = b; b += m; T2 = b; b += m;
    P = b; b += 84LL * F; E = b; b += 252LL * F; C = b; b += 84LL * F; RG = b; b += 252LL * F; PN = b; b += 84LL * F; RGN = b;
  }
};

struct KinCtx {
  const KinSeq* q; const KinParams* P;
  const double *offs, *pose3d, *root_trans, *pose2d, *proj_w, *data_w;
  const int* contact;
  KinWork w;
  double* red;                   // workgroup reduction scratch (LDS on the device): 3 * 16 doubles
  double* lds; int lds_doubles;  // the products' frame tiles (LDS on the device)
  long long t_jv, t_jtu, t_all;  // wall-clock ticks spent in J v / J^T u / the whole solve (first thread's view; monitoring only)
  int o2, o3, o4, o5, o6, o7;    // first row of each residual term after the projection rows
};

// ---- workgroup reductions (fixed tree: results do not depend on scheduling) -------------------------------------------------
// A sum over a KO_FOR loop: every thread adds its own items in loop order, the per-thread sums are combined by a butterfly inside
// each wavefront and then wavefront by wavefront.  The host emulation keeps 512 lane accumulators and combines them the same
// way, so that its sums round like the 512-thread workgroup's (a single running sum over ~50 000 terms is 1000 times less
// accurate, and LSMR's convergence over thousands of iterations feels that).
#ifdef CHD_HOST_EMU
enum { KO_LANES = 512 };
struct KoAcc {
  double l[KO_LANES];
  KoAcc() { for (int i = 0; i < KO_LANES; ++i) l[i] = 0.0; }
  void add(long long i, double v) { l[i & (KO_LANES - 1)] += v; }
  double total() const {
    double s = 0.0;
    for (int w0 = 0; w0 < KO_LANES; w0 += 64) {
      double a[64], t[64];
      for (int k = 0; k < 64; ++k) a[k] = l[w0 + k];
      for (int o = 32; o > 0; o >>= 1) { for (int k = 0; k < 64; ++k) t[k] = a[k] + a[k ^ o]; for (int k = 0; k < 64; ++k) a[k] = t[k]; }
      s += a[0];
    }
    return s;
  }
};
KO_DEV void ko_total3(KinCtx&, const KoAcc& A, const KoAcc& B, const KoAcc& C, double& a, double& b, double& c) { a = A.total(); b = B.total(); c = C.total(); }
#else
struct KoAcc {
  double s = 0.0;
  __device__ void add(long long, double v) { s += v; }
};
KO_DEV void ko_total3(KinCtx& c_, const KoAcc& A, const KoAcc& B, const KoAcc& C, double& a, double& b, double& c) {
  a = A.s; b = B.s; c = C.s;
  for (int o = 32; o > 0; o >>= 1) { a += __shfl_xor(a, o); b += __shfl_xor(b, o); c += __shfl_xor(c, o); }
  const int wv = threadIdx.x >> 6, nw = blockDim.x >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) { c_.red[wv] = a; c_.red[16 + wv] = b; c_.red[32 + wv] = c; }
  __syncthreads();
  double sa = 0, sb = 0, sc = 0;
  for (int i = 0; i < nw; ++i) { sa += c_.red[i]; sb += c_.red[16 + i]; sc += c_.red[32 + i]; }
  a = sa; b = sb; c = sc;
}
#endif
KO_DEV double ko_total(KinCtx& c, const KoAcc& A) { KoAcc z1, z2; double a, b, d; ko_total3(c, A, z1, z2, a, b, d); return a; }
KO_DEV double ko_dot(KinCtx& c, const double* a, const double* b, long long n) {
  KoAcc s;
  for (long long i = KO_TID; i < n; i += KO_NT) s.add(i, a[i] * b[i]);
  return ko_total(c, s);
}

// ---- forward kinematics of one frame (Animation.py:294-323, 379-414; Quaternions.from_euler(order='xyz', world=True)) -------
// R = Rz Ry Rx per joint; global rotation Rg_j = Rg_parent R_j; position p_j = p_parent + Rg_parent offset_j, root at 0
// (the fitted skeleton's root offset is zero; the root's translation is a separate unknown).  With `axes`: the rotation axes
// of the 84 angle unknowns (InverseKinematics.py:205-209): prs Rz Ry e_x, prs Rz e_y, prs e_z, prs = parent's global rotation.
KO_DEV void fk_frame(const KinCtx& c, const double* xf, double* Pf, double* RGf, double* Ef) {
  for (int j = 0; j < NJ; ++j) {
    const double ex = xf[3 + 3 * j], ey = xf[4 + 3 * j], ez = xf[5 + 3 * j];
    const double cx = std::cos(ex), sx = std::sin(ex), cy = std::cos(ey), sy = std::sin(ey), cz = std::cos(ez), sz = std::sin(ez);
    // Rz Ry Rx
    const double R[9] = {cz * cy, cz * sy * sx - sz * cx, cz * sy * cx + sz * sx,
                         sz * cy, sz * sy * sx + cz * cx, sz * sy * cx - cz * sx,
                         -sy, cy * sx, cy * cx};
    const int p = c.P->parents[j];
    double* Rg = RGf + 9 * j;
    if (p < 0) {
      for (int k = 0; k < 9; ++k) Rg[k] = R[k];
      Pf[0] = Pf[1] = Pf[2] = 0.0;
      if (Ef) {
        const double ax[9] = {cz * cy, sz * cy, -sy, -sz, cz, 0.0, 0.0, 0.0, 1.0};
        for (int k = 0; k < 9; ++k) Ef[k] = ax[k];
      }
    } else {
      const double* Rp = RGf + 9 * p;
      for (int r = 0; r < 3; ++r)
        for (int k = 0; k < 3; ++k) Rg[3 * r + k] = Rp[3 * r] * R[k] + Rp[3 * r + 1] * R[3 + k] + Rp[3 * r + 2] * R[6 + k];
      const double* o = c.offs + 3 * j;
      for (int r = 0; r < 3; ++r) Pf[3 * j + r] = Pf[3 * p + r] + Rp[3 * r] * o[0] + Rp[3 * r + 1] * o[1] + Rp[3 * r + 2] * o[2];
      if (Ef) {
        const double l0[3] = {cz * cy, sz * cy, -sy}, l1[3] = {-sz, cz, 0.0};
        double* e = Ef + 9 * j;
        for (int r = 0; r < 3; ++r) {
          e[r] = Rp[3 * r] * l0[0] + Rp[3 * r + 1] * l0[1] + Rp[3 * r + 2] * l0[2];
          e[3 + r] = Rp[3 * r] * l1[0] + Rp[3 * r + 1] * l1[1];
          e[6 + r] = Rp[3 * r + 2];
        }
      }
    }
  }
}

// y of the reference (:356-359) for data joint jd of frame f: the root's entry is its translation, the others are root relative
KO_DEV double y_of(const double* x, const double* P, int f, int jd, int cc) {
  return jd == ROOT ? x[(long long)f * NV + cc] : P[(long long)f * 84 + 3 * BWD[jd] + cc];
}

// ---- residual (:324-483) at x into out; P/RG receive the forward kinematics of x -------------------------------------------
KO_DEV void kin_residual(KinCtx& c, const double* x, double* P, double* RG, double* out) {
  const int F = c.q->F;
  KO_FOR(f, F) fk_frame(c, x + (long long)f * NV, P + (long long)f * 84, RG + (long long)f * 252, nullptr);
  KO_SYNC();
  const double pw = c.q->w[0], sv = c.q->w[1], sa = c.q->w[2], dw = c.q->w[3], vw = c.q->w[4], fw = c.q->w[5];
  KO_FOR(idx, F * NJ) {
    const int f = idx / NJ, jd = idx % NJ;
    double y[3], yr[3];
    for (int k = 0; k < 3; ++k) { y[k] = y_of(x, P, f, jd, k); yr[k] = x[(long long)f * NV + k]; }
    // projection
    {
      const double w = c.proj_w[idx];
      double r0 = 0, r1 = 0;
      if (w > 0) {
        const double ax = jd == ROOT ? yr[0] : y[0] + yr[0], ay = jd == ROOT ? yr[1] : y[1] + yr[1], az = jd == ROOT ? yr[2] : y[2] + yr[2];
        r0 = pw * w * (ax / az - c.pose2d[2 * idx]);
        r1 = pw * w * (ay / az - c.pose2d[2 * idx + 1]);
      }
      out[2 * idx] = r0; out[2 * idx + 1] = r1;
    }
    for (int k = 0; k < 3; ++k) {
      if (f < F - 1) out[c.o2 + 3 * idx + k] = sv * SMOOTH_W[jd] * SMOOTH_VEL[k] * (y[k] - y_of(x, P, f + 1, jd, k));
      if (f < F - 2) {
        const double y1 = y_of(x, P, f + 1, jd, k), y2 = y_of(x, P, f + 2, jd, k);
        out[c.o3 + 3 * idx + k] = sa * ((y2 - y1) - (y1 - y[k]));
      }
      const double tgt = jd == ROOT ? c.root_trans[3 * f + k] : c.pose3d[3 * idx + k];
      out[c.o4 + 3 * idx + k] = dw * (y[k] - tgt) * c.data_w[idx];
    }
    const bool ct = c.contact[idx] == 1;
    if (f < F - 1)
      for (int k = 0; k < 3; ++k)
        out[c.o5 + 3 * idx + k] = ct ? vw * ((yr[k] + y[k]) - (x[(long long)(f + 1) * NV + k] + y_of(x, P, f + 1, jd, k))) : 0.0;
    double d = 0;
    for (int k = 0; k < 3; ++k) d += c.q->floor_n[k] * (yr[k] + y[k] - c.q->floor_p[k]);
    out[c.o6 + idx] = ct ? fw * d : 0.0;
  }
  KO_FOR(idx, (F - 1) * NV) out[c.o7 + idx] = sv * KO_SMOOTH_EULER * (x[idx] - x[idx + NV]);
  KO_SYNC();
}

// ---- linearisation at x: positions, axes, projection coefficients ------------------------------------------------------------
KO_DEV void kin_linearise(KinCtx& c, const double* x) {
  const int F = c.q->F;
  KO_FOR(f, F) fk_frame(c, x + (long long)f * NV, c.w.P + (long long)f * 84, c.w.RG + (long long)f * 252, c.w.E + (long long)f * 252);
  KO_SYNC();
  const double pw = c.q->w[0];
  KO_FOR(idx, F * NJ) {
    const int f = idx / NJ, jd = idx % NJ;
    const double w = c.proj_w[idx];
    double cx = 0, czx = 0, czy = 0;
    if (w > 0) {
      double a[3];
      for (int k = 0; k < 3; ++k) a[k] = jd == ROOT ? x[(long long)f * NV + k] : y_of(x, c.w.P, f, jd, k) + x[(long long)f * NV + k];
      const double ww = pw * w;
      cx = ww / a[2]; czx = -ww * a[0] / (a[2] * a[2]); czy = -ww * a[1] / (a[2] * a[2]);
    }
    c.w.C[3 * idx] = cx; c.w.C[3 * idx + 1] = czx; c.w.C[3 * idx + 2] = czy;
  }
  KO_SYNC();
}

// ---- the two products, matrix free, frame tile by frame tile through LDS -------------------------------------------------------
// A tile is as many consecutive frames as the workgroup's LDS block holds.  Everything a phase reads more than once -- joint
// positions, the per-joint angular velocities / position increments (J v), the per-joint multipliers (J^T u) -- lives in LDS
// for the tile, so the ancestor / descendant walks and the misplaced-root sums are LDS reads, and the intermediates never
// touch HBM.  J v needs the position increments of frames f, f + 1, f + 2 for the rows of frame f: a tile carries two halo frames.

// out = J v (:51-322 applied to a vector).
// With `sumsq`: the fused form LSMR's bidiagonalisation needs, out <- scale * (J v) + keep * out in place and *sumsq = |out|^2
// (every row has exactly one writer; saves writing J v, reading it back and a separate pass for the norm).
KO_DEV void kin_jv(KinCtx& c, const double* v, double* out, const double scale = 1.0, const double keep = 0.0, double* sumsq = nullptr) {
  const int F = c.q->F;
  const double sv = c.q->w[1], sa = c.q->w[2], dw = c.q->w[3], vw = c.q->w[4], fw = c.q->w[5];
  const bool fused = sumsq != nullptr;
  KoAcc acc;
  long long cur = 0;                          // the loop index the running sum belongs to
  auto put = [&](long long r, double val) {
    if (fused) { val = scale * val + keep * out[r]; acc.add(cur, val * val); }
    out[r] = val;
  };
  int TF = c.lds_doubles / 252 - 2;
  if (TF < 1) TF = 1;
  double* LP = c.lds; double* LW = LP + 84 * (TF + 2); double* LD = LW + 84 * (TF + 2);
  for (int f0 = 0; f0 < F; f0 += TF) {
    const int nf = F - f0 < TF ? F - f0 : TF;
    const int nh = F - f0 < nf + 2 ? F - f0 : nf + 2;                 // with the halo
    KO_FOR(idx, nh * NJ) {                    // positions into LDS; omega_j = sum_a v_{j,a} axis_{j,a}
      const int f = f0 + idx / NJ, j = idx % NJ;
      const double* pp = c.w.P + (long long)f * 84 + 3 * j;
      const double* e = c.w.E + (long long)f * 252 + 9 * j;
      const double* vv = v + (long long)f * NV + 3 + 3 * j;
      const double v0 = vv[0], v1 = vv[1], v2 = vv[2];
      for (int k = 0; k < 3; ++k) { LP[3 * idx + k] = pp[k]; LW[3 * idx + k] = v0 * e[k] + v1 * e[3 + k] + v2 * e[6 + k]; }
    }
    KO_SYNC();
    KO_FOR(idx, nh * NJ) {                    // dp_t = sum over the strict ancestors j of t of omega_j x (p_t - p_j); stored in data order
      const int fl = idx / NJ, t = idx % NJ;
      double d[3] = {0, 0, 0};
      if (t == 0) { for (int k = 0; k < 3; ++k) d[k] = v[(long long)(f0 + fl) * NV + k]; }
      else {
        const double* Pf = LP + 84 * fl;
        const double p0 = Pf[3 * t], p1 = Pf[3 * t + 1], p2 = Pf[3 * t + 2];
        for (int j = c.P->parents[t]; j >= 0; j = c.P->parents[j]) {
          const double* om = LW + 84 * fl + 3 * j;
          const double r0 = p0 - Pf[3 * j], r1 = p1 - Pf[3 * j + 1], r2 = p2 - Pf[3 * j + 2];
          d[0] += om[1] * r2 - om[2] * r1; d[1] += om[2] * r0 - om[0] * r2; d[2] += om[0] * r1 - om[1] * r0;
        }
      }
      double* o = LD + 84 * fl + 3 * FWD[t];
      o[0] = d[0]; o[1] = d[1]; o[2] = d[2];
    }
    KO_SYNC();
    KO_FOR(idx, nf * NJ) {                    // the rows of frame f (the reference's order: term by term, frame, joint, coordinate)
      cur = idx;
      const int fl = idx / NJ, jd = idx % NJ, f = f0 + fl;
      const long long gi = (long long)f * NJ + jd;
      const double* dy = LD + 84 * fl + 3 * jd;
      const double* d0 = LD + 84 * fl;                               // data joint 0: where the reference puts the root's projection derivative
      const double* dr = LD + 84 * fl + 3 * ROOT;
      const double* C = c.w.C + 3 * gi;
      const double C0 = C[0], C1 = C[1], C2 = C[2];
      const bool ct = c.contact[gi] == 1;
      const double dwj = dw * c.data_w[gi];
      const bool has1 = f < F - 1, has2 = f < F - 2;
      // the fifteen rows of this (frame, joint): row index, whether it exists, its value.  In the fused form the old entries of all of them
      // are requested before the first one is used (one after the other they were fifteen dependent HBM round trips per item)
      long long rr[15]; bool on[15]; double val[15];
      const double ex = jd == 0 ? dy[0] : d0[0] + dy[0], ey = jd == 0 ? dy[1] : d0[1] + dy[1], ez = jd == 0 ? dy[2] : d0[2] + dy[2];
      rr[0] = 2 * gi; on[0] = true; val[0] = C0 * ex + C1 * ez;
      rr[1] = 2 * gi + 1; on[1] = true; val[1] = C0 * ey + C2 * ez;
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        const int o = 2 + 4 * k;
        rr[o] = c.o2 + 3 * gi + k; on[o] = has1; val[o] = sv * SMOOTH_W[jd] * SMOOTH_VEL[k] * (dy[k] - dy[84 + k]);
        rr[o + 1] = c.o5 + 3 * gi + k; on[o + 1] = has1; val[o + 1] = ct ? vw * ((dr[k] + dy[k]) - (dr[84 + k] + dy[84 + k])) : 0.0;
        rr[o + 2] = c.o3 + 3 * gi + k; on[o + 2] = has2; val[o + 2] = sa * (dy[k] - 2.0 * dy[84 + k] + dy[168 + k]);
        rr[o + 3] = c.o4 + 3 * gi + k; on[o + 3] = true; val[o + 3] = dwj * dy[k];
      }
      double d = 0;
      for (int k = 0; k < 3; ++k) d += c.q->floor_n[k] * (dr[k] + dy[k]);
      rr[14] = c.o6 + gi; on[14] = true; val[14] = ct ? fw * d : 0.0;
      if (fused) {
        double old[15];
#pragma unroll
        for (int q = 0; q < 15; ++q) old[q] = out[on[q] ? rr[q] : 0];
#pragma unroll
        for (int q = 0; q < 15; ++q) { val[q] = scale * val[q] + keep * old[q]; acc.add(cur, on[q] ? val[q] * val[q] : 0.0); }
      }
#pragma unroll
      for (int q = 0; q < 15; ++q) if (on[q]) out[rr[q]] = val[q];
    }
    KO_FOR(idx, nf * NV) {                    // Euler-angle smoothness rows of the tile's frames
      cur = idx;
      const long long g0 = (long long)f0 * NV + idx;
      if (f0 + idx / NV < F - 1) put(c.o7 + g0, sv * KO_SMOOTH_EULER * (v[g0] - v[g0 + NV]));
    }
    KO_SYNC();                                // the next tile overwrites the LDS block
  }
  if (fused) *sumsq = ko_total(c, acc);
}

// out = J^T u.
// With `sumsq`: out <- scale * (J^T u) + keep * out in place and *sumsq = |out|^2 (one writer per unknown).
KO_DEV void kin_jtu(KinCtx& c, const double* u, double* out, const double scale = 1.0, const double keep = 0.0, double* sumsq = nullptr) {
  const int F = c.q->F;
  const double sv = c.q->w[1], sa = c.q->w[2], dw = c.q->w[3], vw = c.q->w[4], fw = c.q->w[5];
  const double se = sv * KO_SMOOTH_EULER;
  const bool fused = sumsq != nullptr;
  KoAcc acc;
  int TF = c.lds_doubles / 336;
  if (TF < 1) TF = 1;
  double* LP = c.lds; double* LQ = LP + 84 * TF; double* LR = LQ + 84 * TF; double* LL = LR + 84 * TF;
  for (int f0 = 0; f0 < F; f0 += TF) {
    const int nf = F - f0 < TF ? F - f0 : TF;
    KO_FOR(idx, nf * NJ) {                    // positions; each joint's own projection rows and contact rows acting on a position they reference
      const int f = f0 + idx / NJ, jd = idx % NJ;
      const long long gi = (long long)f * NJ + jd;
      const bool has1 = f < F - 1, hasm = f >= 1;
      // every load is unconditional (a masked-off one reads entry 0 instead): the requests go out together, not one per branch
      const double* pp = c.w.P + (long long)f * 84 + 3 * jd;       // (LP is in skeleton order: entry jd here is skeleton joint jd)
      const double* C = c.w.C + 3 * gi;
      const double p0 = pp[0], p1 = pp[1], p2 = pp[2], C0 = C[0], C1 = C[1], C2 = C[2];
      const double ux = u[2 * gi], uy = u[2 * gi + 1];
      const bool ct = c.contact[gi] == 1, ctm = c.contact[hasm ? gi - NJ : gi] == 1 && hasm;
      const double u6 = u[c.o6 + gi];
      double u5[3], u5m[3];
#pragma unroll
      for (int k = 0; k < 3; ++k) { u5[k] = u[has1 ? c.o5 + 3 * gi + k : 0]; u5m[k] = u[hasm ? c.o5 + 3 * (gi - NJ) + k : 0]; }
      LP[3 * idx] = p0; LP[3 * idx + 1] = p1; LP[3 * idx + 2] = p2;
      LQ[3 * idx] = C0 * ux; LQ[3 * idx + 1] = C0 * uy; LQ[3 * idx + 2] = C1 * ux + C2 * uy;
      const double uf = fw * u6;
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        double r = 0.0;
        r += (ct && has1) ? vw * u5[k] : 0.0;
        r += ct ? c.q->floor_n[k] * uf : 0.0;
        r -= ctm ? vw * u5m[k] : 0.0;
        LR[3 * idx + k] = r;
      }
    }
    KO_SYNC();
    KO_FOR(idx, nf * NJ) {                    // lambda of data joint jd of frame f
      const int fl = idx / NJ, jd = idx % NJ, f = f0 + fl;
      const long long gi = (long long)f * NJ + jd;
      double lam[3];
      for (int k = 0; k < 3; ++k) lam[k] = LQ[3 * idx + k] + LR[3 * idx + k];
      if (jd == 0) for (int j = 1; j < NJ; ++j) for (int k = 0; k < 3; ++k) lam[k] += LQ[84 * fl + 3 * j + k];                      // the misplaced root column
      if (jd == ROOT) for (int j = 0; j < NJ; ++j) if (j != ROOT) for (int k = 0; k < 3; ++k) lam[k] += LR[84 * fl + 3 * j + k];   // root + joint in the contact rows
      const double dwj = dw * c.data_w[gi];
      const bool b0 = f < F - 1, b1 = f >= 1, b2 = f < F - 2, b3 = f >= 1 && f - 1 < F - 2, b4 = f >= 2;
      double t0[3], t1[3], t2[3], t3[3], t4[3], t5[3];      // all eighteen entries of u requested together (masked-off ones read entry 0)
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        t0[k] = u[b0 ? c.o2 + 3 * gi + k : 0]; t1[k] = u[b1 ? c.o2 + 3 * (gi - NJ) + k : 0];
        t2[k] = u[b2 ? c.o3 + 3 * gi + k : 0]; t3[k] = u[b3 ? c.o3 + 3 * (gi - NJ) + k : 0]; t4[k] = u[b4 ? c.o3 + 3 * (gi - 2 * NJ) + k : 0];
        t5[k] = u[c.o4 + 3 * gi + k];
      }
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        const double s = sv * SMOOTH_W[jd] * SMOOTH_VEL[k];
        lam[k] += b0 ? s * t0[k] : 0.0;
        lam[k] -= b1 ? s * t1[k] : 0.0;
        lam[k] += b2 ? sa * t2[k] : 0.0;
        lam[k] -= b3 ? 2.0 * sa * t3[k] : 0.0;
        lam[k] += b4 ? sa * t4[k] : 0.0;
        lam[k] += dwj * t5[k];
      }
      LL[3 * idx] = lam[0]; LL[3 * idx + 1] = lam[1]; LL[3 * idx + 2] = lam[2];
    }
    KO_SYNC();
    KO_FOR(idx, nf * NJ) {                    // (J^T u)_{j,a} = axis_{j,a} . sum over the strict descendants t of j of (p_t - p_j) x lambda_t
      const int fl = idx / NJ, j = idx % NJ, f = f0 + fl;
      const double* Pf = LP + 84 * fl;
      const double* L = LL + 84 * fl;
      const double q0 = Pf[3 * j], q1 = Pf[3 * j + 1], q2 = Pf[3 * j + 2];
      double M[3] = {0, 0, 0};
      const unsigned mask = c.P->desc[j];
      for (int t = j + 1; t < NJ; ++t) {
        if (!((mask >> t) & 1u)) continue;
        const double* l = L + 3 * FWD[t];
        const double r0 = Pf[3 * t] - q0, r1 = Pf[3 * t + 1] - q1, r2 = Pf[3 * t + 2] - q2;
        M[0] += r1 * l[2] - r2 * l[1]; M[1] += r2 * l[0] - r0 * l[2]; M[2] += r0 * l[1] - r1 * l[0];
      }
      const double* e = c.w.E + (long long)f * 252 + 9 * j;
      double* o = out + (long long)f * NV;
      const bool has1 = f < F - 1, hasm = f >= 1, isroot = j == 0;
      // the six unknowns this item may write (three angles; the root translation for joint 0): Euler-smoothness entries of u and, in
      // the fused form, the old values are requested up front (masked-off requests read entry 0)
      int kk[6]; double ua[6], ub[6], oldv[6], val[6];
#pragma unroll
      for (int q = 0; q < 6; ++q) {
        kk[q] = q < 3 ? 3 + 3 * j + q : q - 3;
        const bool live = q < 3 || isroot;
        ua[q] = u[(live && has1) ? c.o7 + f * NV + kk[q] : 0];
        ub[q] = u[(live && hasm) ? c.o7 + (f - 1) * NV + kk[q] : 0];
        oldv[q] = (fused && live) ? o[kk[q]] : 0.0;
      }
      double ee[9];
#pragma unroll
      for (int q = 0; q < 9; ++q) ee[q] = e[q];
#pragma unroll
      for (int q = 0; q < 6; ++q) {
        double eu = 0.0;
        eu += has1 ? se * ua[q] : 0.0;
        eu -= hasm ? se * ub[q] : 0.0;
        val[q] = (q < 3 ? ee[3 * q] * M[0] + ee[3 * q + 1] * M[1] + ee[3 * q + 2] * M[2] : L[3 * ROOT + (q - 3)]) + eu;
      }
#pragma unroll
      for (int q = 0; q < 6; ++q) {
        const bool live = q < 3 || isroot;
        if (fused) { val[q] = scale * val[q] + keep * oldv[q]; acc.add(idx, live ? val[q] * val[q] : 0.0); }
        if (live) o[kk[q]] = val[q];
      }
    }
    KO_SYNC();
  }
  if (fused) *sumsq = ko_total(c, acc);
}

// ---- LSMR (Fong & Saunders 2011, as scipy.sparse.linalg.lsmr with x0 = None) -----------------------------------------------------
KO_DEV void sym_ortho(double a, double b, double& cc, double& s, double& r) {
  auto sgn = [](double v) { return v > 0 ? 1.0 : (v < 0 ? -1.0 : 0.0); };
  if (b == 0) { cc = sgn(a); s = 0; r = std::fabs(a); }
  else if (a == 0) { cc = 0; s = sgn(b); r = std::fabs(b); }
  else if (std::fabs(b) > std::fabs(a)) { const double tau = a / b; s = sgn(b) / std::sqrt(1 + tau * tau); cc = s * tau; r = b / s; }
  else { const double tau = b / a; cc = sgn(a) / std::sqrt(1 + tau * tau); s = cc * tau; r = a / cc; }
}

// min |J x - b|^2 + damp^2 |x|^2 into c.w.GN; uses U (m), V, H, HB (n).  Returns the iteration count.
// The Golub-Kahan vectors are kept UNNORMALISED (u = su * U, v = sv * V with the scalars in registers): each half step is one
// fused product pass, beta u = A v - alpha u  ->  U <- sv * (J V) - (alpha su) * U,  beta = |U|,  su = 1 / beta, and the same for V.
KO_DEV int kin_lsmr(KinCtx& c, const double* b, double damp, int* istop_out) {
  const long long n = c.q->n, m = c.q->m;
  const KinParams& P = *c.P;
  const int maxiter = P.lsmr_maxiter > 0 ? P.lsmr_maxiter : (int)(m < n ? m : n);
  double *u = c.w.U, *v = c.w.V, *h = c.w.H, *hbar = c.w.HB, *x = c.w.GN;
  KoAcc s0;
  for (long long i = KO_TID; i < m; i += KO_NT) { const double t = b[i]; u[i] = t; s0.add(i, t * t); }
  const double normb = std::sqrt(ko_total(c, s0));
  double beta = normb, alpha = 0, su = 0, sv = 0;
  for (long long i = KO_TID; i < n; i += KO_NT) { x[i] = 0; hbar[i] = 0; v[i] = 0; }
  KO_SYNC();
  if (beta > 0) {
    su = 1 / beta;
    double a2 = 0;
    { const long long t0_ = KO_CLOCK(); kin_jtu(c, u, v, su, 0.0, &a2); c.t_jtu += KO_CLOCK() - t0_; }
    alpha = std::sqrt(a2);
  }
  if (alpha > 0) sv = 1 / alpha;
  for (long long i = KO_TID; i < n; i += KO_NT) h[i] = sv * v[i];
  KO_SYNC();
  int itn = 0, istop = 0;
  double zetabar = alpha * beta, alphabar = alpha, rho = 1, rhobar = 1, cbar = 1, sbar = 0;
  double betadd = beta, betad = 0, rhodold = 1, tautildeold = 0, thetatilde = 0, zeta = 0, d = 0;
  double normA2 = alpha * alpha, maxrbar = 0, minrbar = 1e100;
  const double ctol = P.conlim > 0 ? 1 / P.conlim : 0;
  if (alpha * beta == 0 || normb == 0) { *istop_out = 0; return 0; }
  while (itn < maxiter) {
    ++itn;
    double b2 = 0;
    { const long long t0_ = KO_CLOCK(); kin_jv(c, v, u, sv, -alpha * su, &b2); c.t_jv += KO_CLOCK() - t0_; }
    beta = std::sqrt(b2);
    if (beta > 0) {
      su = 1 / beta;
      double a2 = 0;
      { const long long t0_ = KO_CLOCK(); kin_jtu(c, u, v, su, -beta * sv, &a2); c.t_jtu += KO_CLOCK() - t0_; }
      alpha = std::sqrt(a2);
      if (alpha > 0) sv = 1 / alpha;
    }
    double chat, shat, alphahat, cc, s, ctildeold, stildeold, rhotildeold;
    sym_ortho(alphabar, damp, chat, shat, alphahat);
    const double rhoold = rho;
    sym_ortho(alphahat, beta, cc, s, rho);
    const double thetanew = s * alpha;
    alphabar = cc * alpha;
    const double rhobarold = rhobar, zetaold = zeta, thetabar = sbar * rho, rhotemp = cbar * rho;
    sym_ortho(cbar * rho, thetanew, cbar, sbar, rhobar);
    zeta = cbar * zetabar;
    zetabar = -sbar * zetabar;
    const double k1 = -(thetabar * rho / (rhoold * rhobarold)), k2 = zeta / (rho * rhobar), k3 = -(thetanew / rho);
    KoAcc sx;
    {   // four elements per thread per pass, all sixteen loads issued before the first use (one element at a time leaves four loads in
        // flight per wavefront: the pass is latency bound)
      const double* __restrict__ vr = v; double* __restrict__ hr = h; double* __restrict__ hbr = hbar; double* __restrict__ xr = x;
      long long i = KO_TID;
      for (; i + 3LL * KO_NT < n; i += 4LL * KO_NT) {
        double a[4], b[4], cx[4], dv[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) { const long long k = i + (long long)q * KO_NT; a[q] = hbr[k]; b[q] = hr[k]; cx[q] = xr[k]; dv[q] = vr[k]; }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const long long k = i + (long long)q * KO_NT;
          const double hb = a[q] * k1 + b[q];
          hbr[k] = hb;
          const double xi = cx[q] + k2 * hb;
          xr[k] = xi; sx.add(k, xi * xi);
          hr[k] = b[q] * k3 + sv * dv[q];
        }
      }
      for (; i < n; i += KO_NT) {
        const double hb = hbr[i] * k1 + hr[i];
        hbr[i] = hb;
        const double xi = xr[i] + k2 * hb;
        xr[i] = xi; sx.add(i, xi * xi);
        hr[i] = hr[i] * k3 + sv * vr[i];
      }
    }
    const double normx = std::sqrt(ko_total(c, sx));
    const double betaacute = chat * betadd, betacheck = -shat * betadd;
    const double betahat = cc * betaacute;
    betadd = -s * betaacute;
    const double thetatildeold = thetatilde;
    sym_ortho(rhodold, thetabar, ctildeold, stildeold, rhotildeold);
    thetatilde = stildeold * rhobar;
    rhodold = ctildeold * rhobar;
    betad = -stildeold * betad + ctildeold * betahat;
    tautildeold = (zetaold - thetatildeold * tautildeold) / rhotildeold;
    const double taud = (zeta - thetatilde * tautildeold) / rhodold;
    d += betacheck * betacheck;
    const double normr = std::sqrt(d + (betad - taud) * (betad - taud) + betadd * betadd);
    normA2 += beta * beta;
    const double normA = std::sqrt(normA2);
    normA2 += alpha * alpha;
    maxrbar = maxrbar > rhobarold ? maxrbar : rhobarold;
    if (itn > 1) minrbar = minrbar < rhobarold ? minrbar : rhobarold;
    const double condA = (maxrbar > rhotemp ? maxrbar : rhotemp) / (minrbar < rhotemp ? minrbar : rhotemp);
    const double normar = std::fabs(zetabar);
    const double test1 = normr / normb;
    const double test2 = normA * normr != 0 ? normar / (normA * normr) : INFINITY;
    const double test3 = 1 / condA;
    const double t1 = test1 / (1 + normA * normx / normb);
    const double rtol = P.btol + P.atol * normA * normx / normb;
    if (itn >= maxiter) istop = 7;
    if (1 + test3 <= 1) istop = 6;
    if (1 + test2 <= 1) istop = 5;
    if (1 + t1 <= 1) istop = 4;
    if (test3 <= ctol) istop = 3;
    if (test2 <= P.atol) istop = 2;
    if (test1 <= rtol) istop = 1;
    if (istop > 0) break;
  }
  KO_SYNC();
  *istop_out = istop;
  return itn;
}

// ---- 2-D trust-region problem (scipy solve_trust_region_2d) -------------------------------------------------------------------------
KO_DEV void tr2d(double B00, double B01, double B11, double g0, double g1, double Delta, double& p0, double& p1) {
  if (B00 > 0) {                              // Cholesky succeeds <=> both leading minors positive
    const double l00 = std::sqrt(B00), l10 = B01 / l00, d11 = B11 - l10 * l10;
    if (d11 > 0) {
      const double l11 = std::sqrt(d11);
      const double z0 = -g0 / l00, z1 = (-g1 - l10 * z0) / l11;
      const double q1 = z1 / l11, q0 = (z0 - l10 * q1) / l00;
      if (q0 * q0 + q1 * q1 <= Delta * Delta) { p0 = q0; p1 = q1; return; }
    }
  }
  // boundary p = Delta (sin phi, cos phi): the global minimiser of a degree-2 trigonometric polynomial
  const double a = B00 * Delta * Delta, b = B01 * Delta * Delta, cq = B11 * Delta * Delta, dl = g0 * Delta, fl = g1 * Delta;
  auto val = [&](double ph) { const double s = std::sin(ph), co = std::cos(ph); return 0.5 * (a * s * s + 2 * b * s * co + cq * co * co) + dl * s + fl * co; };
  auto der = [&](double ph) { const double s = std::sin(ph), co = std::cos(ph); return (a - cq) * s * co + b * (co * co - s * s) + dl * co - fl * s; };
  const int K = 360;
  const double step = 6.283185307179586476925 / K;
  double best = 0, bestv = val(0.0);
  for (int i = 0; i < K; ++i) {
    const double lo = i * step, hi = (i + 1) * step;
    double dlo = der(lo), dhi = der(hi);
    if (dlo <= 0 && dhi >= 0) {               // a minimum inside [lo, hi]
      double x0 = lo, x1 = hi;
      for (int it = 0; it < 60; ++it) { const double mid = 0.5 * (x0 + x1); if (der(mid) <= 0) x0 = mid; else x1 = mid; }
      const double ph = 0.5 * (x0 + x1), v = val(ph);
      if (v < bestv) { bestv = v; best = ph; }
    }
    const double v = val(lo);
    if (v < bestv) { bestv = v; best = lo; }
  }
  p0 = Delta * std::sin(best); p1 = Delta * std::cos(best);
}

// ---- the solve: scipy trf_no_bounds (x_scale = 1, linear loss, tr_solver = 'lsmr', regularize = True) ---------------------------------
// stats: cost, nfev, njev, status, total LSMR iterations, last gradient infinity norm
KO_DEV void kin_solve(KinCtx& c, double* xio, double* stats) {
  const long long n = c.q->n, m = c.q->m;
  const KinParams& P = *c.P;
  KinWork& w = c.w;
  c.t_jv = c.t_jtu = 0;
  const long long t_begin = KO_CLOCK();
  for (long long i = KO_TID; i < n; i += KO_NT) w.X[i] = xio[i];
  KO_SYNC();
  kin_residual(c, w.X, w.PN, w.RGN, w.Fv);
  int nfev = 1, njev = 1, status = -1;
  long long lsmr_total = 0;
  kin_linearise(c, w.X);
  double cost = 0.5 * ko_dot(c, w.Fv, w.Fv, m);
  kin_jtu(c, w.Fv, w.G);
  double Delta = std::sqrt(ko_dot(c, w.X, w.X, n));
  if (Delta == 0) Delta = 1.0;
  double g_norm = 0;
  while (true) {
    double gi = 0;
    for (long long i = KO_TID; i < n; i += KO_NT) { const double a = std::fabs(w.G[i]); gi = a > gi ? a : gi; }
#ifndef CHD_HOST_EMU
    {                                          // max over the workgroup
      for (int o = 32; o > 0; o >>= 1) { const double t = __shfl_xor(gi, o); gi = t > gi ? t : gi; }
      __syncthreads();
      if ((threadIdx.x & 63) == 0) c.red[threadIdx.x >> 6] = gi;
      __syncthreads();
      gi = 0;
      for (int i = 0; i < (int)(blockDim.x >> 6); ++i) gi = c.red[i] > gi ? c.red[i] : gi;
    }
#endif
    g_norm = gi;
    if (g_norm < P.gtol) status = 1;
    if (status != -1 || nfev == P.max_nfev) break;
    // regularisation from the Cauchy step (build_quadratic_1d / minimize_quadratic_1d)
    for (long long i = KO_TID; i < n; i += KO_NT) w.S0[i] = -w.G[i];
    KO_SYNC();
    kin_jv(c, w.S0, w.T1);
    double a = 0.5 * ko_dot(c, w.T1, w.T1, m);
    const double gg = ko_dot(c, w.G, w.G, n);
    const double b = -gg, to_tr = Delta / std::sqrt(gg);
    double ag = 0.0;                            // t = 0
    { const double y1 = to_tr * (a * to_tr + b); if (y1 < ag) ag = y1; }
    if (a != 0) { const double ext = -0.5 * b / a; if (0 < ext && ext < to_tr) { const double y2 = ext * (a * ext + b); if (y2 < ag) ag = y2; } }
    const double reg_term = -ag / (Delta * Delta);
    int istop = 0;
    { const int it_ = kin_lsmr(c, w.Fv, std::sqrt(reg_term), &istop); lsmr_total += it_; KO_TRACE("iter cost %.10g Delta %.10g gnorm %.6g damp %.10g lsmr %d istop %d\n", cost, Delta, g_norm, std::sqrt(reg_term), it_, istop); }
    // orthonormal basis of span{g, gn}
    const double ng = std::sqrt(gg);
    for (long long i = KO_TID; i < n; i += KO_NT) w.S0[i] = w.G[i] / ng;
    KO_SYNC();
    double pr = ko_dot(c, w.S0, w.GN, n);
    for (long long i = KO_TID; i < n; i += KO_NT) w.S1[i] = w.GN[i] - pr * w.S0[i];
    KO_SYNC();
    pr = ko_dot(c, w.S0, w.S1, n);            // second pass
    KoAcc s1a;
    for (long long i = KO_TID; i < n; i += KO_NT) { const double t = w.S1[i] - pr * w.S0[i]; w.S1[i] = t; s1a.add(i, t * t); }
    const double s1 = std::sqrt(ko_total(c, s1a));
    const double is1 = s1 > 0 ? 1 / s1 : 0.0;
    for (long long i = KO_TID; i < n; i += KO_NT) w.S1[i] *= is1;
    KO_SYNC();
    kin_jv(c, w.S0, w.T1);
    kin_jv(c, w.S1, w.T2);
    double B00, B01, B11, gS0, gS1, dummy;
    {
      KoAcc a00, a01, a11;
      for (long long i = KO_TID; i < m; i += KO_NT) { a00.add(i, w.T1[i] * w.T1[i]); a01.add(i, w.T1[i] * w.T2[i]); a11.add(i, w.T2[i] * w.T2[i]); }
      ko_total3(c, a00, a01, a11, B00, B01, B11);
      KoAcc g0, g1, z;
      for (long long i = KO_TID; i < n; i += KO_NT) { g0.add(i, w.S0[i] * w.G[i]); g1.add(i, w.S1[i] * w.G[i]); }
      ko_total3(c, g0, g1, z, gS0, gS1, dummy);
    }
    double actual = -1, cost_new = cost;
    while (actual <= 0 && nfev < P.max_nfev) {
      double p0, p1;
      tr2d(B00, B01, B11, gS0, gS1, Delta, p0, p1);
      const double predicted = -(0.5 * (B00 * p0 * p0 + 2 * B01 * p0 * p1 + B11 * p1 * p1) + gS0 * p0 + gS1 * p1);
      const double step_norm = std::sqrt(p0 * p0 + p1 * p1);          // |S p| with orthonormal S
      for (long long i = KO_TID; i < n; i += KO_NT) w.XN[i] = w.X[i] + (p0 * w.S0[i] + p1 * w.S1[i]);
      KO_SYNC();
      kin_residual(c, w.XN, w.PN, w.RGN, w.FN);
      ++nfev;
      double cn, bad, xx;
      {
        KoAcc acn, abad, axx;
        for (long long i = KO_TID; i < m; i += KO_NT) { const double t = w.FN[i]; acn.add(i, t * t); if (!(std::fabs(t) <= 1.79e308)) abad.add(i, 1.0); }
        for (long long i = KO_TID; i < n; i += KO_NT) axx.add(i, w.X[i] * w.X[i]);
        ko_total3(c, acn, abad, axx, cn, bad, xx);
      }
      if (bad > 0) { Delta = 0.25 * step_norm; continue; }
      cost_new = 0.5 * cn;
      actual = cost - cost_new;
      KO_TRACE("   try p (%.10g, %.10g) pred %.10g actual %.10g B (%.8g %.8g %.8g) gS (%.8g %.8g)\n", p0, p1, predicted, actual, B00, B01, B11, gS0, gS1);
      double ratio;
      if (predicted > 0) ratio = actual / predicted;
      else if (predicted == 0 && actual == 0) ratio = 1;
      else ratio = 0;
      double Delta_new = Delta;
      if (ratio < 0.25) Delta_new = 0.25 * step_norm;
      else if (ratio > 0.75 && step_norm > 0.95 * Delta) Delta_new = 2.0 * Delta;
      const bool ftol_ok = actual < P.ftol * cost && ratio > 0.25;
      const bool xtol_ok = step_norm < P.xtol * (P.xtol + std::sqrt(xx));
      if (ftol_ok && xtol_ok) status = 4; else if (ftol_ok) status = 2; else if (xtol_ok) status = 3;
      if (status != -1) break;
      Delta = Delta_new;
    }
    if (actual > 0) {
      for (long long i = KO_TID; i < n; i += KO_NT) w.X[i] = w.XN[i];
      for (long long i = KO_TID; i < m; i += KO_NT) w.Fv[i] = w.FN[i];
      cost = cost_new;
      KO_SYNC();
      kin_linearise(c, w.X);
      ++njev;
      kin_jtu(c, w.Fv, w.G);
    }
  }
  for (long long i = KO_TID; i < n; i += KO_NT) xio[i] = w.X[i];
  if (KO_TID == 0) {
    stats[0] = cost; stats[1] = nfev; stats[2] = njev; stats[3] = status == -1 ? 0 : status; stats[4] = (double)lsmr_total; stats[5] = g_norm;
    const double tall = (double)(KO_CLOCK() - t_begin);
    stats[6] = tall > 0 ? (double)c.t_jv / tall : 0.0; stats[7] = tall > 0 ? (double)c.t_jtu / tall : 0.0;
  }
  KO_SYNC();
}

KO_DEV void kin_bind(KinCtx& c, const KinSeq* q, const KinParams* P, const double* dpool, const int* ipool, double* work, double* red, double* lds, int lds_doubles) {
  c.q = q; c.P = P;
  const int F = q->F;
  const double* d = dpool + q->o_const;
  c.offs = d; d += 84; c.pose3d = d; d += 84LL * F; c.root_trans = d; d += 3LL * F; c.pose2d = d; d += 56LL * F; c.proj_w = d; d += 28LL * F; c.data_w = d;
  c.contact = ipool + q->o_contact;
  c.w.carve(work + q->o_work, F);
  c.red = red; c.lds = lds; c.lds_doubles = lds_doubles;
  c.o2 = 56 * F; c.o3 = c.o2 + 84 * (F - 1); c.o4 = c.o3 + 84 * (F - 2); c.o5 = c.o4 + 84 * F; c.o6 = c.o5 + 84 * (F - 1); c.o7 = c.o6 + 28 * F;
}

}  // namespace chd_kin

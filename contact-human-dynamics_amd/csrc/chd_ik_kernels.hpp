// chd_ik_kernels.hpp -- one damped-least-squares step of the IK back-projection for one frame of one video.
//
// Reference: JacobianInverseKinematicsCK.__call__ / .jacobian (src/skeleton_fitting/ik/InverseKinematics.py:411-561) as
// called by apply_results (src/utils/towr_utils.py:843).  The same source is compiled by hipcc for gfx950 (one workgroup
// per frame, IK_NT threads) and by g++ with -DCHD_HOST_EMU (one emulated thread) for the CPU tests.
//
// Differences from the reference, both exact in real arithmetic:
//  * the step is solved in its dual form dx = J^T (J J^T + lambda^2 I)^-1 e: apply_results uses unit weights, so lambda is
//    the same for all 6 J unknowns and the 6J x 6J LU per frame becomes a 3T x 3T Cholesky (T targets, ~13);
//  * the rotation axes of the Jacobian are formed with rotation matrices instead of quaternion products;
//  * the Jacobian is never stored: an entry is one cross product of an axis with a (target - joint) offset, both already
//    in LDS, so J J^T is accumulated per pair of targets over their common ancestors and J^T y per unknown.
// A frame only needs its neighbours' previous iterate (smoothness term), so every iteration is one launch over all frames
// of all videos with the state double-buffered in HBM.  Per frame the kernel touches 3 x 7J doubles of state in, 7J out
// and 3T target coordinates; everything else lives in LDS (51 J + 6 T + (3T + 1)(3T + 2) / 2 doubles + index tables: 21 KB for J = 33, T = 13; 39 KB for J = 28, T = 25).
#pragma once
#include <cmath>

#ifdef CHD_HOST_EMU
#define IK_DEV static inline
#define IK_HD inline
#define IK_TID 0
#define IK_NT 1
#define IK_SYNC() ((void)0)
#else
#include <hip/hip_runtime.h>
#define IK_DEV __device__ inline
#define IK_HD __host__ __device__ inline
#define IK_TID ((int)threadIdx.x)
#define IK_NT ((int)blockDim.x)
#define IK_SYNC() __syncthreads()
#endif
// one-wavefront sections (short dependent chains in LDS while the other wavefront waits at the next barrier)
#ifdef CHD_HOST_EMU
#define IK_WAVE0 true
#define IK_WLANE 0
#define IK_WSTEP 1
#define IK_WSYNC() ((void)0)
#else
#define IK_WAVE0 (threadIdx.x < 64)
#define IK_WLANE ((int)threadIdx.x)
#define IK_WSTEP 64
#define IK_WSYNC() do { __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup"); __builtin_amdgcn_wave_barrier(); } while (0)
#endif
#define IK_FOR(i, n) for (int i = IK_TID; i < (n); i += IK_NT)
// -DIK_PROFILE: shader-clock ticks per phase of a step, summed over all workgroups of a call (first thread's view; a study build)
#if defined(IK_PROFILE) && !defined(CHD_HOST_EMU)
__device__ unsigned long long ik_prof[16];
#define IK_SEG(k) do { if (threadIdx.x == 0 && (blockIdx.x & 63) == 0) { const long long t_ = (long long)clock64(); atomicAdd(&ik_prof[k], (unsigned long long)(t_ - ik_t0)); ik_t0 = t_; } } while (0)      /* (one workgroup in 64: every workgroup adding to the same twelve words serialises them) */
#define IK_SEG_BEGIN() long long ik_t0 = (long long)clock64()
#else
#define IK_SEG(k) ((void)0)
#define IK_SEG_BEGIN() ((void)0)
#endif

namespace chd_ik {

enum { MAXJ = 64, MAXT = 26, MAXR = 3 * MAXT };

struct IkParams { int iterations, translate; double damping, smoothness, gamma; };

// one video inside the batch pools
struct IkSeq {
  int F, J, T;
  int o_parents, o_tj, o_masks;     // int pool: parents[J], target_joints[T], then the ancestor relation as bit masks:
                                    //   jt_strict[J], jt_self[J]: bit t set when joint j is a strict ancestor of / an ancestor of or equal to target t's joint;
                                    //   ta_strict[T][2], ta_self[T][2]: the same relation per target as 64-bit joint masks (low word, high word)
  long long o_targets;              // double pool: T x F x 3
  long long o_state;                // state buffers: F x (7 J): per frame J quaternions (w x y z) then J translations
};

// per-frame workgroup scratch: views into one block of doubles (LDS on the device), carved for the largest J and T of the batch
struct IkLds {
  double* x;                        // 6J: Euler angles (3J) then translations (3J)
  double *Rl, *Rg;                  // 9J each: local / global rotation matrices, row-major
  double* pg;                       // 3J: global positions
  double* es;                       // 18J: axes of the 3J rotation unknowns, then of the 3J translation unknowns
  double* dx;                       // 6J: J^T y
  double *e, *y;                    // 3T each
  double* G;                        // the lower triangle of the (3T + 1) x (3T + 1) matrix, row by row (entry (r, c), c <= r, at r (r + 1) / 2 + c); row 3T: residual.
                                    // Packed since round 5: half the LDS of the square it was, so that four frames instead of two fit a compute unit -- every phase of a
                                    // step is a short dependent chain (profiles/r05_experiments.md section 9), more frames in flight is what raises the throughput
  int* itab;                        // the sequence's index tables (3J + 5T ints: parents | target joints | masks, as in the pool),
                                    // staged once per step: the chain walks and mask tests would otherwise be dependent L2 round trips
  const unsigned short* pair;       // (row, column) offsets of the idx-th entry of a lower triangle, (i << 8) | j: one table for every size and column, built by the host
                                    // (device memory, 6 KB: it stays in the compute unit's cache; it used to be rebuilt in LDS every step)
  double* col;                      // 4 x (3T + 2): the two columns being eliminated together, as they stood before the pass (two generations: one barrier per pass)
  static IK_HD int tri(int r) { return r * (r + 1) / 2; }
  static IK_HD int doubles(int J, int T) { return 51 * J + 6 * T + tri(3 * T + 1) + 4 * (3 * T + 2) + (3 * J + 5 * T + 1) / 2; }
  static IK_HD int pair_entries() { return MAXR * (MAXR + 1) / 2; }
  static inline void fill_pairs(unsigned short* t) {      // host: idx -> (i, j) of the triangle enumerated row by row
    int idx = 0;
    for (int i = 0; i < MAXR; ++i) for (int j = 0; j <= i; ++j) t[idx++] = (unsigned short)((i << 8) | j);
  }
  IK_HD void carve(double* b, int J, int T) {
    x = b; b += 6 * J; Rl = b; b += 9 * J; Rg = b; b += 9 * J; pg = b; b += 3 * J; es = b; b += 18 * J; dx = b; b += 6 * J;
    e = b; b += 3 * T; y = b; b += 3 * T; G = b; b += tri(3 * T + 1); col = b; b += 4 * (3 * T + 2);
    itab = reinterpret_cast<int*>(b);
  }
};

IK_DEV void quat_to_mat(const double* q, double* m) {          // Quaternions.transforms (Quaternions.py:301-324)
  const double qw = q[0], qx = q[1], qy = q[2], qz = q[3];
  const double x2 = qx + qx, y2 = qy + qy, z2 = qz + qz;
  const double xx = qx * x2, yy = qy * y2, wx = qw * x2, xy = qx * y2, yz = qy * z2, wy = qw * y2, xz = qx * z2, zz = qz * z2, wz = qw * z2;
  m[0] = 1.0 - (yy + zz); m[1] = xy - wz; m[2] = xz + wy;
  m[3] = xy + wz; m[4] = 1.0 - (xx + zz); m[5] = yz - wx;
  m[6] = xz - wy; m[7] = yz + wx; m[8] = 1.0 - (xx + yy);
}
IK_DEV void quat_to_euler(const double* q, double* es) {      // Quaternions.euler('xyz') on the normalised quaternion (Quaternions.py:215-227)
  const double n = std::sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
  const double q0 = q[0] / n, q1 = q[1] / n, q2 = q[2] / n, q3 = q[3] / n;
  es[0] = std::atan2(2 * (q0 * q1 + q2 * q3), 1 - 2 * (q1 * q1 + q2 * q2));
  double s = 2 * (q0 * q2 - q3 * q1);
  s = s < -1.0 ? -1.0 : (s > 1.0 ? 1.0 : s);
  es[1] = std::asin(s);
  es[2] = std::atan2(2 * (q0 * q3 + q1 * q2), 1 - 2 * (q2 * q2 + q3 * q3));
}
IK_DEV void quat_mul(const double* q, const double* r, double* o) {      // Quaternions.__mul__ (Quaternions.py:91-105)
  o[0] = r[0] * q[0] - r[1] * q[1] - r[2] * q[2] - r[3] * q[3];
  o[1] = r[0] * q[1] + r[1] * q[0] - r[2] * q[3] + r[3] * q[2];
  o[2] = r[0] * q[2] + r[1] * q[3] + r[2] * q[0] - r[3] * q[1];
  o[3] = r[0] * q[3] - r[1] * q[2] + r[2] * q[1] + r[3] * q[0];
}
IK_DEV void euler_to_quat(const double* es, double* q) {      // Quaternions.from_euler(order='xyz', world=True): q_z (q_y q_x), Quaternions.py:401-420
  const double k = 1.0 / (1.0 + 1e-10);                       // from_angle_axis divides the unit axis by (1 + 1e-10)
  const double qx[4] = {std::cos(es[0] / 2), std::sin(es[0] / 2) * k, 0, 0};
  const double qy[4] = {std::cos(es[1] / 2), 0, std::sin(es[1] / 2) * k, 0};
  const double qz[4] = {std::cos(es[2] / 2), 0, 0, std::sin(es[2] / 2) * k};
  double t[4];
  quat_mul(qy, qx, t);
  quat_mul(qz, t, q);
}
IK_DEV void mat_mul(const double* a, const double* b, double* o) {
  for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) o[3 * i + j] = a[3 * i] * b[j] + a[3 * i + 1] * b[3 + j] + a[3 * i + 2] * b[6 + j];
}
IK_DEV void mat_vec(const double* a, const double* v, double* o) {
  for (int i = 0; i < 3; ++i) o[i] = a[3 * i] * v[0] + a[3 * i + 1] * v[1] + a[3 * i + 2] * v[2];
}

IK_DEV void cross3(const double* u, const double* v, double* o) {
  o[0] = u[1] * v[2] - u[2] * v[1]; o[1] = u[2] * v[0] - u[0] * v[2]; o[2] = u[0] * v[1] - u[1] * v[0];
}

// One step for frame f of sequence s: reads the state `Xin` (all frames), writes frame f of `Xout`.
IK_DEV void ik_step_frame(const IkSeq& s, const int f, const IkParams& P, const int* ipool, const double* dpool,
                          const double* Xin, double* Xout, const IkLds& L) {
  const int J = s.J, T = s.T, F = s.F;
  const int nvar = P.translate ? 6 * J : 3 * J, R = 3 * T;
  IK_SEG_BEGIN();
  const double* xin = Xin + s.o_state + (long long)f * 7 * J;
  IK_FOR(k, 3 * J + 5 * T) L.itab[k] = ipool[s.o_parents + k];      // parents | tj | masks are contiguous in the pool (IkBatch::build)
  const int* parents = L.itab; const int* tj = parents + J;
  const int* jt_strict = tj + T; const int* jt_self = jt_strict + J; const int* ta_strict = jt_self + J; const int* ta_self = ta_strict + 2 * T;
  // ---- A: unknowns of this frame and the local rotation matrices (Animation.transforms_local)
  IK_FOR(j, J) {
    quat_to_euler(xin + 4 * j, L.x + 3 * j);
    for (int a = 0; a < 3; ++a) L.x[3 * J + 3 * j + a] = xin[4 * J + 3 * j + a];
    quat_to_mat(xin + 4 * j, L.Rl + 9 * j);
  }
  IK_SYNC();
  IK_SEG(0);
  // ---- B: global transforms (Animation.transforms_global): every joint walks up its ancestor chain
  IK_FOR(j, J) {
    double Rm[9], p[3];
    for (int k = 0; k < 9; ++k) Rm[k] = L.Rl[9 * j + k];
    for (int a = 0; a < 3; ++a) p[a] = L.x[3 * J + 3 * j + a];
    for (int a = parents[j]; a >= 0; a = parents[a]) {
      double Rn[9], pn[3];
      mat_mul(L.Rl + 9 * a, Rm, Rn); mat_vec(L.Rl + 9 * a, p, pn);
      for (int k = 0; k < 9; ++k) Rm[k] = Rn[k];
      for (int k = 0; k < 3; ++k) p[k] = pn[k] + L.x[3 * J + 3 * a + k];
    }
    for (int k = 0; k < 9; ++k) L.Rg[9 * j + k] = Rm[k];
    for (int k = 0; k < 3; ++k) L.pg[3 * j + k] = p[k];
  }
  IK_SYNC();
  IK_SEG(1);
  // ---- C: axes of the unknowns (jacobian(), InverseKinematics.py:414-426, 438-443): parent rotation x partial Euler rotations
  IK_FOR(j, J) {
    const double I9[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    const double* Pr = j == 0 ? I9 : L.Rg + 9 * parents[j];          // prs[:, 0] = identity
    const double cy = std::cos(L.x[3 * j + 1]), sy = std::sin(L.x[3 * j + 1]), cz = std::cos(L.x[3 * j + 2]), sz = std::sin(L.x[3 * j + 2]);
    const double ax0[3] = {cz * cy, sz * cy, -sy};              // Rz(z) Ry(y) e_x
    const double ax1[3] = {-sz, cz, 0.0};                       // Rz(z) e_y
    const double ax2[3] = {0.0, 0.0, 1.0};
    mat_vec(Pr, ax0, L.es + 9 * j); mat_vec(Pr, ax1, L.es + 9 * j + 3); mat_vec(Pr, ax2, L.es + 9 * j + 6);
    for (int a = 0; a < 3; ++a) for (int k = 0; k < 3; ++k) L.es[9 * J + 9 * j + 3 * a + k] = Pr[3 * k + a];      // Pr e_a
  }
  // ---- D: residual, stored as row R of G (see F)
  IK_FOR(r, R) { const int t = r / 3, a = r % 3; L.G[IkLds::tri(R) + r] = P.gamma * (dpool[s.o_targets + ((long long)t * F + f) * 3 + a] - L.pg[3 * tj[t] + a]); }
  IK_SYNC();
  IK_SEG(2);
  // ---- E: G = J J^T + lambda^2 I  (dual form of jf.T.dot(jf) + d, InverseKinematics.py:497-502; w = 1 => l = damping / 1.001).
  //         Jacobian entries (InverseKinematics.py:428-447): rows of target t, column of rotation unknown (j, a):
  //         es[3j+a] x (p_target - p_j) if j is a strict ancestor of the target; column of translation unknown (j, a): es[3J+3j+a]
  //         if j is an ancestor or the target itself.  One 3 x 3 block of G per pair of targets (lower triangle of blocks).
  const double lam = P.damping * (1.0 / (1.0 + 0.001));
  IK_FOR(idx, T * (T + 1) / 2) {                                // (t1, t2 <= t1): one pass of the workgroup for T <= 15
    int t1 = (int)((std::sqrt(8.0f * (float)idx + 1.0f) - 1.0f) * 0.5f);
    while (t1 * (t1 + 1) / 2 > idx) --t1;
    while ((t1 + 1) * (t1 + 2) / 2 <= idx) ++t1;
    const int t2 = idx - t1 * (t1 + 1) / 2;
    const double* p1 = L.pg + 3 * tj[t1]; const double* p2 = L.pg + 3 * tj[t2];
    double g[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    // joints above both targets, from the per-target joint masks
    unsigned long long cs = ((unsigned long long)(unsigned)ta_strict[2 * t1] | ((unsigned long long)(unsigned)ta_strict[2 * t1 + 1] << 32)) &
                            ((unsigned long long)(unsigned)ta_strict[2 * t2] | ((unsigned long long)(unsigned)ta_strict[2 * t2 + 1] << 32));
    unsigned long long co = ((unsigned long long)(unsigned)ta_self[2 * t1] | ((unsigned long long)(unsigned)ta_self[2 * t1 + 1] << 32)) &
                            ((unsigned long long)(unsigned)ta_self[2 * t2] | ((unsigned long long)(unsigned)ta_self[2 * t2 + 1] << 32));
    if (!P.translate) co = 0;
    for (unsigned long long rest = cs | co; rest; rest &= rest - 1) {
      const int j = __builtin_ctzll(rest);
      const int m = (int)((cs >> j) & 1) | ((int)((co >> j) & 1) << 1);
      if (m & 1) {
        const double d1[3] = {p1[0] - L.pg[3 * j], p1[1] - L.pg[3 * j + 1], p1[2] - L.pg[3 * j + 2]};
        const double d2[3] = {p2[0] - L.pg[3 * j], p2[1] - L.pg[3 * j + 1], p2[2] - L.pg[3 * j + 2]};
        for (int a = 0; a < 3; ++a) {
          double c1[3], c2[3];
          cross3(L.es + 9 * j + 3 * a, d1, c1); cross3(L.es + 9 * j + 3 * a, d2, c2);
          for (int i = 0; i < 3; ++i) for (int k = 0; k < 3; ++k) g[3 * i + k] += c1[i] * c2[k];
        }
      }
      if (m & 2)
        for (int a = 0; a < 3; ++a) {
          const double* u = L.es + 9 * J + 9 * j + 3 * a;
          for (int i = 0; i < 3; ++i) for (int k = 0; k < 3; ++k) g[3 * i + k] += u[i] * u[k];
        }
    }
    for (int i = 0; i < 3; ++i) for (int k = 0; k < 3; ++k)
      if (3 * t2 + k <= 3 * t1 + i) L.G[IkLds::tri(3 * t1 + i) + 3 * t2 + k] = g[3 * i + k] + ((t1 == t2 && i == k) ? lam * lam : 0.0);      // (lower triangle only)
  }
  IK_SYNC();
  IK_SEG(3);
  // ---- F: (G + lambda^2 I) y = e.  G = L D L^T without pivoting or square roots (G is SPD), right-looking with unscaled
  //         columns U[r][k] = L[r][k] d_k so that a column needs ONE workgroup barrier; the residual rides along as row R
  //         (its eliminated entries are U[R][k] = (D^-1 L^-1 e)_k d_k, i.e. the forward substitution comes for free).
  // Two columns per pass (round 5): the chain of R dependent column steps (barrier, pivot reciprocal, column reads, update) is what bounds the elimination
  // (profiles/r05_experiments.md section 9), so a pass eliminates columns k and k + 1 together.  Every thread recomputes, for its entry's row and column, what column
  // k + 1 looks like after step k (a' = G[r][k+1] - G[r][k] G[k+1][k] / d_k ...) and applies both updates -- the same products subtracted in the same order as column
  // by column, bit for bit the same factor, with half the barriers.  The two columns are read from a copy taken before the pass; the threads that finish the NEXT two
  // columns' entries leave their copy for the next pass.
  const int cw = R + 2;
  IK_FOR(r, R + 1) { L.col[r] = L.G[IkLds::tri(r)]; if (R > 1) L.col[cw + r] = r >= 1 ? L.G[IkLds::tri(r) + 1] : 0.0; }
  IK_SYNC();
  IK_SEG(5);
  int k = 0;
  for (; k + 1 < R; k += 2) {
    const double* c0 = L.col + ((k >> 1) & 1) * 2 * cw; const double* c1 = c0 + cw;      // columns k and k + 1 before this pass
    double* n0 = L.col + (((k >> 1) + 1) & 1) * 2 * cw; double* n1 = n0 + cw;              // columns k + 2 and k + 3 after it
    const double inv0 = 1.0 / c0[k];
    const double m = c0[k + 1];                                    // G[k+1][k]
    const double inv1 = 1.0 / (c1[k + 1] - m * m * inv0);          // 1 / d_{k+1}
    const int n = R - k;                                          // rows k+1 .. R, columns k+1 .. R-1, lower triangle
    for (int idx = IK_TID; idx < n * (n + 1) / 2 - 1; idx += IK_NT) {      // the last entry would be (R, R): not needed
      const unsigned pr = L.pair[idx];
      const int r = k + 1 + (int)(pr >> 8), cc = k + 1 + (int)(pr & 255u);
      double g = L.G[IkLds::tri(r) + cc] - c0[r] * c0[cc] * inv0;        // step k
      if (cc > k + 1) {                                           // step k + 1, with column k + 1 as step k left it
        const double a = c1[r] - c0[r] * m * inv0, b = c1[cc] - c0[cc] * m * inv0;
        g -= a * b * inv1;
      }
      L.G[IkLds::tri(r) + cc] = g;
      if (cc == k + 2) n0[r] = g; else if (cc == k + 3) n1[r] = g;
    }
    IK_SYNC();
    IK_SEG(6);
  }
  for (; k < R; ++k) {                                            // an odd column at the end
    const double inv = 1.0 / L.G[IkLds::tri(k) + k];
    const int n = R - k;
    for (int idx = IK_TID; idx < n * (n + 1) / 2 - 1; idx += IK_NT) {
      const unsigned pr = L.pair[idx];
      const int r = k + 1 + (int)(pr >> 8), cc = k + 1 + (int)(pr & 255u);
      L.G[IkLds::tri(r) + cc] -= L.G[IkLds::tri(r) + k] * L.G[IkLds::tri(cc) + k] * inv;
    }
    IK_SYNC();
    IK_SEG(7);
  }
  IK_FOR(k, R) { L.e[k] = L.G[IkLds::tri(R) + k]; L.G[IkLds::tri(k) + k] = 1.0 / L.G[IkLds::tri(k) + k]; }      // right-hand side of L^T y = D^-1 (.), and 1 / d_k
  IK_SYNC();
  IK_SEG(8);
  // back substitution y_k = (U[R][k] - sum_{r > k} U[r][k] y_r) / d_k by one wavefront, scatter form: once y_k is known
  // every lane r < k takes U[k][r] y_k off its own accumulator e[r] (row k of G is contiguous).  Two unknowns per wavefront
  // synchronisation: y_{k-1} only needs y_k on top of what the lanes hold (same operations in the same order as one at a time).
  if (IK_WAVE0) {
    for (int k = R - 1; k >= 0; k -= 2) {
      const double yk = L.e[k] * L.G[IkLds::tri(k) + k];
      const bool two = k >= 1;
      const double yk1 = two ? (L.e[k - 1] - L.G[IkLds::tri(k) + k - 1] * yk) * L.G[IkLds::tri(k - 1) + k - 1] : 0.0;
      const int lim = two ? k - 1 : 0;
      for (int r = IK_WLANE; r < lim; r += IK_WSTEP) L.e[r] = (L.e[r] - L.G[IkLds::tri(k) + r] * yk) - L.G[IkLds::tri(k - 1) + r] * yk1;
      if (IK_WLANE == 0) { L.y[k] = yk; if (two) L.y[k - 1] = yk1; }
      IK_WSYNC();
    }
  }
  IK_SYNC();
  IK_SEG(9);
  // ---- G: dx = J^T y, one unknown per thread: dx[v] = sum over the targets below joint j of (es[v] x d_t) . y_t = es[v] . (d_t x y_t)
  IK_FOR(v, nvar) {
    const bool rot = v < 3 * J;
    const int j = rot ? v / 3 : (v - 3 * J) / 3;
    double acc[3] = {0, 0, 0};                                   // rotation: sum of d_t x y_t; translation: sum of y_t
    for (unsigned rest = (unsigned)(rot ? jt_strict[j] : jt_self[j]); rest; rest &= rest - 1) {
      const int t = __builtin_ctz(rest);
      if (rot) {
        {
          const double d[3] = {L.pg[3 * tj[t]] - L.pg[3 * j], L.pg[3 * tj[t] + 1] - L.pg[3 * j + 1], L.pg[3 * tj[t] + 2] - L.pg[3 * j + 2]};
          double c[3];
          cross3(d, L.y + 3 * t, c);
          acc[0] += c[0]; acc[1] += c[1]; acc[2] += c[2];
        }
      } else { acc[0] += L.y[3 * t]; acc[1] += L.y[3 * t + 1]; acc[2] += L.y[3 * t + 2]; }
    }
    const double* u = L.es + 3 * v;
    L.dx[v] = u[0] * acc[0] + u[1] * acc[1] + u[2] * acc[2];
  }
  IK_SYNC();
  IK_SEG(10);
  // ---- H: smoothness term on the previous iterate of the neighbouring frames (InverseKinematics.py:506-517);
  //         new rotations from the new Euler angles (:540-544)
  double* xout = Xout + s.o_state + (long long)f * 7 * J;
  const double* xpv = Xin + s.o_state + (long long)(f > 0 ? f - 1 : 0) * 7 * J;
  const double* xav = Xin + s.o_state + (long long)(f < F - 1 ? f + 1 : F - 1) * 7 * J;
  IK_FOR(j, J) {
    double ep[3], ea[3], en[3];
    quat_to_euler(xpv + 4 * j, ep); quat_to_euler(xav + 4 * j, ea);
    for (int a = 0; a < 3; ++a) {
      const int v = 3 * j + a;
      en[a] = L.x[v] + L.dx[v] + P.smoothness * (ep[a] + ea[a] - 2 * L.x[v]);
    }
    euler_to_quat(en, xout + 4 * j);
    for (int a = 0; a < 3; ++a) {
      const int v = 3 * J + 3 * j + a;
      double xn = L.x[v];
      if (P.translate) xn += L.dx[v] + P.smoothness * (xpv[4 * J + 3 * j + a] + xav[4 * J + 3 * j + a] - 2 * L.x[v]);
      xout[4 * J + 3 * j + a] = xn;
    }
  }
  IK_SEG(11);
}

}  // namespace chd_ik

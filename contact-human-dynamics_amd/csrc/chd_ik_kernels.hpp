// chd_ik_kernels.hpp -- one damped-least-squares step of the IK back-projection for one frame of one video.
//
// Reference: JacobianInverseKinematicsCK.__call__ / .jacobian (src/skeleton_fitting/ik/InverseKinematics.py:411-561) as
// called by apply_results (src/utils/towr_utils.py:843).  The same source is compiled by hipcc for gfx950 (one workgroup
// per frame, IK_NT threads) and by g++ with -DCHD_HOST_EMU (one emulated thread) for the CPU tests.
//
// Differences from the reference, both exact in real arithmetic:
//  * the step is solved in its dual form dx = J^T (J J^T + lambda^2 I)^-1 e: apply_results uses unit weights, so lambda is
//    the same for all 6 J unknowns and the 6J x 6J LU per frame becomes a 3T x 3T Cholesky (T targets, ~13);
//  * the rotation axes of the Jacobian are formed with rotation matrices instead of quaternion products.
// A frame only needs its neighbours' previous iterate (smoothness term), so every iteration is one launch over all frames
// of all videos with the state double-buffered in HBM.
#pragma once
#include <cmath>

#ifdef CHD_HOST_EMU
#define IK_DEV static inline
#define IK_TID 0
#define IK_NT 1
#define IK_SYNC() ((void)0)
#else
#include <hip/hip_runtime.h>
#define IK_DEV __device__ inline
#define IK_TID ((int)threadIdx.x)
#define IK_NT ((int)blockDim.x)
#define IK_SYNC() __syncthreads()
#endif
#define IK_FOR(i, n) for (int i = IK_TID; i < (n); i += IK_NT)

namespace chd_ik {

enum { MAXJ = 64, MAXT = 21, MAXR = 3 * MAXT };

struct IkParams { int iterations, translate; double damping, smoothness, gamma; };

// one video inside the batch pools
struct IkSeq {
  int F, J, T;
  int o_parents, o_tj, o_desc;      // int pool: parents[J], target_joints[T], desc[J * T] (bit 0: strict descendant, bit 1: or self)
  long long o_targets;              // double pool: T x F x 3
  long long o_state;                // state buffers: F x (7 J): per frame J quaternions (w x y z) then J translations
  long long o_jm;                   // workspace: F x (3T x 6J) Jacobians
};

// per-frame workgroup scratch (LDS on the device)
struct IkLds {
  double x[6 * MAXJ];               // Euler angles (3J) then translations (3J)
  double Rl[MAXJ][9], Rg[MAXJ][9], pg[MAXJ][3];
  double es[6 * MAXJ][3];           // axes of the 3J rotation unknowns, then of the 3J translation unknowns
  double e[MAXR], y[MAXR];
  double G[MAXR][MAXR + 1];
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

// One step for frame f of sequence s: reads the state `Xin` (all frames), writes frame f of `Xout`.
IK_DEV void ik_step_frame(const IkSeq& s, const int f, const IkParams& P, const int* ipool, const double* dpool,
                          const double* Xin, double* Xout, double* jm_pool, IkLds& L) {
  const int J = s.J, T = s.T, F = s.F;
  const int nvar = P.translate ? 6 * J : 3 * J, R = 3 * T;
  const int* parents = ipool + s.o_parents; const int* tj = ipool + s.o_tj; const int* desc = ipool + s.o_desc;
  const double* xin = Xin + s.o_state + (long long)f * 7 * J;
  double* jm = jm_pool + s.o_jm + (long long)f * R * (6 * J);          // row-major R x nvar
  // ---- A: unknowns of this frame and the local rotation matrices (Animation.transforms_local)
  IK_FOR(j, J) {
    quat_to_euler(xin + 4 * j, L.x + 3 * j);
    for (int a = 0; a < 3; ++a) L.x[3 * J + 3 * j + a] = xin[4 * J + 3 * j + a];
    quat_to_mat(xin + 4 * j, L.Rl[j]);
  }
  IK_SYNC();
  // ---- B: global transforms (Animation.transforms_global): every joint walks up its ancestor chain
  IK_FOR(j, J) {
    double Rm[9], p[3];
    for (int k = 0; k < 9; ++k) Rm[k] = L.Rl[j][k];
    for (int a = 0; a < 3; ++a) p[a] = L.x[3 * J + 3 * j + a];
    for (int a = parents[j]; a >= 0; a = parents[a]) {
      double Rn[9], pn[3];
      mat_mul(L.Rl[a], Rm, Rn); mat_vec(L.Rl[a], p, pn);
      for (int k = 0; k < 9; ++k) Rm[k] = Rn[k];
      for (int k = 0; k < 3; ++k) p[k] = pn[k] + L.x[3 * J + 3 * a + k];
    }
    for (int k = 0; k < 9; ++k) L.Rg[j][k] = Rm[k];
    for (int k = 0; k < 3; ++k) L.pg[j][k] = p[k];
  }
  IK_SYNC();
  // ---- C: axes of the unknowns (jacobian(), InverseKinematics.py:414-426, 438-443): parent rotation x partial Euler rotations
  IK_FOR(j, J) {
    double I9[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    const double* Pr = j == 0 ? I9 : L.Rg[parents[j]];          // prs[:, 0] = identity
    const double cy = std::cos(L.x[3 * j + 1]), sy = std::sin(L.x[3 * j + 1]), cz = std::cos(L.x[3 * j + 2]), sz = std::sin(L.x[3 * j + 2]);
    const double ax0[3] = {cz * cy, sz * cy, -sy};              // Rz(z) Ry(y) e_x
    const double ax1[3] = {-sz, cz, 0.0};                       // Rz(z) e_y
    const double ax2[3] = {0.0, 0.0, 1.0};
    mat_vec(Pr, ax0, L.es[3 * j]); mat_vec(Pr, ax1, L.es[3 * j + 1]); mat_vec(Pr, ax2, L.es[3 * j + 2]);
    for (int a = 0; a < 3; ++a) for (int k = 0; k < 3; ++k) L.es[3 * J + 3 * j + a][k] = Pr[3 * k + a];      // Pr e_a
  }
  // ---- D: residual
  IK_FOR(r, R) { const int t = r / 3, a = r % 3; L.e[r] = P.gamma * (dpool[s.o_targets + ((long long)t * F + f) * 3 + a] - L.pg[tj[t]][a]); }
  IK_SYNC();
  // ---- E: Jacobian (InverseKinematics.py:428-447), R x nvar
  IK_FOR(idx, R * nvar) {
    const int r = idx / nvar, v = idx % nvar, t = r / 3, a = r % 3;
    double val = 0.0;
    if (v < 3 * J) {
      const int j = v / 3;
      if (desc[j * T + t] & 1) {
        const double d0 = L.pg[tj[t]][0] - L.pg[j][0], d1 = L.pg[tj[t]][1] - L.pg[j][1], d2 = L.pg[tj[t]][2] - L.pg[j][2];
        const double* ex = L.es[v];
        val = a == 0 ? ex[1] * d2 - ex[2] * d1 : a == 1 ? ex[2] * d0 - ex[0] * d2 : ex[0] * d1 - ex[1] * d0;
      }
    } else {
      const int j = (v - 3 * J) / 3;
      if (desc[j * T + t] & 2) val = L.es[v][a];
    }
    jm[(long long)r * nvar + v] = val;
  }
  IK_SYNC();
  // ---- F: G = J J^T + lambda^2 I  (dual form of jf.T.dot(jf) + d, InverseKinematics.py:497-502; w = 1 => l = damping / 1.001)
  const double lam = P.damping * (1.0 / (1.0 + 0.001));
  IK_FOR(idx, R * R) {
    const int r = idx / R, cc = idx % R;
    if (cc > r) continue;
    double acc = 0.0;
    for (int v = 0; v < nvar; ++v) acc += jm[(long long)r * nvar + v] * jm[(long long)cc * nvar + v];
    L.G[r][cc] = acc + (r == cc ? lam * lam : 0.0);
  }
  IK_SYNC();
  // ---- G: Cholesky G = C C^T (lower), then C z = e, C^T y = z
  for (int k = 0; k < R; ++k) {
    if (IK_TID == 0) L.G[k][k] = std::sqrt(L.G[k][k]);
    IK_SYNC();
    for (int r = k + 1 + IK_TID; r < R; r += IK_NT) L.G[r][k] /= L.G[k][k];
    IK_SYNC();
    for (int idx = IK_TID; idx < (R - k - 1) * (R - k - 1); idx += IK_NT) {
      const int r = k + 1 + idx / (R - k - 1), cc = k + 1 + idx % (R - k - 1);
      if (cc <= r) L.G[r][cc] -= L.G[r][k] * L.G[cc][k];
    }
    IK_SYNC();
  }
  if (IK_TID == 0) {
    for (int r = 0; r < R; ++r) { double v = L.e[r]; for (int k = 0; k < r; ++k) v -= L.G[r][k] * L.y[k]; L.y[r] = v / L.G[r][r]; }
    for (int r = R - 1; r >= 0; --r) { double v = L.y[r]; for (int k = r + 1; k < R; ++k) v -= L.G[k][r] * L.y[k]; L.y[r] = v / L.G[r][r]; }
  }
  IK_SYNC();
  // ---- H: dx1 = J^T y; smoothness term on the previous iterate of the neighbouring frames (InverseKinematics.py:506-517);
  //         new rotations from the new Euler angles (:540-544)
  double* xout = Xout + s.o_state + (long long)f * 7 * J;
  const double* xpv = Xin + s.o_state + (long long)(f > 0 ? f - 1 : 0) * 7 * J;
  const double* xav = Xin + s.o_state + (long long)(f < F - 1 ? f + 1 : F - 1) * 7 * J;
  IK_FOR(j, J) {
    double ep[3], ea[3], en[3];
    quat_to_euler(xpv + 4 * j, ep); quat_to_euler(xav + 4 * j, ea);
    for (int a = 0; a < 3; ++a) {
      const int v = 3 * j + a;
      double dx = 0.0;
      for (int r = 0; r < R; ++r) dx += jm[(long long)r * nvar + v] * L.y[r];
      en[a] = L.x[v] + dx + P.smoothness * (ep[a] + ea[a] - 2 * L.x[v]);
    }
    euler_to_quat(en, xout + 4 * j);
    for (int a = 0; a < 3; ++a) {
      const int v = 3 * J + 3 * j + a;
      double xn = L.x[v];
      if (P.translate) {
        double dx = 0.0;
        for (int r = 0; r < R; ++r) dx += jm[(long long)r * nvar + v] * L.y[r];
        xn += dx + P.smoothness * (xpv[4 * J + 3 * j + a] + xav[4 * J + 3 * j + a] - 2 * L.x[v]);
      }
      xout[4 * J + 3 * j + a] = xn;
    }
  }
}

}  // namespace chd_ik

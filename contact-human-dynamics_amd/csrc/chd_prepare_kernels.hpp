// chd_prepare_kernels.hpp -- the per-frame numerics of `prepare_input` (towr_utils.py:451-777) for one frame of one skeleton.
//
// The same source compiles for gfx950 (hipcc: one thread per frame, chd_prepare.hip) and, with CHD_HOST_EMU, as a host function the CPU tests compare with
// the NumPy mirror (prepare_input.py, which tests/test_prepare_input.py pins to the files the reference's own prepare_input wrote).
//
//   pass 1 (towr_utils.py:483-535): the body with root rotation and translation zeroed -> global joint positions -> centre of mass (:803-810: mass fraction x
//           mean of the segment's joints) -> hip offsets from it; the body again with the root at -COM -> segment centres -> inertia about the COM
//           (sum over the segments of m (|r|^2 I - r r^T), :521-535);
//   pass 2 (:542-655): the animation as it is, heel joints included -> COM trajectory, toe / heel positions, toe-heel distance.
// Output in the solver's frame: p_solver = -0.01 p[[x, z, y]] (:519-521, 568-571).
#pragma once
#include <math.h>

#include "../../include/chd_prepare.h"

#if defined(__HIPCC__) && !defined(CHD_HOST_EMU)
#define PREP_HD __host__ __device__ inline
#else
#define PREP_HD static inline
#endif

namespace chd_prep {

// rotation matrix of a quaternion as Quaternions.transforms writes it (Quaternions.py:301-324; no normalisation)
PREP_HD void quat_to_matrix(const double* q, double* m) {
  const double w = q[0], x = q[1], y = q[2], z = q[3];
  const double x2 = x + x, y2 = y + y, z2 = z + z;
  m[0] = 1.0 - (y * y2 + z * z2); m[1] = x * y2 - w * z2; m[2] = x * z2 + w * y2;
  m[3] = x * y2 + w * z2; m[4] = 1.0 - (x * x2 + z * z2); m[5] = y * z2 - w * x2;
  m[6] = x * z2 - w * y2; m[7] = y * z2 + w * x2; m[8] = 1.0 - (x * x2 + y * y2);
}

// Animation.transforms_global (Animation.py:294-323): child = parent o local.  R: J x 9 global rotations, P: J x 3 global positions.
// `zero_root`: the root's rotation is the identity; `root_pos`: the root's translation (nullptr = the animation's).  With `positions_only` the rotations
// already in R are reused (the second half of pass 1 only moves the root).
PREP_HD void forward_kinematics(const chd_prep_skeleton& S, const int J, const double* rot, const double* pos, const bool zero_root, const double* root_pos,
                                const bool positions_only, double* R, double* P) {
  for (int j = 0; j < J; ++j) {
    const int a = S.parents[j];
    if (!positions_only) {
      double Rl[9];
      const double ident[4] = {1.0, 0.0, 0.0, 0.0};
      quat_to_matrix((j == 0 && zero_root) ? ident : rot + 4 * j, Rl);
      if (a < 0) { for (int k = 0; k < 9; ++k) R[9 * j + k] = Rl[k]; }
      else {
        const double* Ra = R + 9 * a;
        for (int r = 0; r < 3; ++r)
          for (int c = 0; c < 3; ++c) R[9 * j + 3 * r + c] = Ra[3 * r] * Rl[c] + Ra[3 * r + 1] * Rl[3 + c] + Ra[3 * r + 2] * Rl[6 + c];
      }
    }
    if (a < 0) { for (int k = 0; k < 3; ++k) P[3 * j + k] = root_pos ? root_pos[k] : pos[3 * j + k]; }
    else {
      const double* Ra = R + 9 * a; const double* pl = pos + 3 * j;
      for (int r = 0; r < 3; ++r) P[3 * j + r] = P[3 * a + r] + (Ra[3 * r] * pl[0] + Ra[3 * r + 1] * pl[1] + Ra[3 * r + 2] * pl[2]);
    }
  }
}

// sum over the segments of fraction x mean of the segment's joints (towr_utils.py:803-810); `solver`: of the positions in the solver's frame
PREP_HD void centre_of_mass(const chd_prep_skeleton& S, const double* P, const bool solver, double* com) {
  com[0] = com[1] = com[2] = 0.0;
  for (int s = 0; s < S.n_segments; ++s) {
    double m[3] = {0.0, 0.0, 0.0};
    const int n = S.seg_first[s + 1] - S.seg_first[s];
    for (int k = S.seg_first[s]; k < S.seg_first[s + 1]; ++k) {
      const double* p = P + 3 * S.seg_joint[k];
      if (solver) { m[0] += -0.01 * p[0]; m[1] += -0.01 * p[2]; m[2] += -0.01 * p[1]; }
      else { m[0] += p[0]; m[1] += p[1]; m[2] += p[2]; }
    }
    for (int d = 0; d < 3; ++d) com[d] += S.seg_mass_fraction[s] * (m[d] / n);
  }
}

// one frame.  R, P: scratch for n_joints x 9 / n_joints x 3 doubles.
PREP_HD void prep_frame(const chd_prep_skeleton& S, const double* rot, const double* pos, double* out, double* R, double* P) {
  const int J0 = S.n_joints_body, J1 = S.n_joints;
  // ---- pass 1
  const double zero[3] = {0.0, 0.0, 0.0};
  forward_kinematics(S, J0, rot, pos, true, zero, false, R, P);
  double com[3];
  centre_of_mass(S, P, false, com);
  for (int h = 0; h < 2; ++h) {
    const double* p = P + 3 * S.hip_inds[h];
    out[3 * h] = -0.01 * (p[0] - com[0]); out[3 * h + 1] = -0.01 * (p[2] - com[2]); out[3 * h + 2] = -0.01 * (p[1] - com[1]);
  }
  const double mcom[3] = {0.0 - com[0], 0.0 - com[1], 0.0 - com[2]};
  forward_kinematics(S, J0, rot, pos, true, mcom, true, R, P);
  double I[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};        // xx yy zz xy xz yz
  for (int s = 0; s < S.n_segments; ++s) {
    double r[3] = {0.0, 0.0, 0.0};
    const int n = S.seg_first[s + 1] - S.seg_first[s];
    for (int k = S.seg_first[s]; k < S.seg_first[s + 1]; ++k) {
      const double* p = P + 3 * S.seg_joint[k];
      r[0] += -0.01 * p[0]; r[1] += -0.01 * p[2]; r[2] += -0.01 * p[1];
    }
    for (int d = 0; d < 3; ++d) r[d] /= n;
    const double m = S.seg_mass_fraction[s] * S.mass;
    const double rr = r[0] * r[0] + r[1] * r[1] + r[2] * r[2];
    I[0] += m * (rr - r[0] * r[0]); I[1] += m * (rr - r[1] * r[1]); I[2] += m * (rr - r[2] * r[2]);
    I[3] += m * (0.0 - r[0] * r[1]); I[4] += m * (0.0 - r[0] * r[2]); I[5] += m * (0.0 - r[1] * r[2]);
  }
  for (int k = 0; k < 6; ++k) out[6 + k] = I[k];
  // ---- pass 2
  forward_kinematics(S, J1, rot, pos, false, nullptr, false, R, P);
  centre_of_mass(S, P, true, out + 12);
  const int feet[4] = {S.toe_inds[0], S.heel_inds[0], S.toe_inds[1], S.heel_inds[1]};
  for (int f = 0; f < 4; ++f) {
    const double* p = P + 3 * feet[f];
    out[15 + 3 * f] = -0.01 * p[0]; out[16 + 3 * f] = -0.01 * p[2]; out[17 + 3 * f] = -0.01 * p[1];
  }
  const double dx = out[15] - out[18], dy = out[16] - out[19], dz = out[17] - out[20];
  out[27] = sqrt(dx * dx + dy * dy + dz * dz);
}

}  // namespace chd_prep

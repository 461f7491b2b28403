// Host emulation of the prepare_input kernel (chd_prepare_kernels.hpp compiled with CHD_HOST_EMU) -- TEST INFRASTRUCTURE ONLY.
#define CHD_HOST_EMU 1
#include <vector>
#include "../../contact-human-dynamics_amd/csrc/chd_prepare_kernels.hpp"
extern "C" int prep_emu_frames(const chd_prep_skeleton* S, long long n, const double* rot, const double* pos, double* out) {
  std::vector<double> R(9 * CHD_PREP_MAX_JOINTS), P(3 * CHD_PREP_MAX_JOINTS);
  for (long long f = 0; f < n; ++f) chd_prep::prep_frame(*S, rot + f * S->n_joints * 4, pos + f * S->n_joints * 3, out + f * CHD_PREP_OUT_STRIDE, R.data(), P.data());
  return 0;
}

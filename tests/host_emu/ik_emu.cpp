// CPU emulation of the IK back-projection kernel source (chd_ik_kernels.hpp compiled with -DCHD_HOST_EMU): the same
// per-frame step, driven by the same host packing code as the HIP library, one emulated thread.  Test infrastructure.
#include <memory>
#include <string>
#include <vector>

#include "../../contact-human-dynamics_amd/csrc/chd_ik_host.hpp"

using namespace chd_ik;

static std::string g_err;

extern "C" {
const char* ik_emu_last_error() { return g_err.c_str(); }
int ik_emu_solve_batch(const chd_ik_config* cfg, int B, const chd_ik_seq* in) {
  IkBatch bt;
  if (!bt.build(B, in)) { g_err = bt.err; return 1; }
  const IkParams P = params_of(cfg);
  std::vector<double> x0 = bt.state, x1 = bt.state, scratch((size_t)IkLds::doubles(bt.max_J, bt.max_T));
  std::vector<unsigned short> pairs((size_t)IkLds::pair_entries());
  IkLds::fill_pairs(pairs.data());
  IkLds L;
  L.carve(scratch.data(), bt.max_J, bt.max_T);
  L.pair = pairs.data();
  double* cur = x0.data(); double* nxt = x1.data();
  for (int it = 0; it < P.iterations; ++it) {
    for (size_t wg = 0; wg < bt.frame_seq.size(); ++wg)
      ik_step_frame(bt.seqs[bt.frame_seq[wg]], bt.frame_idx[wg], P, bt.ipool.data(), bt.dpool.data(), cur, nxt, L);
    double* t = cur; cur = nxt; nxt = t;
  }
  bt.scatter(cur, in);
  return 0;
}
}

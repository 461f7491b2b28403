// CPU emulation of the kinematic-optimisation kernel source (chd_kinopt_kernels.hpp compiled with -DCHD_HOST_EMU): the same
// solve, driven by the same host packing code as the HIP library, one emulated thread.  Test infrastructure.
#include <string>
#include <vector>

#include "../../contact-human-dynamics_amd/csrc/chd_kinopt_host.hpp"

using namespace chd_kin;

static std::string g_err;

extern "C" {
const char* kin_emu_last_error() { return g_err.c_str(); }
void kin_emu_config_default(chd_kin_config* cfg) { config_default(cfg); }

int kin_emu_solve_batch(const chd_kin_config* cfg, int B, chd_kin_seq* in) {
  KinBatch bt;
  if (!bt.build(cfg, B, in)) { g_err = bt.err; return 1; }
  const int lds_doubles = cfg->reserved[1] > 0 ? cfg->reserved[1] : 18432;
  std::vector<double> work((size_t)bt.work_total), stats(8 * (size_t)B), red(48), lds((size_t)lds_doubles);
  for (int b = 0; b < B; ++b) {
    KinCtx c;
    kin_bind(c, &bt.seqs[b], &bt.P, bt.dpool.data(), bt.ipool.data(), work.data(), red.data(), lds.data(), lds_doubles);
    kin_solve(c, bt.state.data() + bt.seqs[b].o_x, stats.data() + 8 * b);
  }
  bt.scatter(bt.state.data(), stats.data(), in);
  return 0;
}

// building blocks at the video's start point (analysis / unit tests): mode 0: out = residual (m); 1: out = J v (m), vec = v (n);
// 2: out = J^T u (n), vec = u (m); 3: out = LSMR solution of min |J s - f|^2 + damp^2 |s|^2 (n) after cfg->lsmr_maxiter iterations,
// *aux = damp in, iteration count out
int kin_emu_probe(const chd_kin_config* cfg, chd_kin_seq* in, int mode, const double* vec, double* out, double* aux) {
  KinBatch bt;
  if (!bt.build(cfg, 1, in)) { g_err = bt.err; return 1; }
  const int lds_doubles = cfg->reserved[1] > 0 ? cfg->reserved[1] : 18432;
  std::vector<double> work((size_t)bt.work_total), red(48), lds((size_t)lds_doubles);
  KinCtx c;
  kin_bind(c, &bt.seqs[0], &bt.P, bt.dpool.data(), bt.ipool.data(), work.data(), red.data(), lds.data(), lds_doubles);
  const double* x = bt.state.data();
  const long long n = bt.seqs[0].n, m = bt.seqs[0].m;
  kin_residual(c, x, c.w.PN, c.w.RGN, c.w.Fv);
  kin_linearise(c, x);
  if (mode == 0) { for (long long i = 0; i < m; ++i) out[i] = c.w.Fv[i]; }
  else if (mode == 1) { kin_jv(c, vec, c.w.T2); for (long long i = 0; i < m; ++i) out[i] = c.w.T2[i]; }
  else if (mode == 2) { kin_jtu(c, vec, c.w.G); for (long long i = 0; i < n; ++i) out[i] = c.w.G[i]; }
  else { int istop = 0; const int it = kin_lsmr(c, c.w.Fv, *aux, &istop); for (long long i = 0; i < n; ++i) out[i] = c.w.GN[i]; *aux = it; }
  return 0;
}
}

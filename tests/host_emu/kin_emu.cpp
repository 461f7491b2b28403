// CPU emulation of the kinematic-optimisation kernel source (chd_kinopt_kernels.hpp compiled with -DCHD_HOST_EMU): the same
// solve, driven by the same host packing code as the HIP library; the G workgroups of a clip's cluster are emulated one after the
// other, phase by phase, each with its own "LDS" block.  Test infrastructure.
#include <algorithm>
#include <string>
#include <vector>

#include "../../contact-human-dynamics_amd/csrc/chd_kinopt_host.hpp"

using namespace chd_kin;

static std::string g_err;

namespace {
// one clip's cluster: contexts, LDS blocks
struct EmuCluster {
  KinCtx c;
  KinClip clip; KinLsmr lsmr; double gath[KC_MAXG * KC_PARTS];
  std::vector<KinWg> wg;
  std::vector<std::vector<double>> lds;
  void bind(const KinBatch& bt, int b, double* work) {
    const int G = bt.cluster[b];
    wg.resize(G); lds.assign(G, std::vector<double>((size_t)bt.lds_doubles, 0.0));
    c.k = &clip; c.S = &lsmr; c.gath = gath; c.P = &bt.P; c.G = G;
    kin_bind_clip(clip, &bt.seqs[b], bt.dpool.data(), bt.ipool.data());
    c.wg = wg.data();
    for (int g = 0; g < G; ++g) kin_bind_wg(c, wg[g], g, work, lds[g].data(), bt.lds_doubles);
  }
  // gather / scatter of a vector held in slices
  void get_n(int sel, double* out) const { for (const KinWg& w : wg) for (int i = 0; i < w.nf * NV; ++i) out[(long long)NV * w.a + i] = w.nv[sel][i]; }
  void set_n(int sel, const double* in) { for (KinWg& w : wg) for (int i = 0; i < w.nf * NV; ++i) w.nv[sel][i] = in[(long long)NV * w.a + i]; }
  // the reference's row order (term by term) <-> frame-major
  static long long ref_row(int F, int f, int t) {
    const long long o2 = 56LL * F, o3 = o2 + 84LL * (F - 1), o4 = o3 + 84LL * (F - 2), o5 = o4 + 84LL * F, o6 = o5 + 84LL * (F - 1), o7 = o6 + 28LL * F;
    if (t < R_VEL) return 56LL * f + t;
    if (t < R_ACC) return f < F - 1 ? o2 + 84LL * f + (t - R_VEL) : -1;
    if (t < R_DATA) return f < F - 2 ? o3 + 84LL * f + (t - R_ACC) : -1;
    if (t < R_CVEL) return o4 + 84LL * f + (t - R_DATA);
    if (t < R_FLOOR) return f < F - 1 ? o5 + 84LL * f + (t - R_CVEL) : -1;
    if (t < R_EUL) return o6 + 28LL * f + (t - R_FLOOR);
    return f < F - 1 ? o7 + 87LL * f + (t - R_EUL) : -1;
  }
  void get_m(int sel, double* out) const {
    const int F = c.k->F;
    for (const KinWg& w : wg) for (int l = 0; l < w.nf; ++l) for (int t = 0; t < NR; ++t) { const long long r = ref_row(F, w.a + l, t); if (r >= 0) out[r] = w.mv[sel][NR * l + t]; }
  }
  void set_m(int sel, const double* in) {
    const int F = c.k->F;
    for (KinWg& w : wg) for (int l = 0; l < w.nf; ++l) for (int t = 0; t < NR; ++t) { const long long r = ref_row(F, w.a + l, t); w.mv[sel][NR * l + t] = r >= 0 ? in[r] : 0.0; }
  }
};
}  // namespace

extern "C" {
const char* kin_emu_last_error() { return g_err.c_str(); }
void kin_emu_config_default(chd_kin_config* cfg) { config_default(cfg); }
int kin_emu_cluster_size(const chd_kin_config* cfg, int n_frames) {
  const int lds = cfg->reserved[1] > 0 ? cfg->reserved[1] : (int)KIN_LDS_DOUBLES_DEFAULT;
  int cap = cfg->reserved[2] > 0 ? cfg->reserved[2] : lds_frames(lds);
  if (cap < 2) cap = 2;
  return cluster_size(n_frames, cap);
}

int kin_emu_solve_batch(const chd_kin_config* cfg, int B, chd_kin_seq* in) {
  KinBatch bt;
  if (!bt.build(cfg, B, in)) { g_err = bt.err; return 1; }
  std::vector<double> work((size_t)bt.work_total), stats((size_t)KIN_STATS * B);
  for (int b = 0; b < B; ++b) {
    EmuCluster cl;
    cl.bind(bt, b, work.data());
    kin_solve(cl.c, bt.state.data() + bt.seqs[b].o_x, stats.data() + (size_t)KIN_STATS * b);
  }
  bt.scatter(bt.state.data(), stats.data(), in);
  return 0;
}

// building blocks at the video's start point (analysis / unit tests), vectors in the reference's order: mode 0: out = residual (m); 1: out = J v (m), vec = v (n);
// 2: out = J^T u (n), vec = u (m); 3: out = LSMR solution of min |J s - f|^2 + damp^2 |s|^2 (n) after cfg->lsmr_maxiter iterations,
// *aux = damp in, iteration count out
int kin_emu_probe(const chd_kin_config* cfg, chd_kin_seq* in, int mode, const double* vec, double* out, double* aux) {
  KinBatch bt;
  if (!bt.build(cfg, 1, in)) { g_err = bt.err; return 1; }
  std::vector<double> work((size_t)bt.work_total);
  EmuCluster cl;
  cl.bind(bt, 0, work.data());
  KinCtx& c = cl.c;
  for (KinWg& w : cl.wg) { for (int i = 0; i < HALO_V; ++i) w.vh[i] = 0.0; for (int i = 0; i < HALO_U; ++i) w.uh[i] = 0.0; for (int idx = 0; idx < w.nf * NJ; ++idx) { const long long gi = (long long)w.a * NJ + idx; w.DW[idx] = c.k->wt[3] * c.k->data_w[gi]; w.CT[idx] = (c.k->contact[gi] == 1 ? 1 : 0) | ((gi >= NJ && c.k->contact[gi - NJ] == 1) ? 2 : 0); } }
  cl.set_n(N_X, bt.state.data());
  kin_residual(c, N_X, M_FV);
  kin_linearise(c, N_X);
  KoAcc none[KC_NW][KC_PARTS];
  if (mode == 0) cl.get_m(M_FV, out);
  else if (mode == 1) { cl.set_n(N_S0, vec); kin_exchange_v(c, N_S0); kin_jv<false, false>(c, N_S0, M_T2, 1.0, 0.0, false, none); cl.get_m(M_T2, out); }
  else if (mode == 2) { cl.set_m(M_T1, vec); kin_exchange_u(c, M_T1); kin_jtu<false, false>(c, M_T1, N_G, 1.0, 0.0, false, none); cl.get_n(N_G, out); }
  else { int istop = 0; const int it = kin_lsmr(c, *aux, &istop); cl.get_n(N_GN, out); *aux = it; }
  return 0;
}
}

// Host emulation of the solver kernel — TEST INFRASTRUCTURE ONLY.
// Compiles contact-human-dynamics_amd/csrc/chd_kernels.hpp with CHD_HOST_EMU (one "thread",
// barriers are no-ops) so that the phases of the HIP kernel can be compared with the oracle
// in the CPU-only container.  Never loaded by the product; libchd_phys.so has no CPU path.
#define CHD_HOST_EMU 1
#include <cstdio>
#include <memory>
#include "../../contact-human-dynamics_amd/csrc/chd_model.hpp"
#include "../../contact-human-dynamics_amd/csrc/chd_kernels.hpp"

using namespace chd;

struct Emu {
  SeqModel M;
  chd_config cfg;
  std::vector<double> wd, out_d, lds;
  std::vector<int> wi, out_i;
  Ctx ctx;
  void bind() {
    M.d.cd = M.cd.data(); M.d.ci = M.ci.data(); M.d.wd = wd.data(); M.d.wi = wi.data();
    M.d.out_d = out_d.data(); M.d.out_i = out_i.data();
  }
};

extern "C" {
void* emu_create(const chd_seq_in* in, const chd_config* cfg) {
  try {
    auto e = std::make_unique<Emu>();
    e->cfg = *cfg;
    e->M.build(*in, *cfg);
    e->wd.assign(e->M.wd_size, 0.0); e->wi.assign(e->M.wi_size, 0);
    e->out_d.assign(out_d_size(e->M.d.cap, e->M.d.tot_entries + e->M.d.tot_phases), 0.0); e->out_i.assign(out_i_size(e->M.d.cap), 0);
    e->lds.assign(1 << 20, 0.0);
    e->bind();
    return e.release();
  } catch (const std::exception& ex) { std::fprintf(stderr, "emu_create: %s\n", ex.what()); return nullptr; }
}
void emu_destroy(void* h) { delete (Emu*)h; }
void emu_sizes(void* h, int stage, int* out) {
  Emu* e = (Emu*)h; const StageDesc& S = e->M.d.st[stage];
  out[0] = S.n; out[1] = S.m; out[2] = S.Nb; out[3] = S.bc; out[4] = S.w; out[5] = S.nnz_jac; out[6] = S.valid; out[7] = e->M.d.cap;
}
// dense J (m x n) and H (n x n) from the unfactored KKT storage
static void dense_from_K(Emu* e, int stage, double* J, double* H) {
  Ctx c; c.lds = e->lds.data(); c.lds_cap = (int)e->lds.size();
  bind_stage(c, &e->M.d, stage);
  const int n = c.n, m = c.m;
  if (J) for (int i = 0; i < m; ++i) for (int j = 0; j < n; ++j) J[(size_t)i * n + j] = kget(c, c.pos_row[i], c.pos_var[j]);
  if (H) for (int a = 0; a < n; ++a) for (int b = 0; b < n; ++b) H[(size_t)a * n + b] = kget(c, c.pos_var[a], c.pos_var[b]);
}
int emu_eval_lam(void* h, int stage, const double* x, const double* lam, double* x_out, double* f, double* grad, double* cvals, double* J, double* H);
int emu_eval(void* h, int stage, const double* x, double* x_out, double* f, double* grad, double* cvals, double* J, double* H) {
  return emu_eval_lam(h, stage, x, nullptr, x_out, f, grad, cvals, J, H);
}
int emu_eval_lam(void* h, int stage, const double* x, const double* lam, double* x_out, double* f, double* grad, double* cvals, double* J, double* H) {
  Emu* e = (Emu*)h; e->bind();
  const StageDesc& S = e->M.d.st[stage];
  Ctx c; c.q = &e->M.d;
  double fo[2];
  debug_eval(&e->M.d, e->ctx, stage, x != nullptr, e->lds.data(), (int)e->lds.size(), (double*)x, (double*)lam, fo);
  if (f) *f = fo[0];
  if (x_out) for (int j = 0; j < S.n; ++j) x_out[j] = VN(c, VN_X)[j];
  if (grad) for (int j = 0; j < S.n; ++j) grad[j] = VN(c, VN_G)[j];
  if (cvals) for (int i = 0; i < S.m; ++i) cvals[i] = VM(c, VM_C)[i];
  dense_from_K(e, stage, J, H);
  return (int)fo[1];
}
// factor/solve self test: builds K0 at the initial point of `stage`, adds diag, solves K x = b
int emu_linsolve(void* h, int stage, double dw, double dval, const double* b, double* x, int refine) {
  Emu* e = (Emu*)h; e->bind();
  double fo[2];
  debug_eval(&e->M.d, e->ctx, stage, 0, e->lds.data(), (int)e->lds.size(), nullptr, nullptr, fo);
  Ctx c; c.lds = e->lds.data(); c.lds_cap = 18432;
  bind_stage(c, &e->M.d, stage);
  double* diag = VK(c, VK_DIAG); int* sign = e->M.d.wi + e->M.d.o_sign;
  const double* Dw = e->M.d.cd + c.S->o_Dw;
  for (int j = 0; j < c.n; ++j) { diag[c.pos_var[j]] = dw * Dw[j]; sign[c.pos_var[j]] = 1; }
  for (int i = 0; i < c.m; ++i) { diag[c.pos_row[i]] = -dval; sign[c.pos_row[i]] = -1; }
  kfactor(c, diag, sign);
  double* rhs = VK(c, VK_RHS); double* sol = VK(c, VK_SOL);
  for (int i = 0; i < c.N; ++i) rhs[i] = b[i];
  ksolve(c, rhs, sol, diag, refine);
  for (int i = 0; i < c.N; ++i) x[i] = sol[i];
  // residual check: K x - b
  kmatvec(c, sol, VK(c, VK_T1), diag, nullptr);
  double worst = 0;
  for (int i = 0; i < c.N; ++i) worst = std::max(worst, std::fabs(VK(c, VK_T1)[i] - b[i]));
  std::fprintf(stderr, "emu_linsolve: N=%d Nb=%d bc=%d w=%d bad_pivots=%d residual=%.3e\n", c.N, c.Nb, c.bc, c.w, c.n_bad_pivots, worst);
  return c.n_bad_pivots;
}
int emu_solve(void* h, int stage_first, int stage_last, int lds_doubles) {
  Emu* e = (Emu*)h; e->bind();
  run_sequence(&e->M.d, e->ctx, e->lds.data(), lds_doubles > 0 ? lds_doubles : 18432, e->cfg.tol, e->cfg.stall_window, stage_first, stage_last);
  return 0;
}
// stage-4 fallback: rebuild the tables of stage index 5 with the durations stage 3 left
int emu_rebuild_fallback(void* h) {
  Emu* e = (Emu*)h; e->bind();
  std::vector<double> cur[4];
  const double* ph = e->out_d.data() + out_d_state_off(e->M.d.cap) + 2LL * (e->M.d.tot_entries + e->M.d.tot_phases) + e->M.d.tot_entries;      // the durations stage 3 left (save_state, slot 2)
  for (int k = 0; k < 4; ++k) cur[k].assign(ph + e->M.d.phase_off[k], ph + e->M.d.phase_off[k] + e->M.d.n_phase[k]);
  e->M.build_stage(5, e->cfg, cur, false);
  e->bind();
  return e->M.d.st[5].valid;
}
// the point behind snapshot `snap` in the NLP's variables (what chd_debug_get_state returns on the device): node variables, then all phase durations
int emu_get_state(void* h, int snap, double* node_vars, double* phase_durations) {
  Emu* e = (Emu*)h;
  const SeqDesc& d = e->M.d;
  const long long ns = d.tot_entries + d.tot_phases;
  const double* st = e->out_d.data() + out_d_state_off(d.cap) + snap * ns;
  for (int sp = 0; sp < N_SPLINES; ++sp)
    for (int k = 0; k < d.sp[sp].n_nodes * 6; ++k) {
      const int v = e->M.ci[d.o_varof + d.sp[sp].node_off + k];
      if (v >= 0) node_vars[d.sp[sp].var_off + v] = st[d.sp[sp].node_off + k];
    }
  for (int k = 0; k < d.tot_phases; ++k) phase_durations[k] = st[d.tot_entries + k];
  return d.n_nodesvars;
}
void emu_get_out(void* h, double* od, int* oi) {
  Emu* e = (Emu*)h;
  std::copy(e->out_d.begin(), e->out_d.begin() + out_d_state_off(e->M.d.cap), od);      // statistics, snapshots, timers (not the saved state)
  std::copy(e->out_i.begin(), e->out_i.end(), oi);
}
}

// envelope statistics of the unfactored KKT matrix of `stage` at the initial point (analysis helper)
extern "C" void emu_envelope(void* h, int stage, double* out) {
  Emu* e = (Emu*)h; e->bind();
  double fo[2];
  debug_eval(&e->M.d, e->ctx, stage, 0, e->lds.data(), (int)e->lds.size(), nullptr, nullptr, fo);
  Ctx c; c.lds = e->lds.data(); c.lds_cap = (int)e->lds.size();
  bind_stage(c, &e->M.d, stage);
  long long env = 0, band = 0;
  int maxreach = 0;
  std::vector<int> hist(8, 0);
  for (int i = 0; i < c.Nb; ++i) {
    int first = i;
    for (int k = std::max(0, i - c.w); k < i; ++k) if (c.K0b[(long long)i * c.W2 + (k - i + c.w)] != 0.0) { first = k; break; }
    env += i - first; band += std::min(i, c.w);
    maxreach = std::max(maxreach, i - first);
    hist[std::min(7, (i - first) * 8 / (c.w + 1))]++;
  }
  out[0] = (double)env; out[1] = (double)band; out[2] = maxreach; out[3] = c.w; out[4] = c.Nb;
  for (int k = 0; k < 8; ++k) out[5 + k] = hist[k];
}

// border statistics of the unfactored KKT matrix of `stage` at the initial point (analysis helper):
// out[0] = mean over border rows of the fraction of band columns right of the row's first nonzero, out[1] = bc
extern "C" void emu_border_first(void* h, int stage, double* out, int* first_out) {
  Emu* e = (Emu*)h; e->bind();
  double fo[2];
  debug_eval(&e->M.d, e->ctx, stage, 0, e->lds.data(), (int)e->lds.size(), nullptr, nullptr, fo);
  Ctx c; c.lds = e->lds.data(); c.lds_cap = (int)e->lds.size();
  bind_stage(c, &e->M.d, stage);
  double acc = 0;
  for (int r = 0; r < c.bc; ++r) {
    int first = c.Nb;
    for (int k = 0; k < c.Nb; ++k) if (c.K0x[(long long)r * c.LD + k] != 0.0) { first = k; break; }
    if (first_out) { first_out[r] = first; first_out[512 + r] = c.env[2 * (c.Nb + r)]; }
    acc += (double)(c.Nb - first) / c.Nb;
  }
  out[0] = c.bc ? acc / c.bc : 0; out[1] = c.bc; out[2] = c.Nb; out[3] = c.w;
}

// average number of active window rows per 32-column panel of `stage` (analysis helper): out[0] = mean nact,
// out[1] = mean band part, out[2] = mean border part, out[3] = mean number of 16x16 tiles
extern "C" void emu_nact_stats(void* h, int stage, double* out) {
  Emu* e = (Emu*)h; e->bind();
  double fo[2];
  debug_eval(&e->M.d, e->ctx, stage, 0, e->lds.data(), (int)e->lds.size(), nullptr, nullptr, fo);
  Ctx c; c.lds = e->lds.data(); c.lds_cap = (int)e->lds.size();
  bind_stage(c, &e->M.d, stage);
  double sa = 0, sb = 0, sx = 0, st = 0; int np = 0, mx = 0;
  for (int c0 = 0; c0 < c.Nb; c0 += 32) {
    const int jb = std::min(32, c.Nb - c0), last = c0 + jb - 1;
    const int nbr = std::min(c.Nb - c0, jb + c.w), nbelow = nbr - jb;
    int nb_ = 0, nx_ = 0;
    for (int u = 0; u < nbelow; ++u) if (c.env[2 * (c0 + jb + u)] <= last) ++nb_;
    for (int r = 0; r < c.bc; ++r) if (c.env[2 * (c.Nb + r)] <= last) ++nx_;
    const int nact = nb_ + nx_, nt = (nact + 15) / 16;
    sa += nact; sb += nb_; sx += nx_; st += nt * (nt + 1) / 2; ++np; mx = std::max(mx, nact + jb);
  }
  out[0] = sa / np; out[1] = sb / np; out[2] = sx / np; out[3] = st / np; out[4] = np; out[5] = mx;      // mx: largest front (panel rows + active rows)
}

// front profile of `stage` (analysis helper): for panels of `nb` columns, out[3*p + 0] = active band rows below the panel,
// out[3*p + 1] = active border rows, out[3*p + 2] = jb.  Returns the number of panels.
extern "C" int emu_front_profile(void* h, int stage, int nb, int* out, int cap) {
  Emu* e = (Emu*)h; e->bind();
  double fo[2];
  debug_eval(&e->M.d, e->ctx, stage, 0, e->lds.data(), (int)e->lds.size(), nullptr, nullptr, fo);
  Ctx c; c.lds = e->lds.data(); c.lds_cap = (int)e->lds.size();
  bind_stage(c, &e->M.d, stage);
  int np = 0;
  for (int c0 = 0; c0 < c.Nb && np < cap; c0 += nb, ++np) {
    const int jb = std::min(nb, c.Nb - c0), last = c0 + jb - 1;
    const int nbr = std::min(c.Nb - c0, jb + c.w), nbelow = nbr - jb;
    int nb_ = 0, nx_ = 0;
    for (int u = 0; u < nbelow; ++u) if (c.env[2 * (c0 + jb + u)] <= last) ++nb_;
    for (int r = 0; r < c.bc; ++r) if (c.env[2 * (c.Nb + r)] <= last) ++nx_;
    out[3 * np] = nb_; out[3 * np + 1] = nx_; out[3 * np + 2] = jb;
  }
  return np;
}


// non-zero counts of the unfactored KKT matrix of `stage` at the initial point (analysis helper): out = band envelope entries (both triangles),
// band non-zeros, 64-byte lines of the band rows holding a non-zero, border entries right of each row's first position, border non-zeros, border lines
extern "C" void emu_nnz_stats(void* h, int stage, double* out) {
  Emu* e = (Emu*)h; e->bind();
  double fo[2];
  std::vector<double> lam(e->M.d.st[stage].m, 0.5);
  debug_eval(&e->M.d, e->ctx, stage, 0, e->lds.data(), (int)e->lds.size(), nullptr, lam.data(), fo);
  Ctx c; c.lds = e->lds.data(); c.lds_cap = (int)e->lds.size();
  bind_stage(c, &e->M.d, stage);
  long long env = 0, nz = 0, lines = 0, benv = 0, bnz = 0, blines = 0;
  for (int i = 0; i < c.Nb; ++i) {
    env += c.env[2 * i + 1] - c.env[2 * i] + 1;
    long long lastline = -1;
    for (int k = c.env[2 * i]; k <= c.env[2 * i + 1]; ++k) {
      const long long a = (long long)i * c.W2 + (k - i + c.w);
      if (c.K0b[a] != 0.0) { nz++; if (a / 8 != lastline) { lines++; lastline = a / 8; } }
    }
  }
  for (int r = 0; r < c.bc; ++r) {
    benv += c.LD - c.env[2 * (c.Nb + r)];
    long long lastline = -1;
    for (int k = 0; k < c.LD; ++k) { const long long a = (long long)r * c.LD + k; if (c.K0x[a] != 0.0) { bnz++; if (a / 8 != lastline) { blines++; lastline = a / 8; } } }
  }
  out[0] = (double)env; out[1] = (double)nz; out[2] = (double)lines; out[3] = (double)benv; out[4] = (double)bnz; out[5] = (double)blines; out[6] = c.Nb; out[7] = c.bc;
}

// The occupancy list of the unfactored KKT matrix (chd_kernels.hpp "KKT storage") after TWO evaluations of `stage` (at its initial point with multipliers
// lam0, then at x1 with lam1 -- the second one may move the pattern): out[0] = stored non-zeros that are not in the list (must be 0), out[1] = list entries,
// out[2] = bits set in the masks (band + border rows + the transposed border; must equal out[1]), out[3] = max |list product - dense product| of K0 x for the
// given x, out[4] = max |dense product|, out[5] = entries the second evaluation added to the list, out[6] = listed entries holding +0.0 or -0.0 (allowed)
extern "C" int emu_list_check(void* h, int stage, const double* lam0, const double* x1, const double* lam1, const double* xv, double* out) {
  Emu* e = (Emu*)h; e->bind();
  double fo[2];
  debug_eval(&e->M.d, e->ctx, stage, 0, e->lds.data(), (int)e->lds.size(), nullptr, (double*)lam0, fo);
  Ctx& c = e->ctx;
  const int nnz0 = c.csr_nnz;
  // second evaluation in the same stage (no kreset): what solve_stage does iteration after iteration
  double* x = VN(c, VN_X);
  for (int j = 0; j < c.n; ++j) x[j] = x1[j];
  for (int i = 0; i < c.m; ++i) VM(c, VM_LAM)[i] = lam1[i];
  eval_nlp(c, x, EV_FULL, VM(c, VM_C), VN(c, VN_G), VM(c, VM_LAM));
  const int N = c.N, Nb = c.Nb;
  std::vector<char> listed((size_t)N * N, 0);
  for (int i = 0; i < N; ++i)
    for (int k = c.csr_rp[i]; k < c.csr_rp[i + 1]; ++k) { if (c.csr_row[k] != i) return -1; listed[(size_t)i * N + c.csr_col[k]] = 1; }
  long long missing = 0, zeros = 0, bits = 0;
  std::vector<double> yd(N, 0.0);
  for (int i = 0; i < N; ++i)
    for (int j = 0; j < N; ++j) {
      if (i < Nb && j < Nb && (j - i > c.w || i - j > c.w)) continue;
      const double v = kget(c, i, j);
      if (v != 0.0 && !listed[(size_t)i * N + j]) ++missing;
      if (v == 0.0 && listed[(size_t)i * N + j]) ++zeros;
      yd[i] += v * xv[j];
    }
  for (long long k = 0; k < (long long)Nb * c.MW; ++k) bits += __builtin_popcountll(c.pmb[k]);
  for (long long k = 0; k < (long long)c.bc * c.LW; ++k) bits += __builtin_popcountll(c.pmx[k]);
  for (long long k = 0; k < (long long)Nb * c.CW; ++k) bits += __builtin_popcountll(c.pmt[k]);
  std::vector<double> xs(xv, xv + N), yl(N, 0.0);
  kmatvec(c, xs.data(), yl.data(), nullptr, nullptr);
  double err = 0, mag = 0;
  for (int i = 0; i < N; ++i) { err = std::max(err, std::fabs(yl[i] - yd[i])); mag = std::max(mag, std::fabs(yd[i])); }
  out[0] = (double)missing; out[1] = c.csr_nnz; out[2] = (double)bits; out[3] = err; out[4] = mag; out[5] = c.csr_nnz - nnz0; out[6] = (double)zeros;
  return c.err;
}

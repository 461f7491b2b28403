// ORACLE C API (test infrastructure only — see nlp_model.hpp header).
// Plain-C entry points so tests/ and bench.py's cpu_baseline leg can drive the
// CPU restatement through ctypes.  Not part of the product C-ABI (include/chd_phys.h).
#include "nlp_model.hpp"
#include "ipm_solver.hpp"

using namespace orc;

static int g_verbose = 0, g_inertia_retry = 1, g_stall_window = 0, g_lbfgs = 0, g_study_mask = 0;
extern "C" {
void orc_set_verbose(int v) { g_verbose = v; }
static double g_ratio_low = 0.0;
void orc_set_ratio_low(double v) { g_ratio_low = v; }      // chd_config.damping_rule = 1 <-> 0.25 (IpmOptions::ratio_low)
void orc_set_inertia_retry(int v) { g_inertia_retry = v; }
void orc_set_stall_window(int v) { g_stall_window = v; }
static double g_clip_cap = 1e300;
void orc_set_study_mask(int v, double clip_cap) { g_study_mask = v; g_clip_cap = clip_cap > 0 ? clip_cap : 1e300; }      // Problem::study_mask (0 = the shipped algorithm)
void orc_set_ipopt_like(int v) { g_lbfgs = v; }      // 1: L-BFGS(6) + mu_init 0.1 on every stage, l1 merit (round 2); 2: the same with IPOPT's filter line search (round 5): IpmOptions::lbfgs / filter (oracle-only study modes)

struct orc_seq_in {
  int F;
  double dt;
  const double* hip_l;
  const double* hip_r;
  double leg_len, heel_len, heel_dist, mass;
  const double* inertia;
  const double* com;
  const double* euler;
  const double* ltoe;
  const double* lheel;
  const double* rtoe;
  const double* rheel;
  double normal[3];
  double point[3];
  int start_contact[4];
  int n_phases[4];
  const double* durations[4];
};

struct orc_config {
  double w_com_lin, w_com_ang, w_ee, w_smooth, w_dur;
  int max_iter[6];
  double tol;
};

static SeqInput to_input(const orc_seq_in* in) {
  SeqInput s;
  s.F = in->F; s.dt = in->dt;
  auto cp = [&](const double* p, int cnt) { return std::vector<double>(p, p + cnt); };
  s.hip_l = cp(in->hip_l, in->F * 3); s.hip_r = cp(in->hip_r, in->F * 3);
  s.leg_len = in->leg_len; s.heel_len = in->heel_len; s.heel_dist = in->heel_dist; s.mass = in->mass;
  s.inertia = cp(in->inertia, in->F * 6);
  s.com = cp(in->com, in->F * 3); s.euler = cp(in->euler, in->F * 3);
  s.ltoe = cp(in->ltoe, in->F * 3); s.lheel = cp(in->lheel, in->F * 3);
  s.rtoe = cp(in->rtoe, in->F * 3); s.rheel = cp(in->rheel, in->F * 3);
  for (int d = 0; d < 3; ++d) { s.normal[d] = in->normal[d]; s.point[d] = in->point[d]; }
  for (int e = 0; e < 4; ++e) {
    s.start_contact[e] = in->start_contact[e];
    s.durations[e] = cp(in->durations[e], in->n_phases[e]);
  }
  return s;
}

static Config to_config(const orc_config* c) {
  Config k;
  if (c) {
    k.w_com_lin = c->w_com_lin; k.w_com_ang = c->w_com_ang; k.w_ee = c->w_ee; k.w_smooth = c->w_smooth; k.w_dur = c->w_dur;
    for (int i = 0; i < 6; ++i) k.max_iter[i] = c->max_iter[i];
    k.tol = c->tol;
  }
  return k;
}

void* orc_create(const orc_seq_in* in, const orc_config* cfg) {
  try { return new Problem(to_input(in), to_config(cfg)); } catch (...) { return nullptr; }
}
void orc_destroy(void* h) { delete (Problem*)h; }
void orc_set_stage(void* h, int stage) { ((Problem*)h)->set_stage(stage); }
int orc_n(void* h) { return ((Problem*)h)->n; }
int orc_m(void* h) { return ((Problem*)h)->m; }
double orc_total_time(void* h) { return ((Problem*)h)->T; }
void orc_get_x(void* h, double* x) { ((Problem*)h)->get_x(x); }
void orc_set_x(void* h, const double* x) { ((Problem*)h)->set_x(x); }
// all phase durations of the four end-effectors (NLP order, concatenated): for evaluating the model at a point another solver returned
// (tests/test_quality_gate.py: the HIP path's snapshots, chd_debug_get_state).  Call set_stage afterwards: the rows of a stage depend on the durations.
void orc_set_durations(void* h, const double* durs) {
  Problem* p = (Problem*)h;
  int k = 0;
  for (int e = 0; e < 4; ++e) {
    for (size_t i = 0; i < p->phase_dur[e].size(); ++i) p->phase_dur[e][i] = durs[k++];
    p->update_phase_spline_durations(e);
  }
}
int orc_n_phases(void* h, int e) { return (int)((Problem*)h)->phase_dur[e].size(); }
void orc_eval(void* h, const double* x, double* f, double* grad, double* c, double* J, double* H) {
  ((Problem*)h)->eval(x, f, grad, c, J, H);
}
void orc_eval_lam(void* h, const double* x, const double* lam, double* f, double* grad, double* c, double* J, double* H) {
  ((Problem*)h)->study_mask = g_study_mask; ((Problem*)h)->clip_cap = g_clip_cap;
  ((Problem*)h)->eval(x, f, grad, c, J, H, lam);
}
void orc_bounds(void* h, double* cl, double* cu) {
  Problem* p = (Problem*)h;
  std::vector<double> x(p->n), c(p->m);
  p->get_x(x.data());
  p->eval(x.data(), nullptr, nullptr, c.data(), nullptr, nullptr);
  for (int i = 0; i < p->m; ++i) { cl[i] = p->cl[i]; cu[i] = p->cu[i]; }
}
void orc_row_family(void* h, int* fam) {
  Problem* p = (Problem*)h;
  for (int i = 0; i < p->m; ++i) fam[i] = p->row_family[i];
}
// variable-set layout: offsets of the 10 node sets (+ durations) in x
void orc_var_offsets(void* h, int* off11) {
  Problem* p = (Problem*)h;
  for (int i = 0; i < 10; ++i) off11[i] = p->sp[i].var_off;
  off11[10] = p->n_nodesvars;
}

// Solution sampling (SaveSolution).  Returns number of samples; arrays sized cap*3 / cap.
int orc_sample_solution(void* h, int cap, int* num_frames_header, double* base_lin, double* base_ang_deg,
                        double* ee_pos /*4*cap*3*/, double* ee_force /*4*cap*3*/, int* contact /*4*cap*/) {
  Problem* p = (Problem*)h;
  Problem::Solution s = p->sample_solution();
  if (num_frames_header) *num_frames_header = s.num_frames_header;
  int ns = std::min(cap, s.n_samples);
  for (int i = 0; i < ns * 3; ++i) { base_lin[i] = s.base_lin[i]; base_ang_deg[i] = s.base_ang_deg[i]; }
  for (int e = 0; e < 4; ++e) {
    for (int i = 0; i < ns * 3; ++i) { ee_pos[(size_t)e * cap * 3 + i] = s.ee_pos[e][i]; ee_force[(size_t)e * cap * 3 + i] = s.ee_force[e][i]; }
    for (int i = 0; i < ns; ++i) contact[(size_t)e * cap + i] = s.contact[e][i];
  }
  return s.n_samples;
}

// One IPM solve of the current stage, starting from the current variable values.
// Returns status (0 solved to reference tol, 1 acceptable, -1 max-iter, -2 numerical failure).
int orc_solve_stage(void* h, int stage, int max_iter, double* stats /*8*/) {
  Problem* p = (Problem*)h;
  p->set_stage(stage);
  p->study_mask = g_study_mask; p->clip_cap = g_clip_cap;
  IpmOptions opt;
  opt.max_iter = max_iter > 0 ? max_iter : p->cfg.max_iter[stage];
  opt.ref_tol = p->cfg.tol;
  opt.verbose = g_verbose != 0;
  opt.ratio_low = g_ratio_low;
  opt.inertia_retry = g_inertia_retry != 0;
  opt.stall_window = g_stall_window;
  opt.lbfgs = g_lbfgs != 0;
  opt.filter = g_lbfgs >= 2;          // orc_set_ipopt_like(2): L-BFGS(6) + filter line search (round 5)
  IpmResult r = ipm_solve(*p, opt);
  if (stats) {
    stats[0] = r.iters; stats[1] = r.kkt_error; stats[2] = r.constr_viol; stats[3] = r.objective;
    stats[4] = r.mu; stats[5] = r.n_factor; stats[6] = r.N; stats[7] = r.bandwidth;
  }
  return r.status;
}

}  // extern "C"

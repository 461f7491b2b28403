// chd_model.hpp — host-side construction of the per-sequence NLP structure tables.
//
// Replaces the problem set-up half of the reference's `phys_optim` main():
//   ReadSkeletonInfo/ReadMotionInfo/ReadTerrainInfo/ReadContactInfo   phys_optim.cpp:155-267
//   initial/final base state, ee remap, polynomials per phase          phys_optim.cpp:428-540
//   NlpFormulation::GetVariableSets + initial guesses                  nlp_formulation.cpp:79-203
//   NodesVariablesDynamicEEMotion / ...EEForce index maps              nodes_variables_dynamic_phase_based.cpp:10-151
//   Parameters (which constraint families per stage, sampling steps)   parameters.cpp:46-139
// Nothing here evaluates the NLP: the output is a flat, pointer-free description
// (chd_device.hpp) that one workgroup of the HIP solver consumes.
#pragma once
#include <algorithm>
#include <cmath>
#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/chd_phys.h"
#include "chd_device.hpp"

namespace chd {

static const double kBigBound = 1e20;     // ifopt's +-infinity for bounds
static const double kG = 9.80665;         // TOWR DynamicModel::g_
static const double kMu = 0.5;            // TOWR HeightMap::friction_coeff_
static const double kForceMax = 1000.0;   // parameters.cpp:56 force_limit_in_normal_direction_
static const double kSlack = 0.05;        // [s] band head-room: how far a junction may move while durations are optimised

struct HostSpline {
  int n_nodes = 0, n_polys = 0, n_var = 0, var_off = 0, ee = -1;
  bool phase_based = false;
  std::vector<int> var_of;      // n_nodes*6
  std::vector<int> ph, kin, nin, isc;   // per polynomial (phase-based)
  bool const_node(int nd) const {
    if (!phase_based) return false;
    return (nd > 0 && isc[nd - 1]) || (nd < n_polys && isc[nd]);
  }
};

class SeqModel {
 public:
  SeqDesc d;
  std::vector<double> cd;
  std::vector<int> ci;
  long long wd_size = 0, wi_size = 0;
  HostSpline hs[N_SPLINES];
  std::vector<double> phase_in[N_EE];       // NLP ee order, as read from contact_info.txt
  int stage_m_cap[N_STAGES], stage_task_cap[N_STAGES];
  int w_cap = 0, N_cap = 0, bc_cap = 0;
  double alg_bytes_iter[N_STAGES];

  SeqModel() { std::memset(&d, 0, sizeof(d)); }

  // ---- small helpers -------------------------------------------------------
  static std::vector<double> sample_times(double T, double step) {   // TOWR TimeDiscretizationConstraint ctor
    std::vector<double> ts;
    double t = 0.0;
    ts.push_back(t);
    const int ns = (int)std::floor(T / step);
    for (int i = 0; i < ns; ++i) { t += step; ts.push_back(t); }
    ts.push_back(T);
    return ts;
  }
  static int locate(const std::vector<double>& cum_end, double t) {  // TOWR Spline::GetSegmentID
    const int n = (int)cum_end.size();
    for (int i = 0; i < n; ++i) if (cum_end[i] >= t - 1e-10) return i;
    return n - 1;
  }
  static std::vector<double> cumulate(const std::vector<double>& v) {
    std::vector<double> c(v.size());
    double t = 0;
    for (size_t i = 0; i < v.size(); ++i) { t += v[i]; c[i] = t; }
    return c;
  }
  int push_d(const std::vector<double>& v) { int o = (int)cd.size(); cd.insert(cd.end(), v.begin(), v.end()); return o; }
  int push_d(const double* p, int n) { int o = (int)cd.size(); cd.insert(cd.end(), p, p + n); return o; }
  int push_i(const std::vector<int>& v) { int o = (int)ci.size(); ci.insert(ci.end(), v.begin(), v.end()); return o; }
  int reserve_d(int n) { int o = (int)cd.size(); cd.resize(cd.size() + n, 0.0); return o; }
  int reserve_i(int n) { int o = (int)ci.size(); ci.resize(ci.size() + n, 0); return o; }

  // polynomial durations of spline s for given phase durations (NLP ee order)
  std::vector<double> poly_durations(int s, const std::vector<double>* phase_dur) const {
    const HostSpline& h = hs[s];
    std::vector<double> pd(h.n_polys);
    if (!h.phase_based) { for (int i = 0; i < h.n_polys; ++i) pd[i] = base_dur_[i]; return pd; }
    for (int i = 0; i < h.n_polys; ++i) pd[i] = phase_dur[h.ee][h.ph[i]] / h.nin[i];
    return pd;
  }

  // ---- build ----------------------------------------------------------------
  void build(const chd_seq_in& in, const chd_config& cfg) {
    const int F = in.F;
    if (F < 8) throw std::runtime_error("sequence too short (need at least 8 frames)");
    d.F = F; d.cap = F + 4; d.dt = in.dt;
    d.ratio_low = cfg.damping_rule == 1 ? 0.25 : 0.0;
    d.mass = in.mass; d.leg_len = in.leg_len; d.heel_len = in.heel_len; d.heel_dist = in.heel_dist;
    // NLP ee order 0 L-toe, 1 R-toe, 2 L-heel, 3 R-heel <- file slots 0, 2, 1, 3 (phys_optim.cpp:505-513)
    const int slot[4] = {0, 2, 1, 3};
    const double* ee_file[4] = {in.ltoe, in.lheel, in.rtoe, in.rheel};
    for (int e = 0; e < 4; ++e) {
      const int sl = slot[e];
      if (in.n_phases[sl] < 1) throw std::runtime_error("contact_info: an end-effector has no phases");
      phase_in[e].assign(in.durations[sl], in.durations[sl] + in.n_phases[sl]);
      d.start_contact[e] = in.start_contact[sl] ? 1 : 0;
      d.n_phase[e] = in.n_phases[sl];
    }
    double T = 0;
    for (int k = 0; k < in.n_phases[0]; ++k) T += in.durations[0][k];   // total time from the L-toe schedule (phys_optim.cpp:420-423)
    d.T = T;
    // SaveSolution writes int((T + 1e-5) / dt) + 1 samples (phys_optim.cpp:71-84); prepare_input makes T = (F - 1) dt
    // (towr_utils.py:440).  A schedule longer than the frame count implies does not fit the caller's F + 4 arrays.
    if ((int)((T + 1e-5) / in.dt) + 1 > F + 3) throw std::runtime_error("contact_info: the contact schedule is longer than nframes x dt");
    for (int e = 0; e < 4; ++e) {   // parameters.cpp:150 asserts all schedules share the total time
      double s = 0; for (double v : phase_in[e]) s += v;
      if (std::fabs(s - T) > 1e-6) throw std::runtime_error("contact_info: phase durations of the four end-effectors do not sum to the same total time");
    }
    const double nl = std::sqrt(in.normal[0] * in.normal[0] + in.normal[1] * in.normal[1] + in.normal[2] * in.normal[2]);
    if (!(nl > 0) || in.normal[2] == 0.0) throw std::runtime_error("terrain_info: degenerate floor normal");
    for (int k = 0; k < 3; ++k) { d.normal[k] = in.normal[k]; d.point[k] = in.point[k]; d.gdir[k] = -in.normal[k] / nl; }
    d.hx = -in.normal[0] / in.normal[2];      // ground_plane.cpp:29-41
    d.hy = -in.normal[1] / in.normal[2];
    {  // TOWR HeightMap::GetNormalizedBasis for a plane
      const double nv[3] = {-d.hx, -d.hy, 1.0}, a1[3] = {1, 0, d.hx}, a2[3] = {0, 1, d.hy};
      auto unit = [](const double* v, double* o) { double l = std::sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]); for (int k = 0; k < 3; ++k) o[k] = v[k] / l; };
      unit(nv, d.bn); unit(a1, d.bt1); unit(a2, d.bt2);
    }

    // ---- base spline durations: 0.1 s polynomials (parameters.cpp:109-125)
    base_dur_.clear();
    { double left = T; while (left > 1e-10) { base_dur_.push_back(left > 0.1 ? 0.1 : left); left -= 0.1; } }

    // ---- splines: index maps
    for (int b = 0; b < 2; ++b) {      // nlp_formulation.cpp:106-130 (NodesVariablesAll)
      HostSpline& h = hs[b];
      h.phase_based = false; h.ee = -1;
      h.n_polys = (int)base_dur_.size(); h.n_nodes = h.n_polys + 1;
      h.var_of.assign(h.n_nodes * 6, -1);
      int idx = 0;
      for (int nd = 0; nd < h.n_nodes; ++nd)
        for (int q = 0; q < 6; ++q) {
          // start and final base linear velocity are equality-bounded (nlp_formulation.cpp:120-121):
          // fixed variables are parameters of the NLP, not unknowns
          const bool fixed = (b == 0 && q >= 3 && (nd == 0 || nd == h.n_nodes - 1));
          h.var_of[nd * 6 + q] = fixed ? -1 : idx++;
        }
      h.n_var = idx;
    }
    for (int e = 0; e < 4; ++e) {
      for (int kind = 0; kind < 2; ++kind) {   // 0 motion, 1 force
        HostSpline& h = hs[(kind ? 6 : 2) + e];
        h.phase_based = true; h.ee = e;
        // motion: the contact phases are the constant ones; force: the swing phases (phys_optim.cpp:516-534)
        bool is_const = kind == 0 ? d.start_contact[e] != 0 : d.start_contact[e] == 0;
        h.ph.clear(); h.kin.clear(); h.nin.clear(); h.isc.clear();
        for (int p = 0; p < d.n_phase[e]; ++p) {     // BuildDynamicPolyInfos (nodes_variables_dynamic_phase_based.cpp:10-34)
          if (is_const) { h.ph.push_back(p); h.kin.push_back(0); h.nin.push_back(1); h.isc.push_back(1); }
          else {
            int np = 6;                                // GetPolyChangingPhase (phys_optim.cpp:289-312), parameters.cpp:51-53
            if (phase_in[e][p] > 2.0) np += (int)std::ceil((phase_in[e][p] - 2.0) * (6 / 2.0));
            for (int j = 0; j < np; ++j) { h.ph.push_back(p); h.kin.push_back(j); h.nin.push_back(np); h.isc.push_back(0); }
          }
          is_const = !is_const;
        }
        h.n_polys = (int)h.ph.size(); h.n_nodes = h.n_polys + 1;
        h.var_of.assign(h.n_nodes * 6, -1);
        int idx = 0;
        for (int nd = 0; nd < h.n_nodes; ++nd) {
          if (!h.const_node(nd)) {                     // free node: position and velocity, interleaved per dimension
            for (int dim = 0; dim < 3; ++dim) { h.var_of[nd * 6 + dim] = idx++; h.var_of[nd * 6 + 3 + dim] = idx++; }
          } else if (kind == 0) {                      // stance: one position shared by both nodes, zero velocity (:88-101)
            for (int dim = 0; dim < 3; ++dim) { h.var_of[nd * 6 + dim] = idx; h.var_of[(nd + 1) * 6 + dim] = idx; ++idx; }
            ++nd;
          } else {                                     // swing force: pinned to zero (:137-146)
            ++nd;
          }
        }
        h.n_var = idx;
      }
    }
    int off = 0, eoff = 0, poff = 0;
    d.max_polys = 0;
    for (int s = 0; s < N_SPLINES; ++s) {
      HostSpline& h = hs[s];
      h.var_off = off; off += h.n_var;
      SplineDesc& sd = d.sp[s];
      sd.n_nodes = h.n_nodes; sd.n_polys = h.n_polys; sd.n_var = h.n_var; sd.var_off = h.var_off;
      sd.node_off = eoff; sd.poly_off = poff; sd.phase_based = h.phase_based; sd.ee = h.ee;
      eoff += h.n_nodes * 6; poff += h.n_polys;
      d.max_polys = std::max(d.max_polys, h.n_polys);
    }
    d.n_nodesvars = off; d.tot_entries = eoff; d.tot_polys = poff;
    int phoff = 0;
    for (int e = 0; e < 4; ++e) { d.phase_off[e] = phoff; phoff += d.n_phase[e]; }
    d.tot_phases = phoff;

    // ---- constant pools: data
    d.o_data[0] = push_d(in.com, F * 3);
    d.o_data[1] = push_d(in.euler, F * 3);
    for (int e = 0; e < 4; ++e) d.o_data[2 + e] = push_d(ee_file[slot[e]], F * 3);
    d.o_hip[0] = push_d(in.hip_l, F * 3);
    d.o_hip[1] = push_d(in.hip_r, F * 3);
    d.o_inertia = push_d(in.inertia, F * 6);
    {  // DataCost sample times accumulate by += dt (data_cost.cpp:44-49)
      std::vector<double> tc(F + 2);
      double t = 0; for (int i = 0; i < F + 2; ++i) { tc[i] = t; t += in.dt; }
      d.o_tcost = push_d(tc);
    }
    tdyn_ = sample_times(T, 0.1);      // parameters.cpp:59 dt_constraint_dynamic_ (also dt_constraint_height_ :58)
    trom_ = sample_times(T, 0.08);     // parameters.cpp:57 dt_constraint_range_of_motion_
    d.n_tdyn = (int)tdyn_.size(); d.n_trom = (int)trom_.size();
    d.o_tdyn = push_d(tdyn_); d.o_trom = push_d(trom_);
    {
      std::vector<double> ph;
      for (int e = 0; e < 4; ++e) ph.insert(ph.end(), phase_in[e].begin(), phase_in[e].end());
      d.o_phase_dur0 = push_d(ph);
      d.o_phase_dur_in = d.o_phase_dur0;
    }
    // ---- initial node values (nlp_formulation.cpp:106-186, phys_optim.cpp:440-503)
    {
      std::vector<double> node0(d.tot_entries, 0.0);
      double lin0[3], linF[3], ang0[3], angF[3], v0[3] = {0, 0, 0}, vF[3] = {0, 0, 0};
      for (int k = 0; k < 3; ++k) {
        lin0[k] = in.com[k]; linF[k] = in.com[(F - 1) * 3 + k];
        ang0[k] = in.euler[k]; angF[k] = in.euler[(F - 1) * 3 + k];
        for (int j = 0; j < 5; ++j) {     // vel_avg_over_num = 5 (phys_optim.cpp:442-455, :470-479)
          v0[k] += (in.com[(j + 1) * 3 + k] - in.com[j * 3 + k]) / in.dt;
          vF[k] += (in.com[(F - 1 - j) * 3 + k] - in.com[(F - 2 - j) * 3 + k]) / in.dt;
        }
        v0[k] /= 5; vF[k] /= 5;
      }
      for (int b = 0; b < 2; ++b) {
        const HostSpline& h = hs[b];
        double* nv = &node0[d.sp[b].node_off];
        const double* a = b ? ang0 : lin0; const double* z = b ? angF : linF;
        for (int nd = 0; nd < h.n_nodes; ++nd)
          for (int k = 0; k < 3; ++k) {
            nv[nd * 6 + k] = a[k] + nd / (double)(h.n_nodes - 1) * (z[k] - a[k]);
            nv[nd * 6 + 3 + k] = (z[k] - a[k]) / T;
          }
        if (b == 0) for (int k = 0; k < 3; ++k) { nv[3 + k] = v0[k]; nv[(h.n_nodes - 1) * 6 + 3 + k] = vF[k]; }
      }
      for (int e = 0; e < 4; ++e) {
        const double* data = ee_file[slot[e]];
        const double tx = linF[0], ty = linF[1];
        double tz = (-in.normal[1] * (ty - in.point[1]) - in.normal[0] * (tx - in.point[0])) / in.normal[2] + in.point[2];   // ground_plane.cpp:18-27
        const double a[3] = {data[0], data[1], data[2]}, z[3] = {tx, ty, tz};
        {  // ee motion: straight line from the first frame to the terrain under the final base (nlp_formulation.cpp:148-156)
          const HostSpline& h = hs[2 + e];
          double* nv = &node0[d.sp[2 + e].node_off];
          for (int nd = 0; nd < h.n_nodes; ++nd)
            for (int k = 0; k < 3; ++k) {
              if (h.var_of[nd * 6 + k] >= 0) nv[nd * 6 + k] = a[k] + nd / (double)(h.n_nodes - 1) * (z[k] - a[k]);
              if (h.var_of[nd * 6 + 3 + k] >= 0) nv[nd * 6 + 3 + k] = (z[k] - a[k]) / T;
            }
          // a shared stance variable carries the value written last, i.e. that of the later node
          for (int p = 0; p < h.n_polys; ++p)
            if (h.isc[p]) for (int k = 0; k < 3; ++k) nv[p * 6 + k] = nv[(p + 1) * 6 + k];
        }
        {  // ee force: m g / 4 along +z on every optimised node (nlp_formulation.cpp:174-181)
          const HostSpline& h = hs[6 + e];
          double* nv = &node0[d.sp[6 + e].node_off];
          for (int nd = 0; nd < h.n_nodes; ++nd)
            if (h.var_of[nd * 6 + 2] >= 0) nv[nd * 6 + 2] = in.mass * kG / 4.0;
        }
      }
      d.o_node0 = push_d(node0);
    }
    // ---- constant pools: index maps
    {
      std::vector<int> vo, pinf, vnode(d.n_nodesvars, -1), vspl(d.n_nodesvars, 0);
      for (int s = 0; s < N_SPLINES; ++s) {
        const HostSpline& h = hs[s];
        vo.insert(vo.end(), h.var_of.begin(), h.var_of.end());
        for (int p = 0; p < h.n_polys; ++p) {
          if (h.phase_based) { pinf.push_back(h.ph[p]); pinf.push_back(h.kin[p]); pinf.push_back(h.nin[p]); pinf.push_back(h.isc[p]); }
          else { pinf.push_back(0); pinf.push_back(0); pinf.push_back(1); pinf.push_back(0); }
        }
        for (int k = 0; k < h.n_nodes * 6; ++k)
          if (h.var_of[k] >= 0 && vnode[h.var_off + h.var_of[k]] < 0) { vnode[h.var_off + h.var_of[k]] = k; vspl[h.var_off + h.var_of[k]] = s; }
      }
      d.o_varof = push_i(vo); d.o_pinfo = push_i(pinf); d.o_varnode = push_i(vnode); d.o_varspl = push_i(vspl);
    }

    // ---- stages
    std::vector<double> cur[4];
    for (int e = 0; e < 4; ++e) cur[e] = phase_in[e];
    int max_n = 0, max_m = 0;
    for (int st = 0; st < N_STAGES; ++st) {
      stage_m_cap[st] = -1;
      build_stage(st, cfg, cur, /*first_time=*/true);
      max_n = std::max(max_n, d.st[st].n);
      max_m = std::max(max_m, stage_m_cap[st]);
    }
    d.max_n = max_n; d.max_m = max_m; d.max_N = max_n + max_m;
    for (int st = 0; st < N_STAGES; ++st) { w_cap = std::max(w_cap, d.st[st].w); bc_cap = std::max(bc_cap, d.st[st].bc); }
    w_cap += 48;               // head-room for the stage-4 rebuild after the durations moved
    bc_cap += 8;
    N_cap = d.max_N;
    layout_workspace();
  }

  // Stage definition: which families / cost weights (phys_optim.cpp:544-749).
  static void stage_def(int stage, const chd_config& c, StageDesc& s) {
    const int kin = FAM_TERRAIN | FAM_ROM;      // Parameters::AddLegConstraints     parameters.cpp:78-82
    const int dyn = FAM_DYNAMIC | FAM_FORCE;    // Parameters::AddDynamicsConstraints parameters.cpp:95-98
    s.stage = stage; s.opt_dur = 0; s.w_dur = -1;
    switch (stage) {
      case 0: case 1:      // 1.1 (:544-581), 1.2 (:591-599)
        s.families = FAM_BASEACC | (stage == 1 ? (kin | FAM_HEELDIST) : 0);
        for (int k = 0; k < 3; ++k) { s.w_data[k] = 1.0; s.w_vel[k] = 0.1; s.w_acc[k] = -1; }
        break;
      case 2: case 3: case 5:   // 2.1 (:609-643), 2.2 (:648-656), 4 (:714-749)
        s.families = FAM_BASEACC | kin | dyn | FAM_HEELDIST | (stage != 2 ? FAM_HEIGHT : 0);
        s.w_data[0] = c.w_com_lin; s.w_data[1] = c.w_com_ang; s.w_data[2] = c.w_ee;
        s.w_vel[0] = 0.001; s.w_vel[1] = 0.001; s.w_vel[2] = c.w_smooth;
        for (int k = 0; k < 3; ++k) s.w_acc[k] = 0.0001;
        break;
      case 4:              // 3 (:666-711): durations become variables, no acceleration smoothing (:693)
        s.families = FAM_BASEACC | kin | dyn | FAM_HEIGHT | FAM_HEELDIST | FAM_TOTALTIME;
        s.w_data[0] = c.w_com_lin; s.w_data[1] = c.w_com_ang; s.w_data[2] = c.w_ee;
        s.w_vel[0] = 0.001; s.w_vel[1] = 0.001; s.w_vel[2] = c.w_smooth;
        for (int k = 0; k < 3; ++k) s.w_acc[k] = -1;
        s.w_dur = c.w_dur; s.opt_dur = 1;
        break;
    }
    s.max_iter = c.max_iter[stage];
  }

  // (Re)build the tables of one stage for the given phase durations.  On the first call the
  // regions are appended to the pools; later calls (stage 4 fallback with the durations that
  // stage 3 left) overwrite the same regions.
  void build_stage(int stage, const chd_config& cfg, const std::vector<double>* phase_dur, bool first_time) {
    StageDesc& S = d.st[stage];
    StageDesc keep = S;
    stage_def(stage, cfg, S);
    // polynomial durations / cumulative times in force
    std::vector<double> pd[N_SPLINES], pe[N_SPLINES], phe[N_EE];
    for (int s = 0; s < N_SPLINES; ++s) { pd[s] = poly_durations(s, phase_dur); pe[s] = cumulate(pd[s]); }
    for (int e = 0; e < 4; ++e) phe[e] = cumulate(phase_dur[e]);

    int n = d.n_nodesvars;
    S.n_dur = 0;
    for (int e = 0; e < 4; ++e) S.dur_off[e] = 0;
    if (S.opt_dur) for (int e = 0; e < 4; ++e) { S.dur_off[e] = n; n += d.n_phase[e] - 1; S.n_dur += d.n_phase[e] - 1; }
    S.n = n;

    // ---- rows / tasks, in the fixed family order
    std::vector<int> task; std::vector<double> task_t; std::vector<double> cl, cu;
    std::vector<std::vector<int>> sup;       // structural support (global variable ids) per row
    std::vector<std::vector<int>> sup_wide;  // same with +-1 polynomial slack (bandwidth head-room when durations move)
    auto add_task = [&](int type, int a, int b, double t) { task.push_back(type); task.push_back(a); task.push_back(b); task.push_back((int)cl.size()); task_t.push_back(t); };
    auto add_row = [&](double lo, double hi) { cl.push_back(lo); cu.push_back(hi); sup.emplace_back(); sup_wide.emplace_back(); };
    auto poly_vars = [&](int s, int poly, int dimmask, std::vector<int>& out) {
      const HostSpline& h = hs[s];
      for (int side = 0; side < 2; ++side)
        for (int q = 0; q < 6; ++q) {
          if (!((dimmask >> (q % 3)) & 1)) continue;
          int v = h.var_of[(poly + side) * 6 + q];
          if (v >= 0) out.push_back(h.var_off + v);
        }
    };
    auto at_time = [&](int s, double t, int dimmask, int row) {      // variables of the polynomial active at t
      int p = locate(pe[s], t);
      poly_vars(s, p, dimmask, sup[row]);
      int lo = p, hi = p;
      if (S.opt_dur && hs[s].phase_based) {     // a sample close to a junction may cross it when the durations move
        const double t0 = p > 0 ? pe[s][p - 1] : 0.0;
        if (t - t0 < kSlack && p > 0) lo = p - 1;
        if (pe[s][p] - t < kSlack && p + 1 < hs[s].n_polys) hi = p + 1;
      }
      for (int q = lo; q <= hi; ++q) poly_vars(s, q, dimmask, sup_wide[row]);
    };
    auto dur_at = [&](int e, double t, int row) {
      if (!S.opt_dur) return;
      int cur = locate(phe[e], t), nv = d.n_phase[e] - 1;
      for (int k = 0; k <= std::min(cur, nv - 1); ++k) sup[row].push_back(S.dur_off[e] + k);
      const int cur2 = (phe[e][cur] - t < kSlack) ? cur + 1 : cur;
      for (int k = 0; k <= std::min(cur2, nv - 1); ++k) sup_wide[row].push_back(S.dur_off[e] + k);
    };
    const int all = 7;
    if (S.families & FAM_BASEACC)      // TOWR SplineAccConstraint: acceleration continuity at base-spline junctions
      for (int b = 0; b < 2; ++b)
        for (int j = 0; j + 1 < hs[b].n_polys; ++j) {
          add_task(T_BASEACC, b, j, 0.0);
          for (int k = 0; k < 3; ++k) {
            add_row(0.0, 0.0);
            int r = (int)cl.size() - 1;
            poly_vars(b, j, 1 << k, sup[r]); poly_vars(b, j + 1, 1 << k, sup[r]);
            sup_wide[r] = sup[r];
          }
        }
    if (S.families & FAM_TERRAIN)      // TOWR TerrainConstraint (node 0 skipped; one row per stance polynomial)
      for (int e = 0; e < 4; ++e) {
        const HostSpline& h = hs[2 + e];
        for (int nd = 1; nd < h.n_nodes; ++nd) {
          if (nd < h.n_polys && h.isc[nd]) continue;     // first node of a stance polynomial: same variables as the second
          add_task(T_TERRAIN, e, nd, 0.0);
          const bool stance = h.const_node(nd);
          add_row(0.0, stance ? 0.0 : kBigBound);
          int r = (int)cl.size() - 1;
          const int vz = h.var_of[nd * 6 + 2], vx = h.var_of[nd * 6 + 0], vy = h.var_of[nd * 6 + 1];
          if (vz >= 0) sup[r].push_back(h.var_off + vz);
          if (vx >= 0 && d.hx != 0.0) sup[r].push_back(h.var_off + vx);
          if (vy >= 0 && d.hy != 0.0) sup[r].push_back(h.var_off + vy);
          sup_wide[r] = sup[r];
        }
      }
    if (S.families & FAM_ROM)          // LegLengthConstraint (leg_length_constraint.cpp:36-111)
      for (int e = 0; e < 4; ++e)
        for (int k = 0; k < d.n_trom; ++k) {
          add_task(T_ROM, e, k, trom_[k]);
          const double L = (e < 2) ? d.leg_len : d.heel_len;     // :21-27
          add_row(0.0, 0.5 * L * L);                               // :59
          int r = (int)cl.size() - 1;
          at_time(0, trom_[k], all, r); at_time(1, trom_[k], all, r); at_time(2 + e, trom_[k], all, r);
          dur_at(e, trom_[k], r);
        }
    S.heel_row0 = (S.families & FAM_HEELDIST) ? (int)cl.size() : -1;
    if (S.families & FAM_HEELDIST)     // EEDistConstraint (ee_dist_constraint.cpp:29-94), pairs (0,2), (1,3)
      for (int pr = 0; pr < 2; ++pr)
        for (int k = 0; k < d.n_trom; ++k) {
          add_task(T_HEELDIST, pr, k, trom_[k]);
          const double v = 0.5 * d.heel_dist * d.heel_dist;        // :39
          add_row(v, v);
          int r = (int)cl.size() - 1;
          at_time(2 + pr, trom_[k], all, r); at_time(4 + pr, trom_[k], all, r);
          dur_at(pr, trom_[k], r); dur_at(pr + 2, trom_[k], r);
        }
    S.dyn_first = (int)task_t.size(); S.n_dyn = (S.families & FAM_DYNAMIC) ? d.n_tdyn : 0;
    if (S.families & FAM_DYNAMIC)      // HumanoidDynamicConstraint (humanoid_dynamic_constraint.cpp:63-143)
      for (int k = 0; k < d.n_tdyn; ++k) {
        add_task(T_DYN, 0, k, tdyn_[k]);
        for (int q = 0; q < 6; ++q) {
          add_row(0.0, 0.0);
          int r = (int)cl.size() - 1;
          const int i = q % 3;
          if (q < 3) {     // angular rows
            at_time(0, tdyn_[k], all & ~(1 << i), r); at_time(1, tdyn_[k], all, r);
            for (int e = 0; e < 4; ++e) { at_time(6 + e, tdyn_[k], all & ~(1 << i), r); at_time(2 + e, tdyn_[k], all & ~(1 << i), r); dur_at(e, tdyn_[k], r); }
          } else {         // linear rows
            at_time(0, tdyn_[k], 1 << i, r);
            for (int e = 0; e < 4; ++e) { at_time(6 + e, tdyn_[k], 1 << i, r); dur_at(e, tdyn_[k], r); }
          }
        }
      }
    if (S.families & FAM_FORCE)        // TOWR ForceConstraint: 5 rows per optimised force node
      for (int e = 0; e < 4; ++e) {
        const HostSpline& h = hs[6 + e];
        for (int nd = 0; nd < h.n_nodes; ++nd) {
          if (h.const_node(nd)) continue;
          add_task(T_FORCE, e, nd, 0.0);
          const double lo[5] = {0.0, -kBigBound, 0.0, -kBigBound, 0.0};
          const double hi[5] = {kForceMax, 0.0, kBigBound, 0.0, kBigBound};
          for (int q = 0; q < 5; ++q) {
            add_row(lo[q], hi[q]);
            int r = (int)cl.size() - 1;
            double dir[3]; force_dir(q, dir);
            for (int k = 0; k < 3; ++k) { int v = h.var_of[nd * 6 + k]; if (v >= 0 && dir[k] != 0.0) sup[r].push_back(h.var_off + v); }
            sup_wide[r] = sup[r];
          }
        }
      }
    int n_height_cand = 0;
    if (S.families & FAM_HEIGHT)       // HeightConstraint (height_constraint.cpp:24-58); samples that sit on a stance
      for (int e = 0; e < 4; ++e) {    // node duplicate the terrain equality and are dropped
        const HostSpline& h = hs[2 + e];
        for (int k = 0; k < d.n_tdyn; ++k) {
          ++n_height_cand;
          const double t = tdyn_[k];
          const int p = locate(pe[2 + e], t);
          if (h.isc[p]) continue;
          const double tl = t - (p > 0 ? pe[2 + e][p - 1] : 0.0);
          double tl_seq = t; for (int i = 0; i < p; ++i) tl_seq -= pd[2 + e][i];
          (void)tl;
          if (tl_seq >= pd[2 + e][p] - 1e-9 && p + 1 < h.n_polys && h.isc[p + 1]) continue;
          if (tl_seq <= 1e-9 && p > 0 && h.isc[p - 1]) continue;
          add_task(T_HEIGHT, e, k, t);
          add_row(0.0, kBigBound);
          int r = (int)cl.size() - 1;
          int mask = (d.normal[0] != 0.0 ? 1 : 0) | (d.normal[1] != 0.0 ? 2 : 0) | (d.normal[2] != 0.0 ? 4 : 0);
          at_time(2 + e, t, mask, r); dur_at(e, t, r);
        }
      }
    if (S.families & FAM_TOTALTIME)    // ContactDurationConstraint (total_duration_constraint.cpp:60-82) + PhaseDurations bounds
      for (int e = 0; e < 4; ++e) {
        const int nv = d.n_phase[e] - 1;
        add_task(T_TOTALTIME, e, 0, 0.0);
        add_row(std::max(0.0, d.T - 500.0), d.T);                  // parameters.cpp:60
        int r = (int)cl.size() - 1;
        for (int k = 0; k < nv; ++k) sup[r].push_back(S.dur_off[e] + k);
        sup_wide[r] = sup[r];
        for (int k = 0; k < nv; ++k) {
          add_task(T_DURBOUND, e, k, 0.0);
          add_row(0.0, 500.0);
          int r2 = (int)cl.size() - 1;
          sup[r2].push_back(S.dur_off[e] + k); sup_wide[r2] = sup[r2];
        }
      }
    const int m = (int)cl.size();
    S.m = m; S.n_tasks = (int)task_t.size();
    S.nnz_jac = 0; for (auto& v : sup) S.nnz_jac += (int)v.size();

    // ---- KKT ordering: variables by node time, each row right after its last variable;
    //      long-range variables (shared stance positions, durations) and the rows that touch
    //      only those go to the border.
    std::vector<double> vtime(n, 0.0); std::vector<char> border(n, 0); std::vector<double> Dw(n, 1.0);
    const double fscale = d.mass * kG / 4.0;
    for (int s = 0; s < N_SPLINES; ++s) {
      const HostSpline& h = hs[s];
      std::vector<double> tn(h.n_nodes, 0.0);
      for (int k = 1; k < h.n_nodes; ++k) tn[k] = tn[k - 1] + pd[s][k - 1];
      for (int nd = 0; nd < h.n_nodes; ++nd)
        for (int q = 0; q < 6; ++q) {
          int v = h.var_of[nd * 6 + q];
          if (v < 0) continue;
          const int g = h.var_off + v;
          if (h.phase_based && s < 6 && h.const_node(nd)) {      // shared stance position: border; its time = start of the stance polynomial (first node of the pair)
            border[g] = 1;
            if (nd < h.n_polys && h.isc[nd]) vtime[g] = tn[nd];
          } else vtime[g] = tn[nd];
          if (s >= 6) Dw[g] = 1.0 / (fscale * fscale);
        }
    }
    for (int g = d.n_nodesvars; g < n; ++g) border[g] = 1;
    if (S.opt_dur)
      for (int e = 0; e < N_EE; ++e) {      // a duration variable: the start time of its phase
        double t0 = 0;
        for (int k = 0; k + 1 < d.n_phase[e]; ++k) { vtime[S.dur_off[e] + k] = t0; t0 += phase_dur[e][k]; }
      }
    std::vector<int> band;
    for (int j = 0; j < n; ++j) if (!border[j]) band.push_back(j);
    std::stable_sort(band.begin(), band.end(), [&](int a, int b) { return vtime[a] < vtime[b]; });
    std::vector<int> rank(n, -1);
    for (size_t r = 0; r < band.size(); ++r) rank[band[r]] = (int)r;
    std::vector<std::vector<int>> after(band.size());
    std::vector<int> brow;
    for (int i = 0; i < m; ++i) {
      int last = -1;
      for (int v : sup[i]) last = std::max(last, rank[v]);
      if (last >= 0) after[last].push_back(i); else brow.push_back(i);
    }
    std::vector<int> pos_var(n, -1), pos_row(m, -1);
    int Nb = 0;
    for (size_t r = 0; r < band.size(); ++r) { pos_var[band[r]] = Nb++; for (int i : after[r]) pos_row[i] = Nb++; }
    int bc = 0;
    for (int j = 0; j < n; ++j) if (border[j]) pos_var[j] = Nb + bc++;
    for (int i : brow) pos_row[i] = Nb + bc++;
    S.Nb = Nb; S.bc = bc;

    // ---- half-bandwidth: constraint rows and the Gauss-Newton couplings of the cost terms
    int w = 0;
    for (int i = 0; i < m; ++i) {
      if (pos_row[i] >= Nb) continue;
      for (int v : sup_wide[i]) if (pos_var[v] < Nb) w = std::max(w, std::abs(pos_row[i] - pos_var[v]));
    }
    for (int s = 0; s < 6; ++s) {
      const HostSpline& h = hs[s];
      const double* tc = &cd[d.o_tcost];
      for (int i = 0; i + 1 < d.F + 1; ++i) {
        int p0 = locate(pe[s], tc[i]), p1 = locate(pe[s], tc[std::min(i + 1, d.F)]);
        int lo = std::min(p0, p1), hi = std::max(p0, p1);
        if (S.opt_dur && h.phase_based) {
          const double t0 = lo > 0 ? pe[s][lo - 1] : 0.0;
          if (tc[i] - t0 < kSlack && lo > 0) --lo;
          if (pe[s][hi] - tc[std::min(i + 1, d.F)] < kSlack && hi + 1 < h.n_polys) ++hi;
        }
        for (int k = 0; k < 3; ++k) {
          int pmin = 1 << 30, pmax = -1;
          for (int nd = lo; nd <= hi + 1; ++nd)
            for (int dq = 0; dq < 2; ++dq) {
              int v = h.var_of[nd * 6 + dq * 3 + k];
              if (v < 0) continue;
              int p = pos_var[h.var_off + v];
              if (p >= Nb) continue;
              pmin = std::min(pmin, p); pmax = std::max(pmax, p);
            }
          if (pmax >= 0) w = std::max(w, pmax - pmin);
        }
      }
    }
    if (S.heel_row0 >= 0)      // exact curvature of the heel-distance rows: the row's variables couple with each other -- with the
      for (int i = S.heel_row0; i < S.heel_row0 + 2 * d.n_trom; ++i) {      // slack polynomials of stage 3 they can lie on both sides of the row
        int lo = 1 << 30, hi = -1;
        for (int v : sup_wide[i]) if (v < d.n_nodesvars && pos_var[v] < Nb) { lo = std::min(lo, pos_var[v]); hi = std::max(hi, pos_var[v]); }
        if (hi >= 0) w = std::max(w, hi - lo);
      }
    // ---- envelope of the banded part: efirst[p] / elast[p] = smallest / largest band position coupled to p.
    // L D L^T without pivoting keeps its fill inside this envelope, so the substitution, the mat-vec and the panel
    // row solves only visit [efirst[p], p] (and by symmetry up to elast[p]).
    std::vector<int> efirst(Nb), elast(Nb);
    for (int p = 0; p < Nb; ++p) { efirst[p] = p; elast[p] = p; }
    auto couple = [&](const std::vector<int>& pos) {       // all positions in `pos` are mutually coupled
      int lo = 1 << 30, hi = -1;
      for (int p : pos) if (p < Nb) { lo = std::min(lo, p); hi = std::max(hi, p); }
      if (hi < 0) return;
      for (int p : pos) if (p < Nb) { efirst[p] = std::min(efirst[p], lo); elast[p] = std::max(elast[p], hi); }
    };
    {
      std::vector<int> pos;
      for (int i = 0; i < m; ++i) {      // a row couples with each of its variables (not the variables with each other)
        if (pos_row[i] >= Nb) continue;
        for (int v : sup_wide[i]) { pos.assign({pos_row[i], pos_var[v]}); couple(pos); }
      }
      for (int s = 0; s < 6; ++s) {      // Gauss-Newton couplings of the cost terms, per dimension
        const HostSpline& h = hs[s];
        const double* tc = &cd[d.o_tcost];
        for (int i = 0; i + 1 < d.F + 1; ++i) {
          int p0 = locate(pe[s], tc[i]), p1 = locate(pe[s], tc[std::min(i + 1, d.F)]);
          int lo = std::min(p0, p1), hi = std::max(p0, p1);
          if (S.opt_dur && h.phase_based) {
            const double t0 = lo > 0 ? pe[s][lo - 1] : 0.0;
            if (tc[i] - t0 < kSlack && lo > 0) --lo;
            if (pe[s][hi] - tc[std::min(i + 1, d.F)] < kSlack && hi + 1 < h.n_polys) ++hi;
          }
          for (int k = 0; k < 3; ++k) {
            pos.clear();
            for (int nd = lo; nd <= hi + 1; ++nd)
              for (int dq = 0; dq < 2; ++dq) { int v = h.var_of[nd * 6 + dq * 3 + k]; if (v >= 0) pos.push_back(pos_var[h.var_off + v]); }
            couple(pos);
          }
        }
      }
      if (S.heel_row0 >= 0)              // exact curvature of the heel-distance rows: a row's toe and heel variables couple with each other
        for (int i = S.heel_row0; i < S.heel_row0 + 2 * d.n_trom; ++i) {
          pos.clear();
          for (int v : sup_wide[i]) if (v < d.n_nodesvars) pos.push_back(pos_var[v]);
          couple(pos);
        }
    }
    for (int p = 0; p < Nb; ++p) w = std::max(w, std::max(p - efirst[p], elast[p] - p));      // (safety net: no envelope wider than the band)
    // ---- first band position coupled to each border position (Nb = none): left of it the border row of the KKT matrix
    // and of its factor is structurally zero, so the row joins the factorisation only from that panel on
    std::vector<int> bfirst(bc, Nb);
    {
      auto reach = [&](int pb, int p) { if (pb >= Nb && p >= 0 && p < Nb) bfirst[pb - Nb] = std::min(bfirst[pb - Nb], p); };
      for (int i = 0; i < m; ++i)
        for (int v : sup_wide[i]) { reach(pos_var[v], pos_row[i]); reach(pos_row[i], pos_var[v]); }
      std::vector<int> pos;
      for (int s = 0; s < 6; ++s) {      // Gauss-Newton couplings of the cost terms (same sample windows as above)
        const HostSpline& h = hs[s];
        const double* tc = &cd[d.o_tcost];
        for (int i = 0; i + 1 < d.F + 1; ++i) {
          int p0 = locate(pe[s], tc[i]), p1 = locate(pe[s], tc[std::min(i + 1, d.F)]);
          int lo = std::min(p0, p1), hi = std::max(p0, p1);
          if (S.opt_dur && h.phase_based) {
            const double t0 = lo > 0 ? pe[s][lo - 1] : 0.0;
            if (tc[i] - t0 < kSlack && lo > 0) --lo;
            if (pe[s][hi] - tc[std::min(i + 1, d.F)] < kSlack && hi + 1 < h.n_polys) ++hi;
          }
          for (int k = 0; k < 3; ++k) {
            pos.clear();
            for (int nd = lo; nd <= hi + 1; ++nd)
              for (int dq = 0; dq < 2; ++dq) { int v = h.var_of[nd * 6 + dq * 3 + k]; if (v >= 0) pos.push_back(pos_var[h.var_off + v]); }
            int lob = Nb;
            for (int pp : pos) if (pp < Nb) lob = std::min(lob, pp);
            for (int pp : pos) reach(pp, lob < Nb ? lob : -1);
          }
        }
      }
      if (S.heel_row0 >= 0)              // (same couplings seen from the border: a stance position of one contact point and the other point's variables)
        for (int i = S.heel_row0; i < S.heel_row0 + 2 * d.n_trom; ++i) {
          int lob = Nb;
          for (int v : sup_wide[i]) if (v < d.n_nodesvars && pos_var[v] < Nb) lob = std::min(lob, pos_var[v]);
          for (int v : sup_wide[i]) if (v < d.n_nodesvars) reach(pos_var[v], lob < Nb ? lob : -1);
        }
      if (S.opt_dur)      // a duration moves every later sample of its end-effector: couples with all node values from its phase on
        for (int e = 0; e < N_EE; ++e) {
          const HostSpline& h = hs[2 + e];
          const int nv = d.n_phase[e] - 1;
          for (int k = 0; k < nv; ++k) {
            int pk = 0;
            while (pk < h.n_polys && h.ph[pk] < k) ++pk;
            for (int nd = pk; nd < h.n_nodes; ++nd)
              for (int q6 = 0; q6 < 6; ++q6) { int v = h.var_of[nd * 6 + q6]; if (v >= 0) reach(pos_var[S.dur_off[e] + k], pos_var[h.var_off + v]); }
            // exact node x duration block of the constraint rows (chd_device.hpp, XB_*): the duration also couples with the force nodes of its end-effector
            // (dynamics rows) and, from the start of its phase on, with the nodes of the foot's other contact point (heel-distance rows) and of the two
            // base splines (leg-length and dynamics rows)
            {
              const HostSpline& hf = hs[6 + e];
              int pf = 0;
              while (pf < hf.n_polys && hf.ph[pf] < k) ++pf;
              for (int nd = pf; nd < hf.n_nodes; ++nd)
                for (int q6 = 0; q6 < 6; ++q6) { int v = hf.var_of[nd * 6 + q6]; if (v >= 0) reach(pos_var[S.dur_off[e] + k], pos_var[hf.var_off + v]); }
              const double t0 = std::max(0.0, (k > 0 ? phe[e][k - 1] : 0.0) - kSlack);
              const int others[3] = {0, 1, 2 + (e + 2) % 4};
              for (int oi = 0; oi < 3; ++oi) {
                const HostSpline& ho = hs[others[oi]];
                for (int nd = locate(pe[others[oi]], t0); nd < ho.n_nodes; ++nd)
                  for (int q6 = 0; q6 < 6; ++q6) { int v = ho.var_of[nd * 6 + q6]; if (v >= 0) reach(pos_var[S.dur_off[e] + k], pos_var[ho.var_off + v]); }
              }
            }
          }
        }
    }
    // ---- order of the border variables: by the start time of their phase (stance position: its stance phase, duration: its own phase), ties by index.
    // The oracle uses the same rule (ipm_solver.hpp): the positional pivot test of the factorisation depends on the elimination order.  The border rows that
    // reach a band column k are then (almost) a prefix [0, rcnt[k]) of the border, which the column passes of the substitution / mat-vec stop at -- a row inside
    // the prefix that does not reach the column holds zeros there.
    {
      std::vector<int> bvars;
      for (int j = 0; j < n; ++j) if (pos_var[j] >= Nb) bvars.push_back(j);
      std::stable_sort(bvars.begin(), bvars.end(), [&](int a, int b) { return vtime[a] < vtime[b]; });
      std::vector<int> nb_first(bc, Nb);
      std::vector<int> newpos(n, -1);
      for (size_t r = 0; r < bvars.size(); ++r) { newpos[bvars[r]] = Nb + (int)r; nb_first[r] = bfirst[pos_var[bvars[r]] - Nb]; }
      for (int i = 0; i < m; ++i) if (pos_row[i] >= Nb) nb_first[pos_row[i] - Nb] = bfirst[pos_row[i] - Nb];      // border rows keep their places (after the variables)
      for (int j : bvars) pos_var[j] = newpos[j];
      bfirst = nb_first;
    }
    std::vector<int> rcnt(Nb, 0);
    for (int r = 0; r < bc; ++r) if (bfirst[r] < Nb) for (int k = bfirst[r]; k < Nb; ++k) rcnt[k] = std::max(rcnt[k], r + 1);
    S.w = w; S.valid = 1;

    // ---- store
    if (first_time) {
      int mcap = m, tcap = S.n_tasks;
      if (stage == 5) { int extra = n_height_cand; mcap = m + extra; tcap = S.n_tasks + extra; }   // durations may have moved
      stage_m_cap[stage] = mcap; stage_task_cap[stage] = tcap;
      S.o_pos_var = reserve_i(n); S.o_pos_row = reserve_i(mcap); S.o_task = reserve_i(4 * tcap); S.o_env = reserve_i(2 * (n + mcap)); S.o_rcnt = reserve_i(n + mcap);
      S.o_cl = reserve_d(mcap); S.o_cu = reserve_d(mcap); S.o_Dw = reserve_d(n); S.o_task_t = reserve_d(tcap);
    } else {
      S.o_pos_var = keep.o_pos_var; S.o_pos_row = keep.o_pos_row; S.o_task = keep.o_task; S.o_env = keep.o_env; S.o_rcnt = keep.o_rcnt;
      S.o_cl = keep.o_cl; S.o_cu = keep.o_cu; S.o_Dw = keep.o_Dw; S.o_task_t = keep.o_task_t;
      if (m > stage_m_cap[stage] || S.n_tasks > stage_task_cap[stage] || w > w_cap || bc > bc_cap || n + m > N_cap) S.valid = 0;
    }
    if (S.valid) {
      std::copy(pos_var.begin(), pos_var.end(), ci.begin() + S.o_pos_var);
      std::copy(pos_row.begin(), pos_row.end(), ci.begin() + S.o_pos_row);
      std::copy(task.begin(), task.end(), ci.begin() + S.o_task);
      {   // second entry: last row whose envelope reaches column p (covers the fill-in of the factor as well)
        std::vector<int> clast(Nb);
        for (int p = 0; p < Nb; ++p) clast[p] = elast[p];
        for (int i = 0; i < Nb; ++i) for (int k = efirst[i]; k <= i; ++k) clast[k] = std::max(clast[k], i);
        for (int p = 0; p < Nb; ++p) { ci[S.o_env + 2 * p] = efirst[p]; ci[S.o_env + 2 * p + 1] = clast[p]; }
        for (int r = 0; r < bc; ++r) { ci[S.o_env + 2 * (Nb + r)] = bfirst[r]; ci[S.o_env + 2 * (Nb + r) + 1] = Nb; }
        std::copy(rcnt.begin(), rcnt.end(), ci.begin() + S.o_rcnt);
      }
      std::copy(cl.begin(), cl.end(), cd.begin() + S.o_cl);
      std::copy(cu.begin(), cu.end(), cd.begin() + S.o_cu);
      std::copy(Dw.begin(), Dw.end(), cd.begin() + S.o_Dw);
      std::copy(task_t.begin(), task_t.end(), cd.begin() + S.o_task_t);
    }
    // SURVEY 8(d) algorithmic bytes of one interior-point iteration of this stage
    {
      const double hb = w, L = 6;
      alg_bytes_iter[stage] = 8.0 * (3.0 * n + 2.0 * m + 2.0 * S.nnz_jac + 2.0 * m * (hb + 1) + 4.0 * L * n) + 8.0 * d.F * 30.0;
    }
  }

  void force_dir(int q, double* dir) const {
    for (int k = 0; k < 3; ++k) {
      switch (q) {
        case 0: dir[k] = d.bn[k]; break;
        case 1: dir[k] = d.bt1[k] - kMu * d.bn[k]; break;
        case 2: dir[k] = d.bt1[k] + kMu * d.bn[k]; break;
        case 3: dir[k] = d.bt2[k] - kMu * d.bn[k]; break;
        default: dir[k] = d.bt2[k] + kMu * d.bn[k]; break;
      }
    }
  }

  // ---- workspace layout ---------------------------------------------------------
  enum { NV_N = 10, NV_M = 24, NV_NN = 9 };    // number of n-, m- and N-sized solver vectors (chd_kernels.hpp)
  void layout_workspace() {
    long long o = 0;
    auto take = [&](long long cnt) { long long r = o; o += (cnt + 1) & ~1LL; return (int)r; };
    d.o_node = take(d.tot_entries);
    d.o_poly_dur = take(d.tot_polys);
    d.o_pend = take(d.tot_polys);
    d.o_phase_dur = take(d.tot_phases);
    d.o_phend = take(d.tot_phases);
    d.o_ttot = take(16);
    d.o_vec_n = take((long long)NV_N * d.max_n);
    d.o_vec_m = take((long long)NV_M * d.max_m);
    d.o_vec_N = take((long long)NV_NN * d.max_N);
    d.o_scache = take(6LL * (d.F + 2) * SC_STRIDE);
    d.d2_slots = 2 * d.n_tdyn + 2 * d.n_trom + d.F + 2;      // DYN, HEIGHT, ROM, HEELDIST, cost samples
    d.o_d2tab = take(4LL * d.d2_slots * D2_STRIDE);
    d.o_x2tab = take(2LL * d.n_trom * X2_STRIDE);
    d.o_rcache = take(4LL * d.n_trom * RC_STRIDE);
    d.o_xtab = take(4LL * (4LL * d.n_tdyn + 5LL * d.n_trom) * XR_STRIDE);      // node x duration records (chd_device.hpp: XB_*)
    d.side_cap = 24 * d.max_n + 4096;
    d.o_side_v = take(d.side_cap);
    const long long Nb_cap = N_cap, W2 = 2LL * w_cap + 1, LD = N_cap;
    d.o_pmb = take(Nb_cap * ((W2 + 63) / 64)); d.o_pmx = take((long long)bc_cap * ((LD + 63) / 64)); d.o_pmt = take(Nb_cap * ((bc_cap + 63) / 64));
    d.sz_K0b = Nb_cap * W2; d.sz_K0x = (long long)bc_cap * LD;
    d.sz_Kfb = Nb_cap * (w_cap + 1); d.sz_Kfx = (long long)bc_cap * LD;
    long long big = o;
    auto take_big = [&](long long cnt) { long long r = big; big += (cnt + 1) & ~1LL; return r; };
    long long k0b = take_big(d.sz_K0b), k0x = take_big(d.sz_K0x), kfb = take_big(d.sz_Kfb), kfx = take_big(d.sz_Kfx);
    if (big >= (1LL << 31)) throw std::runtime_error("sequence too large: KKT workspace exceeds 2^31 doubles");
    d.o_K0b = (int)k0b; d.o_K0x = (int)k0x; d.o_Kfb = (int)kfb; d.o_Kfx = (int)kfx;
    wd_size = big;
    long long oi = 0;
    auto take_i = [&](long long cnt) { long long r = oi; oi += cnt; return (int)r; };
    d.o_flags = take_i(d.max_m);
    d.o_first = take_i(6LL * (d.max_polys + 2));
    d.o_sign = take_i(d.max_N);
    d.o_envw = take_i(2LL * d.max_N);
    d.o_rcntw = take_i(d.max_N);
    d.csr_cap = (int)std::min<long long>(48LL * N_cap + (long long)bc_cap * N_cap, 1LL << 28);      // band rows ~14 entries each; the border up to half dense, listed in both orientations
    d.o_side_pq = take_i(2LL * d.side_cap);
    d.o_csr_rp = take_i(d.max_N + 2); d.o_csr_col = take_i(d.csr_cap); d.o_csr_row = take_i(d.csr_cap);
    wi_size = oi;
  }

 private:
  std::vector<double> base_dur_, tdyn_, trom_;
};

}  // namespace chd

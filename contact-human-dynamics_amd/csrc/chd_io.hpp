// chd_io.hpp — native text-file I/O at the drop-in boundary.
//   inputs : phys_optim_in_<char>/{skel,motion,terrain,contact}_info.txt, token semantics of the
//            reference readers (operator>> on whitespace-separated tokens, phys_optim.cpp:155-267)
//   outputs: sol_out_*.txt with the line layout / 10-significant-digit formatting of SaveSolution
//            (phys_optim.cpp:63-143) and success_log.txt (phys_optim.cpp:145-153)
#pragma once
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <iterator>
#include <sstream>
#include <string>
#include <vector>

#include "../../include/chd_phys.h"

namespace chd {
namespace io {

struct SeqFiles {
  int F = 0;
  double dt = 0;
  std::vector<double> hip_l, hip_r, inertia, blocks[6];   // blocks: com, euler, ltoe, lheel, rtoe, rheel
  double scal[4] = {0, 0, 0, 0};                           // leg_len, heel_len, heel_dist, mass
  double normal[3] = {0, 0, 1}, point[3] = {0, 0, 0};
  int start[4] = {0, 0, 0, 0};
  std::vector<double> dur[4];
  void fill(chd_seq_in& s) const {
    s.F = F; s.dt = dt; s.hip_l = hip_l.data(); s.hip_r = hip_r.data();
    s.leg_len = scal[0]; s.heel_len = scal[1]; s.heel_dist = scal[2]; s.mass = scal[3];
    s.inertia = inertia.data(); s.com = blocks[0].data(); s.euler = blocks[1].data();
    s.ltoe = blocks[2].data(); s.lheel = blocks[3].data(); s.rtoe = blocks[4].data(); s.rheel = blocks[5].data();
    for (int k = 0; k < 3; ++k) { s.normal[k] = normal[k]; s.point[k] = point[k]; }
    for (int e = 0; e < 4; ++e) { s.start_contact[e] = start[e]; s.n_phases[e] = (int)dur[e].size(); s.durations[e] = dur[e].data(); }
  }
};

inline bool read_tokens(const std::string& path, std::vector<std::string>& tok, std::string& err) {
  std::ifstream f(path);
  if (!f.good()) { err = "cannot open " + path; return false; }
  tok.assign(std::istream_iterator<std::string>(f), std::istream_iterator<std::string>());
  return true;
}
inline bool to_doubles(const std::vector<std::string>& tok, size_t from, size_t count, std::vector<double>& out, std::string& err, const std::string& name) {
  if (from + count > tok.size()) { err = name + ": expected " + std::to_string(from + count) + " tokens, found " + std::to_string(tok.size()); return false; }
  out.resize(count);
  for (size_t i = 0; i < count; ++i) {
    char* end = nullptr;
    out[i] = std::strtod(tok[from + i].c_str(), &end);
    if (end == tok[from + i].c_str()) { err = name + ": bad number '" + tok[from + i] + "'"; return false; }
  }
  return true;
}

inline bool read_inputs(const std::string& dir, int nframes, SeqFiles& s, std::string& err) {
  const size_t F = (size_t)nframes;
  s.F = nframes;
  std::vector<std::string> t;
  std::vector<double> v;
  if (!read_tokens(dir + "/skel_info.txt", t, err)) return false;          // ReadSkeletonInfo phys_optim.cpp:169-188
  if (!to_doubles(t, 0, 3 * F, s.hip_l, err, "skel_info.txt") || !to_doubles(t, 3 * F, 3 * F, s.hip_r, err, "skel_info.txt")) return false;
  if (!to_doubles(t, 6 * F, 4, v, err, "skel_info.txt")) return false;
  for (int k = 0; k < 4; ++k) s.scal[k] = v[k];
  if (!to_doubles(t, 6 * F + 4, 6 * F, s.inertia, err, "skel_info.txt")) return false;
  if (!read_tokens(dir + "/motion_info.txt", t, err)) return false;        // ReadMotionInfo :190-208
  if (!to_doubles(t, 0, 1, v, err, "motion_info.txt")) return false;
  s.dt = v[0];
  for (int b = 0; b < 6; ++b) if (!to_doubles(t, 1 + b * 3 * F, 3 * F, s.blocks[b], err, "motion_info.txt")) return false;
  if (!read_tokens(dir + "/terrain_info.txt", t, err)) return false;       // ReadTerrainInfo :210-224
  if (!to_doubles(t, 0, 6, v, err, "terrain_info.txt")) return false;
  for (int k = 0; k < 3; ++k) { s.normal[k] = v[k]; s.point[k] = v[3 + k]; }
  if (!read_tokens(dir + "/contact_info.txt", t, err)) return false;       // ReadContactInfo :226-267
  size_t pos = 0;
  for (int e = 0; e < 4; ++e) {
    if (pos + 2 > t.size()) { err = "contact_info.txt: truncated"; return false; }
    if (t[pos] != "0" && t[pos] != "1") { err = "contact_info.txt: start flag must be 0 or 1 (operator>> into bool)"; return false; }
    s.start[e] = t[pos] == "1"; ++pos;
    const int n = std::atoi(t[pos].c_str()); ++pos;
    if (n < 1) { err = "contact_info.txt: phase count < 1"; return false; }
    if (!to_doubles(t, pos, (size_t)n, s.dur[e], err, "contact_info.txt")) return false;
    pos += (size_t)n;
  }
  return true;
}

// storage behind the caller-allocated arrays of chd_seq_out
struct SnapStore {
  std::vector<double> d[CHD_N_SNAPSHOTS][4];
  std::vector<unsigned char> c[CHD_N_SNAPSHOTS];
  void bind(chd_seq_out& o, int cap) {
    for (int s = 0; s < CHD_N_SNAPSHOTS; ++s) {
      d[s][0].assign((size_t)cap * 3, 0.0); d[s][1].assign((size_t)cap * 3, 0.0);
      d[s][2].assign((size_t)4 * cap * 3, 0.0); d[s][3].assign((size_t)4 * cap * 3, 0.0);
      c[s].assign((size_t)4 * cap, 0);
      chd_snapshot& sn = o.snap[s];
      sn.capacity = cap; sn.n_samples = 0; sn.num_frames_header = 0;
      sn.base_lin = d[s][0].data(); sn.base_ang_deg = d[s][1].data(); sn.ee_pos = d[s][2].data(); sn.ee_force = d[s][3].data();
      sn.contact = c[s].data();
    }
  }
};

inline void put_line(std::FILE* f, const double* v, int n) {       // ofstream precision(10), default float field == %.10g
  for (int i = 0; i < n; ++i) std::fprintf(f, i ? " %.10g" : "%.10g", v[i]);
  std::fputc('\n', f);
}

inline bool write_solution(const std::string& path, double dt, const chd_snapshot& sn, std::string& err) {
  std::FILE* f = std::fopen(path.c_str(), "w");
  if (!f) { err = "cannot write " + path; return false; }
  const int ns = sn.n_samples < sn.capacity ? sn.n_samples : sn.capacity;
  std::fprintf(f, "dt\n%.10g\nnum_frames\n%d\nnum_feet\n%d\n", dt, sn.num_frames_header, CHD_N_EE);
  std::fprintf(f, "base_lin\n"); put_line(f, sn.base_lin, ns * 3);
  std::fprintf(f, "base_ang\n"); put_line(f, sn.base_ang_deg, ns * 3);
  for (int e = 0; e < CHD_N_EE; ++e) { std::fprintf(f, "foot%d_pos\n", e); put_line(f, sn.ee_pos + (size_t)e * sn.capacity * 3, ns * 3); }
  for (int e = 0; e < CHD_N_EE; ++e) { std::fprintf(f, "foot%d_force\n", e); put_line(f, sn.ee_force + (size_t)e * sn.capacity * 3, ns * 3); }
  for (int e = 0; e < CHD_N_EE; ++e) {
    std::fprintf(f, "foot%d_contact\n", e);
    for (int i = 0; i < ns; ++i) std::fprintf(f, i ? " %d" : "%d", (int)sn.contact[(size_t)e * sn.capacity + i]);
    std::fputc('\n', f);
  }
  std::fclose(f);
  return true;
}

inline bool write_outputs(const std::string& dir, double dt, const chd_seq_out& o, std::string& err) {
  static const char* names[CHD_N_SNAPSHOTS] = {"sol_out_no_dynamics.txt", "sol_out_dynamics.txt", "sol_out_durations.txt"};   // phys_optim.cpp:601, :659, :757
  for (int s = 0; s < CHD_N_SNAPSHOTS; ++s) if (!write_solution(dir + "/" + names[s], dt, o.snap[s], err)) return false;
  std::FILE* f = std::fopen((dir + "/success_log.txt").c_str(), "w");
  if (!f) { err = "cannot write " + dir + "/success_log.txt"; return false; }
  std::fprintf(f, "dynamics %d\ndurations %d\n", o.dynamics_succeed ? 1 : 0, o.durations_succeed ? 1 : 0);
  std::fclose(f);
  return true;
}

}  // namespace io
}  // namespace chd

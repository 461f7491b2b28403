// Host build of the native text-file I/O of the drop-in boundary (contact-human-dynamics_amd/csrc/chd_io.hpp is plain
// C++ and is compiled into libchd_phys.so unchanged): lets the CPU-only tests drive the reader and the writer that
// `chd_phys_solve_dirs` uses on the GPU box.  Test infrastructure.
#include <cstring>

#include "../../contact-human-dynamics_amd/csrc/chd_io.hpp"

using namespace chd::io;

static void put_err(const std::string& e, char* err, int errlen) {
  if (err && errlen > 0) { std::strncpy(err, e.c_str(), (size_t)errlen - 1); err[errlen - 1] = 0; }
}

extern "C" {

// reals: hip_l (3F) hip_r (3F) inertia (6F) six motion blocks (18F) | dt leg_len heel_len heel_dist mass normal(3) point(3) |
//        durations of the four end effectors (file order), back to back, at most 256
// ints : start[4], n_phases[4]
int io_emu_read(const char* dir, int F, double* reals, int* ints, char* err, int errlen) {
  SeqFiles s; std::string e;
  if (!read_inputs(dir, F, s, e)) { put_err(e, err, errlen); return 1; }
  chd_seq_in in;
  s.fill(in);
  double* p = reals;
  auto put = [&](const double* v, size_t n) { std::memcpy(p, v, n * sizeof(double)); p += n; };
  put(in.hip_l, 3 * (size_t)F); put(in.hip_r, 3 * (size_t)F); put(in.inertia, 6 * (size_t)F);
  for (const double* b : {in.com, in.euler, in.ltoe, in.lheel, in.rtoe, in.rheel}) put(b, 3 * (size_t)F);
  const double sc[5] = {in.dt, in.leg_len, in.heel_len, in.heel_dist, in.mass};
  put(sc, 5); put(in.normal, 3); put(in.point, 3);
  size_t nd = 0;
  for (int k = 0; k < 4; ++k) {
    ints[k] = in.start_contact[k]; ints[4 + k] = in.n_phases[k];
    if (nd + (size_t)in.n_phases[k] > 256) { put_err("too many phases for the test buffer", err, errlen); return 2; }
    put(in.durations[k], (size_t)in.n_phases[k]); nd += (size_t)in.n_phases[k];
  }
  return in.F == F ? 0 : 3;
}

// writes the three sol_out_*.txt (all from the same arrays) and success_log.txt into `dir`
int io_emu_write(const char* dir, double dt, int capacity, int n_samples, int header, double* base_lin, double* base_ang_deg,
                 double* ee_pos, double* ee_force, unsigned char* contact, int dyn_ok, int dur_ok, char* err, int errlen) {
  chd_seq_out o;
  std::memset(&o, 0, sizeof(o));
  for (int k = 0; k < CHD_N_SNAPSHOTS; ++k) {
    chd_snapshot& sn = o.snap[k];
    sn.capacity = capacity; sn.n_samples = n_samples; sn.num_frames_header = header;
    sn.base_lin = base_lin; sn.base_ang_deg = base_ang_deg; sn.ee_pos = ee_pos; sn.ee_force = ee_force; sn.contact = contact;
  }
  o.dynamics_succeed = dyn_ok; o.durations_succeed = dur_ok;
  std::string e;
  if (!write_outputs(dir, dt, o, e)) { put_err(e, err, errlen); return 1; }
  return 0;
}

}

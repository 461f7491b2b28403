// phys_optim — command-line drop-in for the reference's per-video child process.
//
// The reference driver runs (scripts/run_phys_mocap.py:159-174)
//     os.chdir(<--towr_phys_optim_path>); subprocess.run(['./phys_optim', '--in_dir', D_in, '--nframes', F, '--out_dir', D_out,
//                    '--w_com_lin', a, '--w_com_ang', b, '--w_ee', c, '--w_smooth', d, '--w_dur', e])
// and ignores the return code.  This executable accepts exactly the gflags of towr_phys_optim/phys_optim.cpp:23-31 (both
// "--flag value" and "--flag=value", one or two dashes), reads the four input files, solves the sequence on the MI355X
// through libchd_phys.so (chd_phys_solve_dirs with B = 1) and writes sol_out_{no_dynamics,dynamics,durations}.txt and
// success_log.txt into --out_dir, so the unmodified driver works with --towr_phys_optim_path pointing at this directory.
// One sequence per process leaves 255 of 256 compute units idle: the batched entry points (include/chd_phys.h,
// INTEGRATION.md) are the intended interface; this wrapper exists for compatibility.  There is no CPU path: without a
// HIP device the process exits with status 2.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>

#include "../../include/chd_phys.h"

static void usage() {
  std::fprintf(stderr,
               "usage: phys_optim --in_dir DIR --nframes N --out_dir DIR [--w_com_lin 0.4] [--w_com_ang 1.7] [--w_ee 0.3] [--w_smooth 0.1] [--w_dur 0.1]\n"
               "       (flags of the reference's towr_phys_optim/phys_optim.cpp:23-31; extra: --device ID, --stall_window N)\n");
}

int main(int argc, char** argv) {
  std::string in_dir = "./", out_dir = "sol_out";       // gflags defaults, phys_optim.cpp:23-25
  int nframes = 100, device = 0, check_args = 0;
  chd_config cfg;
  chd_config_default(&cfg);
  for (int i = 1; i < argc; ++i) {
    std::string a = argv[i];
    if (a == "-h" || a == "--help" || a == "-help") { usage(); return 0; }
    if (a.size() < 2 || a[0] != '-') { std::fprintf(stderr, "phys_optim: unexpected argument '%s'\n", a.c_str()); usage(); return 1; }
    a.erase(0, a[1] == '-' ? 2 : 1);
    std::string val;
    bool has_val = false;
    const size_t eq = a.find('=');
    if (eq != std::string::npos) { val = a.substr(eq + 1); a.erase(eq); has_val = true; }
    if (a == "check_args") { check_args = 1; continue; }
    if (!has_val) {
      if (i + 1 >= argc) { std::fprintf(stderr, "phys_optim: flag --%s needs a value\n", a.c_str()); return 1; }
      val = argv[++i];
    }
    char* end = nullptr;
    auto num = [&](double& dst) { dst = std::strtod(val.c_str(), &end); return end != val.c_str() && *end == 0; };
    bool ok = true;
    double t = 0;
    if (a == "in_dir") in_dir = val;
    else if (a == "out_dir") out_dir = val;
    else if (a == "nframes") { ok = num(t); nframes = (int)t; }
    else if (a == "w_com_lin") ok = num(cfg.w_com_lin);
    else if (a == "w_com_ang") ok = num(cfg.w_com_ang);
    else if (a == "w_ee") ok = num(cfg.w_ee);
    else if (a == "w_smooth") ok = num(cfg.w_smooth);
    else if (a == "w_dur") ok = num(cfg.w_dur);
    else if (a == "device") { ok = num(t); device = (int)t; }
    else if (a == "stall_window") { ok = num(t); cfg.stall_window = (int)t; }
    else { std::fprintf(stderr, "phys_optim: unknown flag --%s\n", a.c_str()); usage(); return 1; }      // gflags: "ERROR: unknown command line flag"
    if (!ok) { std::fprintf(stderr, "phys_optim: bad value '%s' for --%s\n", val.c_str(), a.c_str()); return 1; }
  }
  if (check_args) {
    std::printf("{\"in_dir\": \"%s\", \"out_dir\": \"%s\", \"nframes\": %d, \"w_com_lin\": %.17g, \"w_com_ang\": %.17g, \"w_ee\": %.17g, \"w_smooth\": %.17g, "
                "\"w_dur\": %.17g, \"device\": %d, \"stall_window\": %d}\n",
                in_dir.c_str(), out_dir.c_str(), nframes, cfg.w_com_lin, cfg.w_com_ang, cfg.w_ee, cfg.w_smooth, cfg.w_dur, device, cfg.stall_window);
    return 0;
  }
  if (chd_phys_version() != CHD_PHYS_ABI_VERSION) {
    std::fprintf(stderr, "phys_optim: libchd_phys.so has ABI version %d, this executable was built against %d\n", chd_phys_version(), CHD_PHYS_ABI_VERSION);
    return 2;
  }
  chd_handle* h = nullptr;
  if (chd_phys_create(&cfg, device, &h) != 0) {
    std::fprintf(stderr, "phys_optim: no usable HIP device %d (this build has no CPU path)\n", device);
    return 2;
  }
  const char* ind = in_dir.c_str();
  const char* outd = out_dir.c_str();
  int status = 0;
  const int rc = chd_phys_solve_dirs(h, 1, &ind, &outd, &nframes, &status);
  if (rc != 0 || status != 0) {
    std::fprintf(stderr, "phys_optim: %s (status %d)\n", chd_phys_last_error(h), status);
    chd_phys_destroy(h);
    return 3;
  }
  std::printf("Saving final solution...\n");      // phys_optim.cpp:755
  chd_phys_destroy(h);
  return 0;
}

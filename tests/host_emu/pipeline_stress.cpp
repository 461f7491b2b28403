// pipeline_stress.cpp — the HOST side of libchd_phys.so under ThreadSanitizer / AddressSanitizer (TEST INFRASTRUCTURE; VERDICT r04 item 9, SURVEY 5 "sanitizers").
//
// Compiles contact-human-dynamics_amd/csrc/chd_phys.hip itself as C++ (CHD_HOST_EMU: the kernel source as a host function; tests/host_emu/hip_stub.hpp: streams
// are threads, a launch starts one thread per resident workgroup), so what runs is the library's own solve_pipelined -- chunk plan, table builder threads,
// lanes with their page-locked staging, finisher threads, stage-4 fallback launches, workspace slots claimed by compare-and-swap, workspace growth while
// launches are in flight -- with real concurrency, and the sanitizer watching.  tests/test_sanitizers.py builds this twice (-fsanitize=thread, -fsanitize=address)
// and runs it on directories of short synthetic sequences.
//
//   pipeline_stress <list file: one "in_dir out_root nframes" per line> <max_iter> <plan> [<plan> ...]
//      plan = pipeline_chunk[:max_workgroups]   (chunk 0 = automatic, < 0 = one chunk)
// For every plan: a fresh handle, chd_phys_solve_dirs over all directories into out_root/plan<k>/, then -- on the same, warm handle -- the first half again
// through chd_phys_solve_batch.  Exit code 0 = every call succeeded and the warm call reproduced the first one's statistics.
#define CHD_HOST_EMU 1
#define CHD_HOST_EMU_HIP_STUB "../../tests/host_emu/hip_stub.hpp"
#include "../../contact-human-dynamics_amd/csrc/chd_phys.hip"

#include <fstream>
#include <sstream>
#include <sys/stat.h>

int main(int argc, char** argv) {
  if (argc < 4) { std::fprintf(stderr, "usage: pipeline_stress <list> <max_iter> <plan> ...\n"); return 2; }
  std::vector<std::string> in_dirs, out_roots; std::vector<int> nframes;
  {
    std::ifstream f(argv[1]);
    std::string a, b; int n;
    while (f >> a >> b >> n) { in_dirs.push_back(a); out_roots.push_back(b); nframes.push_back(n); }
  }
  const int B = (int)in_dirs.size();
  if (B == 0) { std::fprintf(stderr, "empty list\n"); return 2; }
  const int max_iter = std::atoi(argv[2]);
  int rc_all = 0;
  for (int p = 3; p < argc; ++p) {
    int chunk = 0, maxwg = 0;
    { std::string s = argv[p]; const size_t c = s.find(':'); chunk = std::atoi(s.substr(0, c).c_str()); if (c != std::string::npos) maxwg = std::atoi(s.substr(c + 1).c_str()); }
    chd_config cfg; chd_config_default(&cfg);
    for (int i = 0; i < CHD_N_STAGES; ++i) cfg.max_iter[i] = max_iter;
    cfg.max_iter[4] = max_iter < 4 ? max_iter : 4;          // stage 3 capped low: most sequences take the stage-4 fallback launch (the finishers' second launch)
    cfg.pipeline_chunk = chunk; cfg.max_workgroups = maxwg;
    chd_handle* h = nullptr;
    if (chd_phys_create(&cfg, 0, &h) != 0) { std::fprintf(stderr, "create failed\n"); return 3; }
    std::vector<std::string> outs(B);
    std::vector<const char*> ip(B), op(B);
    for (int i = 0; i < B; ++i) {
      outs[i] = out_roots[i] + "/plan" + std::to_string(p - 3);
      mkdir(outs[i].c_str(), 0777);
      ip[i] = in_dirs[i].c_str(); op[i] = outs[i].c_str();
    }
    std::vector<int> status(B, 99);
    int rc = chd_phys_solve_dirs(h, B, ip.data(), op.data(), nframes.data(), status.data());
    chd_call_stats cs; chd_phys_get_call_stats(h, &cs);
    int bad = 0; for (int v : status) bad += v != 0;
    std::printf("plan %s: rc %d, %d chunks of %d, %d sequences, %d fallbacks, %lld iterations, %d bad status (%s)\n", argv[p], rc, cs.n_chunks, cs.chunk, cs.n_sequences, cs.n_fallback, cs.total_iters, bad, chd_phys_last_error(h));
    if (rc != 0 || bad) rc_all = 1;
    // warm handle, in-memory interface, the first half: the lanes' buffers and the workspaces are reused
    {
      const int B2 = B / 2 > 0 ? B / 2 : 1;
      std::vector<io::SeqFiles> files(B2); std::vector<chd_seq_in> in(B2); std::vector<chd_seq_out> out(B2); std::vector<io::SnapStore> store(B2);
      std::string err;
      for (int i = 0; i < B2; ++i) { if (!io::read_inputs(in_dirs[i].c_str(), nframes[i], files[i], err)) { std::fprintf(stderr, "read %s: %s\n", in_dirs[i].c_str(), err.c_str()); return 4; } files[i].fill(in[i]); store[i].bind(out[i], nframes[i] + 4); }
      rc = chd_phys_solve_batch(h, B2, in.data(), out.data());
      chd_call_stats c2; chd_phys_get_call_stats(h, &c2);
      long long it2 = 0; for (int i = 0; i < B2; ++i) for (int s = 0; s < CHD_N_STAGES; ++s) it2 += out[i].stage_status[s] == 9 ? 0 : out[i].stage_iters[s];
      std::printf("  warm call: rc %d, %d sequences, %lld iterations (summed from the results: %lld)\n", rc, c2.n_sequences, c2.total_iters, it2);
      if (rc != 0 || it2 != c2.total_iters) rc_all = 1;
    }
    chd_phys_destroy(h);
  }
  return rc_all;
}

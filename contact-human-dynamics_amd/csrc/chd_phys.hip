// chd_phys.hip — libchd_phys.so: the C ABI of include/chd_phys.h on top of the gfx950 solver kernel.
//
// One workgroup per sequence; a batch is one launch (plus one more launch for the sequences
// whose stage 3 did not converge: "STAGE 4: Durations failed ..." phys_optim.cpp:714-749).
// There is no CPU solve path in this library: without a HIP device chd_phys_create fails.
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <thread>
#include <vector>

#include "../../include/chd_phys.h"
#include "chd_model.hpp"
#include "chd_kernels.hpp"
#include "chd_io.hpp"

using namespace chd;

#define CHD_MAX_THREADS 512

__global__ __launch_bounds__(CHD_MAX_THREADS) void chd_solve_kernel(const SeqDesc* descs, const int* index, int lds_doubles, double tol,
                                                                    int stage_first, int stage_last) {
  extern __shared__ double lds[];
  const SeqDesc* q = descs + (index ? index[blockIdx.x] : (int)blockIdx.x);
  run_sequence(q, (LdsD*)lds, lds_doubles, tol, stage_first, stage_last);
}

__global__ __launch_bounds__(CHD_MAX_THREADS) void chd_debug_eval_kernel(const SeqDesc* descs, int seq, int stage, int use_x, int lds_doubles, double* f_out) {
  extern __shared__ double lds[];
  debug_eval(descs + seq, stage, use_x, (LdsD*)lds, lds_doubles, f_out);
}

struct chd_handle {
  int device = 0;
  hipStream_t stream = nullptr;
  chd_config cfg;
  std::string err;
  int lds_bytes = 0;
  int threads = CHD_MAX_THREADS;
};

struct chd_batch {
  int B = 0;
  std::vector<SeqModel> models;
  std::vector<SeqDesc> descs;            // host copy with DEVICE pointers
  std::vector<long long> off_cd, off_ci, off_wd, off_wi, off_od, off_oi;
  long long tot_cd = 0, tot_ci = 0, tot_wd = 0, tot_wi = 0, tot_od = 0, tot_oi = 0;
  double *d_cd = nullptr, *d_wd = nullptr, *d_od = nullptr, *d_f = nullptr;
  int *d_ci = nullptr, *d_wi = nullptr, *d_oi = nullptr, *d_index = nullptr;
  SeqDesc* d_descs = nullptr;
  std::vector<double> h_od;
  std::vector<int> h_oi;
  bool solved = false;
  chd_batch_stats stats{};
  hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};
};

static int fail(chd_handle* h, const std::string& msg) { if (h) h->err = msg; return -1; }
#define HIP_TRY(h, call)                                                                                         \
  do { hipError_t e_ = (call); if (e_ != hipSuccess) return fail(h, std::string(#call) + ": " + hipGetErrorString(e_)); } while (0)

extern "C" {

int chd_phys_version(void) { return CHD_PHYS_ABI_VERSION; }

void chd_config_default(chd_config* c) {
  c->w_com_lin = 0.4; c->w_com_ang = 1.7; c->w_ee = 0.3; c->w_smooth = 0.1; c->w_dur = 0.1;     // phys_optim.cpp:27-31
  const int mi[CHD_N_STAGES] = {7000, 7000, 7000, 2500, 2000, 7000};                              // :571, :640, :652, :706, :743
  for (int i = 0; i < CHD_N_STAGES; ++i) c->max_iter[i] = mi[i];
  c->tol = 1e-3;                                                                                   // :578
  c->threads_per_sequence = 0;
  for (int i = 0; i < 7; ++i) c->reserved[i] = 0;
}

int chd_phys_create(const chd_config* cfg, int device_id, chd_handle** out) {
  if (!out) return -1;
  *out = nullptr;
  chd_handle* h = new chd_handle();
  if (cfg) h->cfg = *cfg; else chd_config_default(&h->cfg);
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
    std::fprintf(stderr, "chd_phys_create: no HIP device available (this library has no CPU path)\n");
    delete h; return -2;
  }
  if (device_id < 0 || device_id >= ndev) { std::fprintf(stderr, "chd_phys_create: bad device id %d (have %d)\n", device_id, ndev); delete h; return -3; }
  h->device = device_id;
  if (hipSetDevice(device_id) != hipSuccess || hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking) != hipSuccess) { delete h; return -4; }
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, device_id) != hipSuccess) { delete h; return -5; }
  size_t lds = prop.maxSharedMemoryPerMultiProcessor;
  if (lds > 160 * 1024) lds = 160 * 1024;
  if (lds < 64 * 1024) lds = 64 * 1024;
  h->lds_bytes = (int)lds - 4096;     // leave room for the compiler's own static LDS
  hipFuncSetAttribute((const void*)chd_solve_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, h->lds_bytes);
  hipFuncSetAttribute((const void*)chd_debug_eval_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, h->lds_bytes);
  int t = h->cfg.threads_per_sequence > 0 ? h->cfg.threads_per_sequence : CHD_MAX_THREADS;
  if (t > CHD_MAX_THREADS) t = CHD_MAX_THREADS;
  if (t < 64) t = 64;
  h->threads = (t / 64) * 64;
  *out = h;
  return 0;
}

void chd_phys_destroy(chd_handle* h) {
  if (!h) return;
  if (h->stream) (void)hipStreamDestroy(h->stream);
  delete h;
}

const char* chd_phys_last_error(const chd_handle* h) { return h ? h->err.c_str() : "null handle"; }

void chd_batch_free(chd_handle* h, chd_batch* b) {
  if (!b) return;
  if (h) (void)hipSetDevice(h->device);
  (void)hipFree(b->d_cd); (void)hipFree(b->d_ci); (void)hipFree(b->d_wd); (void)hipFree(b->d_wi); (void)hipFree(b->d_od); (void)hipFree(b->d_oi);
  (void)hipFree(b->d_descs); (void)hipFree(b->d_index); (void)hipFree(b->d_f);
  for (int k = 0; k < 4; ++k) if (b->ev[k]) (void)hipEventDestroy(b->ev[k]);
  delete b;
}

int chd_batch_upload(chd_handle* h, int B, const chd_seq_in* in, chd_batch** out) {
  if (!h || !in || !out || B <= 0) return fail(h, "chd_batch_upload: bad arguments");
  *out = nullptr;
  HIP_TRY(h, hipSetDevice(h->device));
  chd_batch* b = new chd_batch();
  b->B = B;
  b->models.resize(B);
  // ---- structure tables on the host, in parallel
  std::vector<std::string> errs(B);
  {
    unsigned nt = std::thread::hardware_concurrency();
    if (nt == 0) nt = 4;
    if (nt > 32) nt = 32;
    if ((int)nt > B) nt = B;
    std::vector<std::thread> pool;
    for (unsigned t = 0; t < nt; ++t)
      pool.emplace_back([&, t]() {
        for (int i = t; i < B; i += nt) {
          try { b->models[i].build(in[i], h->cfg); } catch (const std::exception& e) { errs[i] = e.what(); }
        }
      });
    for (auto& th : pool) th.join();
  }
  for (int i = 0; i < B; ++i)
    if (!errs[i].empty()) { std::string m = "sequence " + std::to_string(i) + ": " + errs[i]; chd_batch_free(h, b); return fail(h, m); }
  // ---- pool layout
  auto al = [](long long v) { return (v + 31) & ~31LL; };
  b->off_cd.resize(B); b->off_ci.resize(B); b->off_wd.resize(B); b->off_wi.resize(B); b->off_od.resize(B); b->off_oi.resize(B);
  for (int i = 0; i < B; ++i) {
    SeqModel& M = b->models[i];
    b->off_cd[i] = b->tot_cd; b->tot_cd += al((long long)M.cd.size());
    b->off_ci[i] = b->tot_ci; b->tot_ci += al((long long)M.ci.size());
    b->off_wd[i] = b->tot_wd; b->tot_wd += al(M.wd_size);
    b->off_wi[i] = b->tot_wi; b->tot_wi += al(M.wi_size);
    b->off_od[i] = b->tot_od; b->tot_od += al(out_d_size(M.d.cap));
    b->off_oi[i] = b->tot_oi; b->tot_oi += al(out_i_size(M.d.cap));
  }
  auto bail = [&](const char* what, hipError_t e) { std::string m = std::string(what) + ": " + hipGetErrorString(e); chd_batch_free(h, b); return fail(h, m); };
  hipError_t e;
  if ((e = hipMalloc((void**)&b->d_cd, b->tot_cd * 8)) != hipSuccess) return bail("hipMalloc cd", e);
  if ((e = hipMalloc((void**)&b->d_ci, b->tot_ci * 4)) != hipSuccess) return bail("hipMalloc ci", e);
  if ((e = hipMalloc((void**)&b->d_wd, b->tot_wd * 8)) != hipSuccess) return bail("hipMalloc wd", e);
  if ((e = hipMalloc((void**)&b->d_wi, b->tot_wi * 4)) != hipSuccess) return bail("hipMalloc wi", e);
  if ((e = hipMalloc((void**)&b->d_od, b->tot_od * 8)) != hipSuccess) return bail("hipMalloc od", e);
  if ((e = hipMalloc((void**)&b->d_oi, b->tot_oi * 4)) != hipSuccess) return bail("hipMalloc oi", e);
  if ((e = hipMalloc((void**)&b->d_descs, sizeof(SeqDesc) * B)) != hipSuccess) return bail("hipMalloc descs", e);
  if ((e = hipMalloc((void**)&b->d_index, sizeof(int) * B)) != hipSuccess) return bail("hipMalloc index", e);
  if ((e = hipMalloc((void**)&b->d_f, 64)) != hipSuccess) return bail("hipMalloc f", e);
  if ((e = hipMemsetAsync(b->d_wd, 0, b->tot_wd * 8, h->stream)) != hipSuccess) return bail("memset wd", e);
  if ((e = hipMemsetAsync(b->d_wi, 0, b->tot_wi * 4, h->stream)) != hipSuccess) return bail("memset wi", e);
  if ((e = hipMemsetAsync(b->d_od, 0, b->tot_od * 8, h->stream)) != hipSuccess) return bail("memset od", e);
  if ((e = hipMemsetAsync(b->d_oi, 0, b->tot_oi * 4, h->stream)) != hipSuccess) return bail("memset oi", e);
  // ---- stage pools through one staging buffer each
  {
    std::vector<double> hcd(b->tot_cd, 0.0);
    std::vector<int> hci(b->tot_ci, 0);
    for (int i = 0; i < B; ++i) {
      std::copy(b->models[i].cd.begin(), b->models[i].cd.end(), hcd.begin() + b->off_cd[i]);
      std::copy(b->models[i].ci.begin(), b->models[i].ci.end(), hci.begin() + b->off_ci[i]);
    }
    if ((e = hipMemcpy(b->d_cd, hcd.data(), b->tot_cd * 8, hipMemcpyHostToDevice)) != hipSuccess) return bail("copy cd", e);
    if ((e = hipMemcpy(b->d_ci, hci.data(), b->tot_ci * 4, hipMemcpyHostToDevice)) != hipSuccess) return bail("copy ci", e);
  }
  b->descs.resize(B);
  for (int i = 0; i < B; ++i) {
    SeqDesc dd = b->models[i].d;
    dd.cd = (const GD*)(b->d_cd + b->off_cd[i]); dd.ci = (const GI*)(b->d_ci + b->off_ci[i]);
    dd.wd = (GD*)(b->d_wd + b->off_wd[i]); dd.wi = (GI*)(b->d_wi + b->off_wi[i]);
    dd.out_d = (GD*)(b->d_od + b->off_od[i]); dd.out_i = (GI*)(b->d_oi + b->off_oi[i]);
    b->descs[i] = dd;
  }
  if ((e = hipMemcpy(b->d_descs, b->descs.data(), sizeof(SeqDesc) * B, hipMemcpyHostToDevice)) != hipSuccess) return bail("copy descs", e);
  for (int k = 0; k < 4; ++k) if ((e = hipEventCreate(&b->ev[k])) != hipSuccess) return bail("hipEventCreate", e);
  if ((e = hipStreamSynchronize(h->stream)) != hipSuccess) return bail("sync", e);
  *out = b;
  return 0;
}

static int fetch_raw(chd_handle* h, chd_batch* b) {
  b->h_od.resize(b->tot_od); b->h_oi.resize(b->tot_oi);
  HIP_TRY(h, hipMemcpy(b->h_od.data(), b->d_od, b->tot_od * 8, hipMemcpyDeviceToHost));
  HIP_TRY(h, hipMemcpy(b->h_oi.data(), b->d_oi, b->tot_oi * 4, hipMemcpyDeviceToHost));
  return 0;
}

int chd_batch_solve(chd_handle* h, chd_batch* b) {
  if (!h || !b) return fail(h, "chd_batch_solve: bad arguments");
  HIP_TRY(h, hipSetDevice(h->device));
  const int lds_doubles = h->lds_bytes / 8;
  b->stats = chd_batch_stats{};
  HIP_TRY(h, hipMemsetAsync(b->d_od, 0, b->tot_od * 8, h->stream));
  // ---- launch 1: stages 1.1, 1.2, 2.1, 2.2, 3 for every sequence
  HIP_TRY(h, hipEventRecord(b->ev[0], h->stream));
  hipLaunchKernelGGL(chd_solve_kernel, dim3(b->B), dim3(h->threads), h->lds_bytes, h->stream, b->d_descs, (const int*)nullptr, lds_doubles,
                     h->cfg.tol, 0, 4);
  HIP_TRY(h, hipGetLastError());
  HIP_TRY(h, hipEventRecord(b->ev[1], h->stream));
  HIP_TRY(h, hipStreamSynchronize(h->stream));
  float ms = 0;
  HIP_TRY(h, hipEventElapsedTime(&ms, b->ev[0], b->ev[1]));
  b->stats.kernel_ms[0] = ms;
  // ---- which sequences need the stage-4 fallback (phys_optim.cpp:714)
  auto t0 = std::chrono::steady_clock::now();
  std::vector<double> st((size_t)N_STAGES * RS_STRIDE);
  std::vector<int> idx;
  for (int i = 0; i < b->B; ++i) {
    HIP_TRY(h, hipMemcpy(st.data(), b->d_od + b->off_od[i], st.size() * 8, hipMemcpyDeviceToHost));
    if ((int)st[4 * RS_STRIDE + RS_STATUS] != 0) idx.push_back(i);
  }
  b->stats.n_fallback = (int)idx.size();
  if (!idx.empty()) {
    // durations left by stage 3 -> rebuild the tables of the fallback stage on the host
    std::vector<char> ok(idx.size(), 1);
    std::vector<std::vector<double>> ph(idx.size());
    for (size_t k = 0; k < idx.size(); ++k) {
      const SeqModel& M = b->models[idx[k]];
      ph[k].resize(M.d.tot_phases);
      HIP_TRY(h, hipMemcpy(ph[k].data(), b->d_wd + b->off_wd[idx[k]] + M.d.o_phase_dur, ph[k].size() * 8, hipMemcpyDeviceToHost));
    }
    unsigned nt = std::thread::hardware_concurrency();
    if (nt == 0) nt = 4;
    if (nt > 32) nt = 32;
    if (nt > idx.size()) nt = (unsigned)idx.size();
    std::vector<std::thread> pool;
    for (unsigned t = 0; t < nt; ++t)
      pool.emplace_back([&, t]() {
        for (size_t k = t; k < idx.size(); k += nt) {
          SeqModel& M = b->models[idx[k]];
          std::vector<double> cur[4];
          for (int e = 0; e < 4; ++e) cur[e].assign(ph[k].begin() + M.d.phase_off[e], ph[k].begin() + M.d.phase_off[e] + M.d.n_phase[e]);
          try { M.build_stage(5, h->cfg, cur, false); } catch (...) { M.d.st[5].valid = 0; }
        }
      });
    for (auto& th : pool) th.join();
    for (size_t k = 0; k < idx.size(); ++k) {
      const int i = idx[k];
      SeqModel& M = b->models[i];
      const StageDesc& S = M.d.st[5];
      b->descs[i].st[5] = S;
      if (S.valid) {
        // the stage's regions are contiguous in the pools: [o_pos_var, o_env + 2*(n + m_cap)) and [o_cl, o_task_t + task_cap)
        const long long i0 = S.o_pos_var, i1 = S.o_env + 2LL * (S.n + M.stage_m_cap[5]);
        const long long d0 = S.o_cl, d1 = S.o_task_t + (long long)M.stage_task_cap[5];
        HIP_TRY(h, hipMemcpy(b->d_ci + b->off_ci[i] + i0, M.ci.data() + i0, (i1 - i0) * 4, hipMemcpyHostToDevice));
        HIP_TRY(h, hipMemcpy(b->d_cd + b->off_cd[i] + d0, M.cd.data() + d0, (d1 - d0) * 8, hipMemcpyHostToDevice));
      }
      HIP_TRY(h, hipMemcpy(b->d_descs + i, &b->descs[i], sizeof(SeqDesc), hipMemcpyHostToDevice));
    }
    HIP_TRY(h, hipMemcpy(b->d_index, idx.data(), idx.size() * sizeof(int), hipMemcpyHostToDevice));
    b->stats.host_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    HIP_TRY(h, hipEventRecord(b->ev[2], h->stream));
    hipLaunchKernelGGL(chd_solve_kernel, dim3((unsigned)idx.size()), dim3(h->threads), h->lds_bytes, h->stream, b->d_descs, (const int*)b->d_index,
                       lds_doubles, h->cfg.tol, 5, 5);
    HIP_TRY(h, hipGetLastError());
    HIP_TRY(h, hipEventRecord(b->ev[3], h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    HIP_TRY(h, hipEventElapsedTime(&ms, b->ev[2], b->ev[3]));
    b->stats.kernel_ms[1] = ms;
  } else {
    b->stats.host_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
  }
  b->solved = true;
  // ---- accounting
  if (fetch_raw(h, b) != 0) return -1;
  for (int i = 0; i < b->B; ++i) {
    const double* s = b->h_od.data() + b->off_od[i];
    for (int stg = 0; stg < N_STAGES; ++stg) {
      if (stg == 5 && (int)s[4 * RS_STRIDE + RS_STATUS] == 0) continue;
      const double it = s[stg * RS_STRIDE + RS_ITERS];
      b->stats.total_iters += (long long)it;
      b->stats.total_factorizations += (long long)s[stg * RS_STRIDE + RS_NFACT];
      b->stats.alg_bytes += it * b->models[i].alg_bytes_iter[stg];
    }
    const double* tm = s + N_STAGES * RS_STRIDE + 3LL * 10 * b->models[i].d.cap * 3;     // 100 MHz ticks
    for (int k = 0; k < 24; ++k) b->stats.phase_ms[k] += tm[k] * 1e-5;
    if (tm[5] * 1e-5 > b->stats.max_seq_ms) b->stats.max_seq_ms = tm[5] * 1e-5;
  }
  return 0;
}

int chd_batch_get_stats(chd_handle* h, chd_batch* b, chd_batch_stats* out) {
  if (!h || !b || !out) return fail(h, "chd_batch_get_stats: bad arguments");
  *out = b->stats;
  return 0;
}

int chd_batch_fetch(chd_handle* h, chd_batch* b, chd_seq_out* out) {
  if (!h || !b || !out) return fail(h, "chd_batch_fetch: bad arguments");
  if (!b->solved) return fail(h, "chd_batch_fetch: batch not solved");
  HIP_TRY(h, hipSetDevice(h->device));
  if (b->h_od.empty() && fetch_raw(h, b) != 0) return -1;
  for (int i = 0; i < b->B; ++i) {
    const SeqModel& M = b->models[i];
    const int cap = M.d.cap;
    const double* od = b->h_od.data() + b->off_od[i];
    const int* oi = b->h_oi.data() + b->off_oi[i];
    chd_seq_out& o = out[i];
    const bool fb = (int)od[4 * RS_STRIDE + RS_STATUS] != 0;
    for (int s = 0; s < N_STAGES; ++s) {
      const double* r = od + s * RS_STRIDE;
      const bool ran = (s < 5) || fb;
      o.stage_status[s] = ran ? (int)r[RS_STATUS] : 9;
      o.stage_iters[s] = ran ? (int)r[RS_ITERS] : 0;
      o.stage_kkt_error[s] = ran ? r[RS_KKT] : 0.0;
      o.stage_constr_viol[s] = ran ? r[RS_VIOL] : 0.0;
      o.stage_objective[s] = ran ? r[RS_OBJ] : 0.0;
    }
    o.dynamics_succeed = o.stage_status[3] == 0;                                     // phys_optim.cpp:655
    o.durations_succeed = fb ? (o.stage_status[5] == 0) : 1;                         // :709, :747
    const StageDesc& S = M.d.st[4];
    o.n_vars = S.n; o.n_rows = S.m; o.kkt_dim = S.n + S.m; o.kkt_halfband = S.w; o.kkt_border = S.bc;
    o.nnz_jac = M.d.st[3].nnz_jac;
    for (int s = 0; s < CHD_N_SNAPSHOTS; ++s) {
      chd_snapshot& sn = o.snap[s];
      const int ns = oi[2 * s];
      sn.n_samples = ns; sn.num_frames_header = oi[2 * s + 1];
      const int cnt = ns < sn.capacity ? ns : sn.capacity;
      const double* blk = od + N_STAGES * RS_STRIDE + (long long)s * 10 * cap * 3;
      if (sn.base_lin) std::copy(blk, blk + cnt * 3, sn.base_lin);
      if (sn.base_ang_deg) std::copy(blk + (long long)cap * 3, blk + (long long)cap * 3 + cnt * 3, sn.base_ang_deg);
      for (int e = 0; e < 4; ++e) {
        if (sn.ee_pos) std::copy(blk + (long long)(2 + e) * cap * 3, blk + (long long)(2 + e) * cap * 3 + cnt * 3, sn.ee_pos + (long long)e * sn.capacity * 3);
        if (sn.ee_force) std::copy(blk + (long long)(6 + e) * cap * 3, blk + (long long)(6 + e) * cap * 3 + cnt * 3, sn.ee_force + (long long)e * sn.capacity * 3);
        if (sn.contact) for (int k = 0; k < cnt; ++k) sn.contact[(long long)e * sn.capacity + k] = (unsigned char)oi[8 + ((long long)s * 4 + e) * cap + k];
      }
    }
  }
  return 0;
}

int chd_phys_solve_batch(chd_handle* h, int B, const chd_seq_in* in, chd_seq_out* out) {
  chd_batch* b = nullptr;
  int rc = chd_batch_upload(h, B, in, &b);
  if (rc != 0) return rc;
  rc = chd_batch_solve(h, b);
  if (rc == 0) rc = chd_batch_fetch(h, b, out);
  chd_batch_free(h, b);
  return rc;
}

int chd_phys_solve_dirs(chd_handle* h, int B, const char* const* in_dirs, const char* const* out_dirs, const int* nframes, int* status) {
  if (!h || B <= 0 || !in_dirs || !out_dirs || !nframes) return fail(h, "chd_phys_solve_dirs: bad arguments");
  std::vector<io::SeqFiles> files(B);
  std::vector<int> good;
  for (int i = 0; i < B; ++i) {
    std::string err;
    const bool ok = io::read_inputs(in_dirs[i], nframes[i], files[i], err);
    if (status) status[i] = ok ? 0 : -1;
    if (ok) good.push_back(i); else h->err = std::string(in_dirs[i]) + ": " + err;
  }
  if (good.empty()) return fail(h, "chd_phys_solve_dirs: no readable input directory (" + h->err + ")");
  std::vector<chd_seq_in> in(good.size());
  std::vector<chd_seq_out> out(good.size());
  std::vector<io::SnapStore> store(good.size());
  for (size_t k = 0; k < good.size(); ++k) {
    files[good[k]].fill(in[k]);
    store[k].bind(out[k], nframes[good[k]] + 4);
  }
  int rc = chd_phys_solve_batch(h, (int)good.size(), in.data(), out.data());
  if (rc != 0) return rc;
  for (size_t k = 0; k < good.size(); ++k) {
    std::string err;
    if (!io::write_outputs(out_dirs[good[k]], files[good[k]].dt, out[k], err)) { if (status) status[good[k]] = -2; h->err = err; }
  }
  return 0;
}

int chd_debug_sizes(chd_handle* h, chd_batch* b, int seq, int stage, int* n, int* m, int* kkt_dim, int* halfband, int* border) {
  if (!h || !b || seq < 0 || seq >= b->B || stage < 0 || stage >= N_STAGES) return fail(h, "chd_debug_sizes: bad arguments");
  const StageDesc& S = b->models[seq].d.st[stage];
  if (n) *n = S.n; if (m) *m = S.m; if (kkt_dim) *kkt_dim = S.n + S.m; if (halfband) *halfband = S.w; if (border) *border = S.bc;
  return 0;
}

int chd_debug_eval(chd_handle* h, chd_batch* b, int seq, int stage, const double* x, double* x_out, double* f, double* grad, double* cvals,
                   double* J, double* H) {
  if (!h || !b || seq < 0 || seq >= b->B || stage < 0 || stage >= N_STAGES) return fail(h, "chd_debug_eval: bad arguments");
  HIP_TRY(h, hipSetDevice(h->device));
  const SeqModel& M = b->models[seq];
  const SeqDesc& dd = b->descs[seq];
  const StageDesc& S = M.d.st[stage];
  const int n = S.n, m = S.m;
  double* wd = b->d_wd + b->off_wd[seq];
  if (x) HIP_TRY(h, hipMemcpy(wd + dd.o_vec_n + (long long)VN_XT * dd.max_n, x, n * 8, hipMemcpyHostToDevice));
  hipLaunchKernelGGL(chd_debug_eval_kernel, dim3(1), dim3(h->threads), h->lds_bytes, h->stream, b->d_descs, seq, stage, x ? 1 : 0, h->lds_bytes / 8, b->d_f);
  HIP_TRY(h, hipGetLastError());
  HIP_TRY(h, hipStreamSynchronize(h->stream));
  double fo[2];
  HIP_TRY(h, hipMemcpy(fo, b->d_f, 16, hipMemcpyDeviceToHost));
  if (f) *f = fo[0];
  if (x_out) HIP_TRY(h, hipMemcpy(x_out, wd + dd.o_vec_n + (long long)VN_X * dd.max_n, n * 8, hipMemcpyDeviceToHost));
  if (grad) HIP_TRY(h, hipMemcpy(grad, wd + dd.o_vec_n + (long long)VN_G * dd.max_n, n * 8, hipMemcpyDeviceToHost));
  if (cvals) HIP_TRY(h, hipMemcpy(cvals, wd + dd.o_vec_m + (long long)VM_C * dd.max_m, m * 8, hipMemcpyDeviceToHost));
  if (J || H) {
    const int Nb = S.Nb, bc = S.bc, w = S.w, W2 = 2 * w + 1, LD = n + m;
    std::vector<double> K0b((size_t)Nb * W2), K0x((size_t)bc * LD);
    HIP_TRY(h, hipMemcpy(K0b.data(), wd + dd.o_K0b, K0b.size() * 8, hipMemcpyDeviceToHost));
    if (bc) HIP_TRY(h, hipMemcpy(K0x.data(), wd + dd.o_K0x, K0x.size() * 8, hipMemcpyDeviceToHost));
    const int* pv = M.ci.data() + S.o_pos_var; const int* pr = M.ci.data() + S.o_pos_row;
    auto get = [&](int p, int q) -> double {
      if (p < Nb && q < Nb) { int dl = q - p; if (dl > w || dl < -w) return 0.0; return K0b[(size_t)p * W2 + (dl + w)]; }
      const int hi = p > q ? p : q, lo = p > q ? q : p;
      return K0x[(size_t)(hi - Nb) * LD + lo];
    };
    if (J) for (int i = 0; i < m; ++i) for (int j = 0; j < n; ++j) J[(size_t)i * n + j] = get(pr[i], pv[j]);
    if (H) for (int a = 0; a < n; ++a) for (int c2 = 0; c2 < n; ++c2) H[(size_t)a * n + c2] = get(pv[a], pv[c2]);
  }
  return (int)fo[1];
}

}  // extern "C"

// chd_ik.hip -- C ABI of the IK back-projection step (include/chd_ik.h) on HIP / gfx950.
// One launch per solver iteration; one workgroup of 128 (<= 16 targets) or 256 threads per (video, frame), four to seven of them per compute unit; the state (local rotations and
// translations of every joint of every frame) is double-buffered in HBM because a frame reads its neighbours' previous
// iterate.  See chd_ik_kernels.hpp for the per-frame step and its reference citations.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <string>

#include "chd_ik_host.hpp"

using namespace chd_ik;

namespace {
thread_local std::string g_err;
thread_local double g_kernel_ms = 0.0;
thread_local long long g_frames = 0;
int fail(const std::string& what, hipError_t e = hipSuccess) {
  g_err = e == hipSuccess ? what : what + ": " + hipGetErrorString(e);
  return 1;
}
}  // namespace

__global__ void __launch_bounds__(512) chd_ik_step_kernel(const IkSeq* seqs, const int* frame_seq, const int* frame_idx, IkParams P,
                                                          const int* ipool, const double* dpool, const double* Xin, double* Xout,
                                                          int max_J, int max_T, const unsigned short* pair_tab) {
  extern __shared__ double scratch[];          // IkLds::doubles(max_J, max_T) doubles
  IkLds L;
  L.carve(scratch, max_J, max_T);
  L.pair = pair_tab;
  const int wg = blockIdx.x;
  ik_step_frame(seqs[frame_seq[wg]], frame_idx[wg], P, ipool, dpool, Xin, Xout, L);
}

extern "C" {

const char* chd_ik_version(void) { return "chd_ik 0.1 (gfx950)"; }
void chd_ik_config_default(chd_ik_config* cfg) {      // towr_utils.py:843
  cfg->iterations = 30; cfg->translate = 1; cfg->damping = 7.0; cfg->smoothness = 0.001; cfg->gamma = 1.0;
}
const char* chd_ik_last_error(void) { return g_err.c_str(); }
double chd_ik_last_kernel_ms(void) { return g_kernel_ms; }
long long chd_ik_last_frames(void) { return g_frames; }

int chd_ik_solve_batch(const chd_ik_config* cfg, int device, int B, const chd_ik_seq* in) {
  if (!cfg || !in || B < 1) return fail("bad arguments");
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) return fail("no HIP device (this library has no CPU path)");
  if (device < 0 || device >= ndev) return fail("device index out of range");
  hipError_t e;
  if ((e = hipSetDevice(device)) != hipSuccess) return fail("hipSetDevice", e);
  IkBatch bt;
  if (!bt.build(B, in)) return fail(bt.err);
  const IkParams P = params_of(cfg);
  // a stream of its own per call (stream-ordered allocations, asynchronous copies, one synchronisation at the end): calls from two host
  // threads queue back to back on the device instead of waiting for each other's kernels (see chd_kinopt.hip)
  IkSeq* d_seqs = nullptr; int *d_fs = nullptr, *d_fi = nullptr, *d_ip = nullptr; double *d_dp = nullptr, *d_x0 = nullptr, *d_x1 = nullptr; unsigned short* d_pair = nullptr;
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  hipStream_t st = nullptr;
  if ((e = hipStreamCreateWithFlags(&st, hipStreamNonBlocking)) != hipSuccess) return fail("hipStreamCreate", e);
  bool pool_ok = true;                          // stream-ordered allocation; plain hipMalloc / hipFree where the runtime has no memory pools
  auto dmalloc = [&](void** p, size_t n) {
    hipError_t r = pool_ok ? hipMallocAsync(p, n, st) : hipErrorNotSupported;
    if (r == hipErrorNotSupported) { pool_ok = false; (void)hipGetLastError(); r = hipMalloc(p, n); }
    return r;
  };
  auto release = [&]() {
    for (void* p : {(void*)d_seqs, (void*)d_fs, (void*)d_fi, (void*)d_ip, (void*)d_dp, (void*)d_x0, (void*)d_x1, (void*)d_pair}) if (p && pool_ok) (void)hipFreeAsync(p, st);
    (void)hipStreamSynchronize(st);
    if (!pool_ok) for (void* p : {(void*)d_seqs, (void*)d_fs, (void*)d_fi, (void*)d_ip, (void*)d_dp, (void*)d_x0, (void*)d_x1, (void*)d_pair}) if (p) (void)hipFree(p);
    if (ev0) (void)hipEventDestroy(ev0);
    if (ev1) (void)hipEventDestroy(ev1);
    (void)hipStreamDestroy(st);
  };
#define IK_TRY(call, what) if ((e = (call)) != hipSuccess) { release(); return fail(what, e); }
  const size_t nwg = bt.frame_seq.size(), nst = bt.state.size();
  IK_TRY(dmalloc((void**)&d_seqs, sizeof(IkSeq) * bt.seqs.size()), "hipMalloc seqs");
  IK_TRY(dmalloc((void**)&d_fs, sizeof(int) * nwg), "hipMalloc frame map");
  IK_TRY(dmalloc((void**)&d_fi, sizeof(int) * nwg), "hipMalloc frame map");
  IK_TRY(dmalloc((void**)&d_ip, sizeof(int) * bt.ipool.size()), "hipMalloc ints");
  IK_TRY(dmalloc((void**)&d_dp, sizeof(double) * bt.dpool.size()), "hipMalloc targets");
  IK_TRY(dmalloc((void**)&d_x0, sizeof(double) * nst), "hipMalloc state");
  IK_TRY(dmalloc((void**)&d_x1, sizeof(double) * nst), "hipMalloc state");
  IK_TRY(hipMemcpyAsync(d_seqs, bt.seqs.data(), sizeof(IkSeq) * bt.seqs.size(), hipMemcpyHostToDevice, st), "copy seqs");
  IK_TRY(hipMemcpyAsync(d_fs, bt.frame_seq.data(), sizeof(int) * nwg, hipMemcpyHostToDevice, st), "copy frame map");
  IK_TRY(hipMemcpyAsync(d_fi, bt.frame_idx.data(), sizeof(int) * nwg, hipMemcpyHostToDevice, st), "copy frame map");
  IK_TRY(hipMemcpyAsync(d_ip, bt.ipool.data(), sizeof(int) * bt.ipool.size(), hipMemcpyHostToDevice, st), "copy ints");
  IK_TRY(hipMemcpyAsync(d_dp, bt.dpool.data(), sizeof(double) * bt.dpool.size(), hipMemcpyHostToDevice, st), "copy targets");
  IK_TRY(hipMemcpyAsync(d_x0, bt.state.data(), sizeof(double) * nst, hipMemcpyHostToDevice, st), "copy state");
  std::vector<unsigned short> pairs((size_t)IkLds::pair_entries());      // the elimination's (row, column) table: one for every size
  IkLds::fill_pairs(pairs.data());
  IK_TRY(dmalloc((void**)&d_pair, sizeof(unsigned short) * pairs.size()), "hipMalloc pair table");
  IK_TRY(hipMemcpyAsync(d_pair, pairs.data(), sizeof(unsigned short) * pairs.size(), hipMemcpyHostToDevice, st), "copy pair table");
  const size_t lds = sizeof(double) * (size_t)IkLds::doubles(bt.max_J, bt.max_T);      // 21 KB for J = 33, T = 13; 39 KB for the kinematic optimisation (J = 28, T = 25): four frames per compute unit
  if (lds > 48 * 1024) IK_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(chd_ik_step_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds), "hipFuncSetAttribute");
  IK_TRY(hipEventCreate(&ev0), "hipEventCreate");
  IK_TRY(hipEventCreate(&ev1), "hipEventCreate");
  // A step is a sequence of short dependent chains (the 3T x 3T elimination 57 % of it: profiles/r05_experiments.md section 9): frames in flight per compute unit are what
  // counts.  128 threads for the back-projection's 13 targets (39 x 39); for the 25 targets (75 x 75) of the kinematic optimisation's initialisation CHD_IK_THREADS (default 256:
  // with the packed matrix four workgroups of 256 fit a compute unit's LDS and registers, of 512 only two)
  unsigned nthreads = bt.max_T > 16 ? 256u : 128u;
  if (const char* e_ = getenv("CHD_IK_THREADS")) { const int v = atoi(e_); if (v == 128 || v == 256 || v == 512) nthreads = (unsigned)v; }
  double* cur = d_x0; double* nxt = d_x1;
  IK_TRY(hipEventRecord(ev0, st), "hipEventRecord");
  (void)hipGetLastError();      // an error another library of the process left behind in this thread (hipBLASLt's kernel look-ups do) is not this launch's
  for (int it = 0; it < P.iterations; ++it) {
    hipLaunchKernelGGL(chd_ik_step_kernel, dim3((unsigned)nwg), dim3(nthreads), lds, st, d_seqs, d_fs, d_fi, P, d_ip, d_dp, cur, nxt, bt.max_J, bt.max_T, (const unsigned short*)d_pair);
    IK_TRY(hipGetLastError(), "launch");
    double* t = cur; cur = nxt; nxt = t;
  }
  IK_TRY(hipEventRecord(ev1, st), "hipEventRecord");
  std::vector<double> fin(nst);
  IK_TRY(hipMemcpyAsync(fin.data(), cur, sizeof(double) * nst, hipMemcpyDeviceToHost, st), "copy result");
  IK_TRY(hipStreamSynchronize(st), "synchronize");
  float ms = 0.0f;
  IK_TRY(hipEventElapsedTime(&ms, ev0, ev1), "hipEventElapsedTime");
#undef IK_TRY
  bt.scatter(fin.data(), in);
#ifdef IK_PROFILE
  {
    unsigned long long pr[16];
    if (hipMemcpyFromSymbol(pr, HIP_SYMBOL(ik_prof), sizeof(pr)) == hipSuccess) {
      double tot = 0; for (int k = 0; k < 12; ++k) tot += (double)pr[k];
      fprintf(stderr, "IK_PROFILE %lld frames x %d iterations, ticks per frame-step:", (long long)nwg, P.iterations);
      for (int k = 0; k < 12; ++k) fprintf(stderr, " [%d] %.0f", k, (double)pr[k] / ((double)((nwg + 63) / 64) * P.iterations));
      fprintf(stderr, " total %.0f\n", tot / ((double)((nwg + 63) / 64) * P.iterations));
      unsigned long long z[16] = {0}; (void)hipMemcpyToSymbol(HIP_SYMBOL(ik_prof), z, sizeof(z));
    }
  }
#endif
  release();
  g_kernel_ms = ms; g_frames = (long long)nwg;
  g_err.clear();
  return 0;
}

}  // extern "C"

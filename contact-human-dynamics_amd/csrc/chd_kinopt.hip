// chd_kinopt.hip -- C ABI of the kinematic optimisation's least-squares solves (include/chd_kinopt.h) on HIP / gfx950.
// One workgroup of 512 threads per (video, stage) problem runs the whole trust-region solve (chd_kinopt_kernels.hpp);
// the host packs the batch into three pools (constants, contacts, start points) and allocates one workspace per video.
#include <hip/hip_runtime.h>

#include <string>

#include "chd_kinopt_host.hpp"

using namespace chd_kin;

namespace {
thread_local std::string g_err;
thread_local double g_kernel_ms = 0.0;
int fail(const std::string& what, hipError_t e = hipSuccess) {
  g_err = e == hipSuccess ? what : what + ": " + hipGetErrorString(e);
  return 1;
}
}  // namespace

// waves per SIMD the register budget is set for: 2 = one 512-thread workgroup per compute unit with up to 256 VGPRs,
// 4 = two resident workgroups with up to 128 each
#ifndef CHD_KIN_WAVES_PER_EU
#define CHD_KIN_WAVES_PER_EU 2
#endif
__global__ void __launch_bounds__(512, CHD_KIN_WAVES_PER_EU) chd_kin_solve_kernel(const KinSeq* seqs, KinParams P, const double* dpool, const int* ipool, double* work,
                                                            double* state, double* stats, int lds_doubles) {
  extern __shared__ double tile[];              // the products' frame tiles
  __shared__ double red[48];
  __shared__ KinParams Ps;
  if (threadIdx.x == 0) Ps = P;
  __syncthreads();
  KinCtx c;
  const KinSeq* q = seqs + blockIdx.x;
  kin_bind(c, q, &Ps, dpool, ipool, work, red, tile, lds_doubles);
  kin_solve(c, state + q->o_x, stats + 8 * blockIdx.x);
}

extern "C" {

const char* chd_kin_version(void) { return "chd_kinopt 0.1 (gfx950)"; }
void chd_kin_config_default(chd_kin_config* cfg) { config_default(cfg); }
const char* chd_kin_last_error(void) { return g_err.c_str(); }
double chd_kin_last_kernel_ms(void) { return g_kernel_ms; }

int chd_kin_solve_batch(const chd_kin_config* cfg, int device, int B, chd_kin_seq* in) {
  if (!cfg || !in || B < 1) return fail("bad arguments");
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) return fail("no HIP device (this library has no CPU path)");
  if (device < 0 || device >= ndev) return fail("device index out of range");
  hipError_t e;
  if ((e = hipSetDevice(device)) != hipSuccess) return fail("hipSetDevice", e);
  {   // LDS doubles per workgroup (default 18 432 = 144 KB, or reserved[1]): the product passes address 3 * 84 * (TF + 2) doubles with TF >= 1 (kin_jv) and 336 (kin_jtu).
      // The default is checked against the device like an explicit value: a device / partition mode with less LDS fails here with a message, not in the launch.
    int lds_max = 0;
    if ((e = hipDeviceGetAttribute(&lds_max, hipDeviceAttributeMaxSharedMemoryPerBlock, device)) != hipSuccess) return fail("hipDeviceGetAttribute", e);
    const long long want = cfg->reserved[1] > 0 ? cfg->reserved[1] : 18432;
    if (want < 756 || want * 8 > lds_max) return fail(cfg->reserved[1] > 0 ? "reserved[1] (LDS doubles per workgroup) out of range: 756 .. device limit" : "the device offers less than the 144 KB of LDS per workgroup the default frame tiles need: set reserved[1] (LDS doubles per workgroup, >= 756)");
  }
  KinBatch bt;
  if (!bt.build(cfg, B, in)) return fail(bt.err);
  // Everything of a call is ordered on a stream of its own (stream-ordered allocations, asynchronous copies, one synchronisation at the
  // end): two host threads can keep the device busy back to back -- with the default stream and hipDeviceSynchronize each call would also
  // wait for the other thread's kernel, and the host steps of the two would fall into lockstep (kinematic_optimizer.KinematicOptimizer.optimize)
  KinSeq* d_seqs = nullptr; double *d_dp = nullptr, *d_work = nullptr, *d_state = nullptr, *d_stats = nullptr; int* d_ip = nullptr;
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
    for (void* p : {(void*)d_seqs, (void*)d_dp, (void*)d_work, (void*)d_state, (void*)d_stats, (void*)d_ip}) if (p && pool_ok) (void)hipFreeAsync(p, st);
    (void)hipStreamSynchronize(st);
    if (!pool_ok) for (void* p : {(void*)d_seqs, (void*)d_dp, (void*)d_work, (void*)d_state, (void*)d_stats, (void*)d_ip}) if (p) (void)hipFree(p);
    if (ev0) (void)hipEventDestroy(ev0);
    if (ev1) (void)hipEventDestroy(ev1);
    (void)hipStreamDestroy(st);
  };
#define KIN_TRY(call, what) if ((e = (call)) != hipSuccess) { release(); return fail(what, e); }
  KIN_TRY(dmalloc((void**)&d_seqs, sizeof(KinSeq) * bt.seqs.size()), "hipMalloc descriptors");
  KIN_TRY(dmalloc((void**)&d_dp, sizeof(double) * bt.dpool.size()), "hipMalloc constants");
  KIN_TRY(dmalloc((void**)&d_ip, sizeof(int) * bt.ipool.size()), "hipMalloc contacts");
  KIN_TRY(dmalloc((void**)&d_work, sizeof(double) * (size_t)bt.work_total), "hipMalloc workspace");
  KIN_TRY(dmalloc((void**)&d_state, sizeof(double) * bt.state.size()), "hipMalloc state");
  KIN_TRY(dmalloc((void**)&d_stats, sizeof(double) * 8 * (size_t)B), "hipMalloc statistics");
  KIN_TRY(hipMemcpyAsync(d_seqs, bt.seqs.data(), sizeof(KinSeq) * bt.seqs.size(), hipMemcpyHostToDevice, st), "copy descriptors");
  KIN_TRY(hipMemcpyAsync(d_dp, bt.dpool.data(), sizeof(double) * bt.dpool.size(), hipMemcpyHostToDevice, st), "copy constants");
  KIN_TRY(hipMemcpyAsync(d_ip, bt.ipool.data(), sizeof(int) * bt.ipool.size(), hipMemcpyHostToDevice, st), "copy contacts");
  KIN_TRY(hipMemcpyAsync(d_state, bt.state.data(), sizeof(double) * bt.state.size(), hipMemcpyHostToDevice, st), "copy start points");
  KIN_TRY(hipMemsetAsync(d_stats, 0, sizeof(double) * 8 * (size_t)B, st), "clear statistics");
  KIN_TRY(hipEventCreate(&ev0), "hipEventCreate");
  KIN_TRY(hipEventCreate(&ev1), "hipEventCreate");
  KIN_TRY(hipEventRecord(ev0, st), "hipEventRecord");
  // 512 threads: measured against 256 and 1024 (profiles/r02h_kinopt/sweep.md); results are bitwise reproducible for a fixed
  // workgroup size (fixed reduction trees) and move at the solve's own sensitivity level when it changes
  const int nthreads = cfg->reserved[0] == 256 ? 256 : 512;
  // 144 KB of LDS per workgroup: tiles of 71 frames for J v, 54 for J^T u.  (The kernel's 256 VGPRs allow one 512-thread workgroup
  // per compute unit anyway; against 72 KB the larger tiles halve the number of phases and barriers per product: 2.54 -> 2.32 s of
  // least-squares kernels for 256 clips x 100 frames.  A 128-VGPR build with two resident workgroups was measured too: 3.45 s.)
  const int lds_doubles = cfg->reserved[1] > 0 ? cfg->reserved[1] : 18432;
  KIN_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(chd_kin_solve_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(sizeof(double) * lds_doubles)), "hipFuncSetAttribute");
  (void)hipGetLastError();      // an error another library of the process left behind in this thread is not this launch's
  hipLaunchKernelGGL(chd_kin_solve_kernel, dim3((unsigned)B), dim3((unsigned)nthreads), sizeof(double) * (size_t)lds_doubles, st, d_seqs, bt.P, d_dp, d_ip, d_work, d_state, d_stats, lds_doubles);
  KIN_TRY(hipGetLastError(), "launch");
  KIN_TRY(hipEventRecord(ev1, st), "hipEventRecord");
  std::vector<double> fin(bt.state.size()), stats(8 * (size_t)B);
  KIN_TRY(hipMemcpyAsync(fin.data(), d_state, sizeof(double) * fin.size(), hipMemcpyDeviceToHost, st), "copy solutions");
  KIN_TRY(hipMemcpyAsync(stats.data(), d_stats, sizeof(double) * stats.size(), hipMemcpyDeviceToHost, st), "copy statistics");
  KIN_TRY(hipStreamSynchronize(st), "synchronize");
  float ms = 0.0f;
  KIN_TRY(hipEventElapsedTime(&ms, ev0, ev1), "hipEventElapsedTime");
#undef KIN_TRY
  bt.scatter(fin.data(), stats.data(), in);
  release();
  g_kernel_ms = ms;
  g_err.clear();
  return 0;
}

}  // extern "C"

// chd_kinopt.hip -- C ABI of the kinematic optimisation's least-squares solves (include/chd_kinopt.h) on HIP / gfx950.
// A cluster of G = ceil(frames / 13) workgroups of 512 threads per (video, stage) problem runs the whole trust-region solve with LSMR's state in
// the LDS of its G compute units (chd_kinopt_kernels.hpp); the host packs the batch into three pools (constants, contacts, start points), allocates
// one workspace per video and launches one persistent grid per cluster size.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <mutex>
#include <string>

#include "chd_kinopt_host.hpp"

using namespace chd_kin;

namespace {
thread_local std::string g_err;
thread_local double g_kernel_ms = 0.0;
thread_local int g_retried = 0;                 // the last call on this thread fell back to one workgroup per clip (chd_kin_last_call_retried)
int fail(const std::string& what, hipError_t e = hipSuccess) {
  g_err = e == hipSuccess ? what : what + ": " + hipGetErrorString(e);
  return 1;
}
}  // namespace

// One workgroup of 512 threads per compute unit (up to 256 VGPRs, the LDS block below), persistent: consecutive workgroups form clusters of G, each cluster
// takes clips from the launch's queue until it is empty.  ALL workgroups of a launch must be resident at once -- the members of a cluster poll each other's
// published values (chd_kinopt_kernels.hpp, kc_sync) -- so the host sizes the grid by the device's occupancy for this kernel and never has two launches in flight.
__global__ void __launch_bounds__(512, 2) chd_kin_solve_kernel(const KinSeq* seqs, const int* order, int n_clips, KinParams P, const double* dpool, const int* ipool, double* work,
                                                               double* state, double* stats, KinSlot* slots, int* queue, int* abort_flag, int G, int lds_doubles, int test_absent) {
  extern __shared__ double lds[];               // received halos + the slice's share of LSMR's state (kin_bind_wg)
  __shared__ double red[16 * KC_PARTS], gath[KC_MAXG * KC_PARTS];
  __shared__ KinParams Ps;
  __shared__ KinClip clip;
  __shared__ KinWg wg;
  __shared__ KinLsmr lsmr;
  __shared__ int dead;
  if (threadIdx.x == 0) { Ps = P; dead = 0; }
  __syncthreads();
  KinCtx c;
  const int cluster = blockIdx.x / G, g = blockIdx.x % G;
  if (test_absent && G > 1 && cluster == 0 && g == G - 1) return;      // (test hook: a member that never shows up)
  c.patience = test_absent ? KC_PATIENCE / 20 : KC_PATIENCE;
  c.slots = slots + (size_t)cluster * G; c.epoch = 0; c.G = G; c.abort_flag = abort_flag; c.dead = (KO_LDSQ int*)&dead;
  c.red = (KO_LDSQ double*)red; c.gath = (KO_LDSQ double*)gath; c.P = (const KO_LDSQ KinParams*)&Ps; c.k = (KO_LDSQ KinClip*)&clip; c.wg = (KO_LDSQ KinWg*)&wg; c.S = (KO_LDSQ KinLsmr*)&lsmr;
  if (threadIdx.x == 0) wg.g = g;
  __syncthreads();
  kin_lane_tables(c);
  for (;;) {
    KoAcc pick[1][KC_PARTS];                    // the cluster's first thread draws a clip; the number reaches the others as a "sum"
    if (g == 0 && threadIdx.x == 0) pick[0][0].s = (double)atomicAdd(queue, 1);
    kc_sync(c, pick, 1);
    const int idx = (int)kc_sum(c, 0);
    if (kc_dead(c) || idx >= n_clips) break;
    const int b = order[idx];
    if (threadIdx.x == 0) {
      kin_bind_clip(*c.k, seqs + b, dpool, ipool);
      kin_bind_wg(c, *c.wg, g, work, lds, lds_doubles);
    }
    __syncthreads();
    kin_solve(c, state + seqs[b].o_x, stats + (size_t)KIN_STATS * b);
  }
}

static std::mutex g_launch_mutex;               // one spinning launch on the device at a time (per process)

extern "C" {

const char* chd_kin_version(void) { return "chd_kinopt 0.1 (gfx950)"; }
void chd_kin_config_default(chd_kin_config* cfg) { config_default(cfg); }
const char* chd_kin_last_error(void) { return g_err.c_str(); }
double chd_kin_last_kernel_ms(void) { return g_kernel_ms; }
int chd_kin_last_call_retried(void) { return g_retried; }

static int solve_batch_once(const chd_kin_config* cfg, int device, int B, chd_kin_seq* in, bool* not_resident);

// The clusters' workgroups wait on each other, so a launch needs all of them resident at once; the grid is sized by the occupancy the runtime reports, which
// holds on an exclusive device.  When something else holds compute units (a second process or rank on the same GPU, a compute-unit mask, a long co-tenant
// kernel) a cluster's bounded wait (5 s) ends the launch with an error flag.  The batch is then solved ONCE MORE with one workgroup per clip (frames per
// workgroup = the longest clip: the slices live in device memory and the synchronisation never leaves the compute unit -- round 4's form, ~4 x slower, no
// co-residency assumption), from the caller's unchanged start points; chd_kin_last_call_retried() tells.  reserved[0] = 1 turns the retry off (the error is
// returned: tests, and callers who would rather fail fast).  The device should be exclusive to the process for the cluster form to pay.
int chd_kin_solve_batch(const chd_kin_config* cfg, int device, int B, chd_kin_seq* in) {
  g_retried = 0;
  bool not_resident = false;
  const int rc = solve_batch_once(cfg, device, B, in, &not_resident);
  if (rc == 0 || !not_resident || !cfg || cfg->reserved[0] == 1) return rc;
  chd_kin_config one = *cfg;
  int fmax = 2;
  for (int b = 0; b < B; ++b) fmax = in[b].n_frames > fmax ? in[b].n_frames : fmax;
  one.reserved[2] = fmax;                       // one workgroup per clip
  one.reserved[3] = 0;
  const std::string first = g_err;
  const int rc2 = solve_batch_once(&one, device, B, in, &not_resident);
  if (rc2 != 0) { g_err = first + "; the retry with one workgroup per clip failed too: " + g_err; return rc2; }
  g_retried = 1;
  return 0;
}

static int solve_batch_once(const chd_kin_config* cfg, int device, int B, chd_kin_seq* in, bool* not_resident) {
  *not_resident = false;
  if (!cfg || !in || B < 1) return fail("bad arguments");
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) return fail("no HIP device (this library has no CPU path)");
  if (device < 0 || device >= ndev) return fail("device index out of range");
  hipError_t e;
  if ((e = hipSetDevice(device)) != hipSuccess) return fail("hipSetDevice", e);
  int lds_max = 0, n_cu = 0;
  if ((e = hipDeviceGetAttribute(&lds_max, hipDeviceAttributeMaxSharedMemoryPerBlock, device)) != hipSuccess) return fail("hipDeviceGetAttribute", e);
  if ((e = hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, device)) != hipSuccess) return fail("hipDeviceGetAttribute", e);
  {   // LDS doubles per workgroup (default 19 760 = 154 KB, or reserved[1]), checked against the device like an explicit value: a device / partition mode with
      // less LDS fails here with a message, not in the launch
    const long long want = cfg->reserved[1] > 0 ? cfg->reserved[1] : (long long)KIN_LDS_DOUBLES_DEFAULT;
    if (want < HALO_V + KC_HALO || want * 8 + 4608 > lds_max) return fail(cfg->reserved[1] > 0 ? "reserved[1] (LDS doubles per workgroup) out of range: 598 .. device limit - 128" : "the device offers less than the 159 KB of LDS per workgroup the default slices need: set reserved[1] (LDS doubles per workgroup, >= 598)");
  }
  KinBatch bt;
  if (!bt.build(cfg, B, in)) return fail(bt.err);
  // Everything of a call is ordered on a stream of its own (stream-ordered allocations, asynchronous copies, one synchronisation at the
  // end): two host threads can keep the device busy back to back -- with the default stream and hipDeviceSynchronize each call would also
  // wait for the other thread's kernel, and the host steps of the two would fall into lockstep (kinematic_optimizer.KinematicOptimizer.optimize)
  KinSeq* d_seqs = nullptr; double *d_dp = nullptr, *d_work = nullptr, *d_state = nullptr, *d_stats = nullptr; int *d_ip = nullptr, *d_order = nullptr; KinSlot* d_slots = nullptr;
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
    for (void* p : {(void*)d_seqs, (void*)d_dp, (void*)d_work, (void*)d_state, (void*)d_stats, (void*)d_ip, (void*)d_order, (void*)d_slots}) if (p && pool_ok) (void)hipFreeAsync(p, st);
    (void)hipStreamSynchronize(st);
    if (!pool_ok) for (void* p : {(void*)d_seqs, (void*)d_dp, (void*)d_work, (void*)d_state, (void*)d_stats, (void*)d_ip, (void*)d_order, (void*)d_slots}) if (p) (void)hipFree(p);
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
  KIN_TRY(dmalloc((void**)&d_stats, sizeof(double) * KIN_STATS * (size_t)B), "hipMalloc statistics");
  KIN_TRY(hipMemcpyAsync(d_seqs, bt.seqs.data(), sizeof(KinSeq) * bt.seqs.size(), hipMemcpyHostToDevice, st), "copy descriptors");
  KIN_TRY(hipMemcpyAsync(d_dp, bt.dpool.data(), sizeof(double) * bt.dpool.size(), hipMemcpyHostToDevice, st), "copy constants");
  KIN_TRY(hipMemcpyAsync(d_ip, bt.ipool.data(), sizeof(int) * bt.ipool.size(), hipMemcpyHostToDevice, st), "copy contacts");
  KIN_TRY(hipMemcpyAsync(d_state, bt.state.data(), sizeof(double) * bt.state.size(), hipMemcpyHostToDevice, st), "copy start points");
  KIN_TRY(hipMemsetAsync(d_stats, 0, sizeof(double) * KIN_STATS * (size_t)B, st), "clear statistics");
  // 512 threads per workgroup (one (frame, joint) item per thread for slices of up to 16 frames); results are bitwise reproducible for fixed
  // reserved[] values (fixed reduction trees) and independent of the batch a clip is in
  const int nthreads = 512;
  const int lds_doubles = bt.lds_doubles;
  KIN_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(chd_kin_solve_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(sizeof(double) * lds_doubles)), "hipFuncSetAttribute");
  int per_cu = 0;
  KIN_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, chd_kin_solve_kernel, nthreads, sizeof(double) * (size_t)lds_doubles), "hipOccupancyMaxActiveBlocksPerMultiprocessor");
  const long long resident = (long long)per_cu * n_cu;
  if (resident < KC_MAXG) { release(); return fail("the device cannot hold a cluster of 16 workgroups of this kernel resident"); }
  // the clips of a call by cluster size: one launch per size, in one order buffer; slots and queue counters of all launches cleared up front
  std::vector<int> order; std::vector<long long> grid;
  for (const KinGroup& g : bt.groups) {
    long long clusters = resident / g.G;
    if (clusters > (long long)g.clips.size()) clusters = (long long)g.clips.size();
    grid.push_back(clusters * g.G);
    order.insert(order.end(), g.clips.begin(), g.clips.end());
  }
  long long slots_total = 0;
  for (long long g : grid) slots_total += g;
  KIN_TRY(dmalloc((void**)&d_order, sizeof(int) * (order.size() + bt.groups.size() + 1)), "hipMalloc clip order");      // + the queues' counters + the abort flag
  KIN_TRY(dmalloc((void**)&d_slots, sizeof(KinSlot) * (size_t)slots_total), "hipMalloc cluster slots");
  KIN_TRY(hipMemcpyAsync(d_order, order.data(), sizeof(int) * order.size(), hipMemcpyHostToDevice, st), "copy clip order");
  KIN_TRY(hipMemsetAsync(d_order + order.size(), 0, sizeof(int) * (bt.groups.size() + 1), st), "clear queues");
  KIN_TRY(hipMemsetAsync(d_slots, 0, sizeof(KinSlot) * (size_t)slots_total, st), "clear cluster slots");
  KIN_TRY(hipEventCreate(&ev0), "hipEventCreate");
  KIN_TRY(hipEventCreate(&ev1), "hipEventCreate");
  std::vector<double> fin(bt.state.size()), stats((size_t)KIN_STATS * B);
  int gave_up = 0;
  {
    std::lock_guard<std::mutex> only_one(g_launch_mutex);
    KIN_TRY(hipEventRecord(ev0, st), "hipEventRecord");
    (void)hipGetLastError();      // an error another library of the process left behind in this thread is not this launch's
    size_t o_order = 0, o_slot = 0;
    for (size_t k = 0; k < bt.groups.size(); ++k) {
      const KinGroup& g = bt.groups[k];
      hipLaunchKernelGGL(chd_kin_solve_kernel, dim3((unsigned)grid[k]), dim3((unsigned)nthreads), sizeof(double) * (size_t)lds_doubles, st, d_seqs, d_order + o_order, (int)g.clips.size(), bt.P,
                         d_dp, d_ip, d_work, d_state, d_stats, d_slots + o_slot, d_order + order.size() + k, d_order + order.size() + bt.groups.size(), g.G, lds_doubles, cfg->reserved[3] == 0x7e57 ? 1 : 0);
      KIN_TRY(hipGetLastError(), "launch");
      o_order += g.clips.size(); o_slot += (size_t)grid[k];
    }
    KIN_TRY(hipEventRecord(ev1, st), "hipEventRecord");
    // the results come back INSIDE the lock: a copy from pageable memory is a small kernel of the runtime, and once the next caller's launch holds every compute unit
    // (its workgroups fill LDS and the register file) that kernel waits for the whole launch -- this caller's host steps then no longer overlap it (measured: both
    // halves of a 256-clip batch returned from their first solve only when the SECOND half's kernel had ended)
    KIN_TRY(hipMemcpyAsync(&gave_up, d_order + order.size() + bt.groups.size(), sizeof(int), hipMemcpyDeviceToHost, st), "copy abort flag");
    KIN_TRY(hipMemcpyAsync(fin.data(), d_state, sizeof(double) * fin.size(), hipMemcpyDeviceToHost, st), "copy solutions");
    KIN_TRY(hipMemcpyAsync(stats.data(), d_stats, sizeof(double) * stats.size(), hipMemcpyDeviceToHost, st), "copy statistics");
    KIN_TRY(hipStreamSynchronize(st), "synchronize");
  }
  float ms = 0.0f;
  KIN_TRY(hipEventElapsedTime(&ms, ev0, ev1), "hipEventElapsedTime");
#undef KIN_TRY
  if (gave_up) { release(); *not_resident = true; return fail("a cluster of workgroups waited too long (5 s) for a member: the launch was not fully resident (another process on the device, or a partition with fewer compute units than reported?)"); }
  bt.scatter(fin.data(), stats.data(), in);
#ifdef KIN_PROFILE
  {
    double seg[16] = {0}, tot = 0, its = 0;
    for (int b = 0; b < B; ++b) { for (int k = 0; k < 16; ++k) { seg[k] += stats[(size_t)KIN_STATS * b + 8 + k]; tot += stats[(size_t)KIN_STATS * b + 8 + k]; } its += stats[(size_t)KIN_STATS * b + 4]; }
    fprintf(stderr, "KIN_PROFILE %d clips, %.0f LSMR iterations, ticks per iteration:", B, its);
    for (int k = 0; k < 12; ++k) fprintf(stderr, " [%d] %.0f", k, seg[k] / its);
    fprintf(stderr, " total %.0f\n", tot / its);
    fprintf(stderr, "KIN_PROFILE waits per iteration by rank (sync 1 / sync 2):");
    for (int g = 0; g < 16; ++g) { double a = 0, b2 = 0; for (int b = 0; b < B; ++b) { a += stats[(size_t)KIN_STATS * b + 24 + 2 * g]; b2 += stats[(size_t)KIN_STATS * b + 25 + 2 * g]; } if (a > 0) fprintf(stderr, " [%d] %.0f / %.0f", g, a / its, b2 / its); }
    fprintf(stderr, "\n");
  }
#endif
  release();
  g_kernel_ms = ms;
  g_err.clear();
  return 0;
}

}  // extern "C"

// chd_phys.hip — libchd_phys.so: the C ABI of include/chd_phys.h on top of the gfx950 solver kernel.
//
// One workgroup per sequence; a batch is one launch (plus one more launch for the sequences
// whose stage 3 did not converge: "STAGE 4: Durations failed ..." phys_optim.cpp:714-749).
// There is no CPU solve path in this library: without a HIP device chd_phys_create fails.
#ifdef CHD_HOST_EMU_HIP_STUB        // sanitizer builds of the HOST side (tests/host_emu/pipeline_stress.cpp): a stand-in runtime whose streams are threads and whose kernel
#include CHD_HOST_EMU_HIP_STUB     // launches run the host emulation of the kernel source; test infrastructure -- libchd_phys.so is never built this way
#else
#include <hip/hip_runtime.h>
#endif

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/chd_phys.h"
#include "chd_model.hpp"
#include "chd_kernels.hpp"
#include "chd_io.hpp"

using namespace chd;

// ---- roctx ranges (SURVEY 5 "tracing"): the host phases of a call -- table build, upload + launch, wait / fallback, fetch, file output -- show up as named
// ranges on rocprofv3's marker track (`rocprofv3 --marker-trace --kernel-trace`), next to the kernel launches they surround.  The marker library is looked
// up at run time (no link dependency): without it the ranges are no-ops.
#ifndef CHD_HOST_EMU
#include <dlfcn.h>
namespace {
struct RoctxApi {
  int (*push)(const char*) = nullptr; int (*pop)() = nullptr;
  RoctxApi() {
    void* h = dlopen("librocprofiler-sdk-roctx.so", RTLD_LAZY | RTLD_LOCAL);
    if (!h) h = dlopen("libroctx64.so", RTLD_LAZY | RTLD_LOCAL);
    if (h) { push = (int (*)(const char*))dlsym(h, "roctxRangePushA"); pop = (int (*)())dlsym(h, "roctxRangePop"); if (!push || !pop) push = nullptr; }
  }
};
RoctxApi& roctx_api() { static RoctxApi a; return a; }
}  // namespace
struct TraceRange {
  bool on;
  explicit TraceRange(const std::string& name) : on(roctx_api().push != nullptr) { if (on) roctx_api().push(name.c_str()); }
  ~TraceRange() { if (on) roctx_api().pop(); }
};
#else
struct TraceRange { explicit TraceRange(const std::string&) {} };
#endif

#define CHD_MAX_THREADS 512

// A launch is persistent: `grid` resident workgroups (one per compute unit: a workgroup needs the whole LDS) take sequence
// after sequence from a queue (`order[0 .. n_items)`, next index = atomic counter) until it is drained.  A workgroup owns
// a workspace (KKT storage, solver vectors: ~30 MB at 90 frames) for the lifetime of the handle; a sequence owns only its
// inputs, structure tables and results.  The descriptor of the sequence being solved and the solver context sit in LDS.
static_assert(sizeof(SeqDesc) % 8 == 0, "SeqDesc is copied word by word");
static_assert(sizeof(SeqDesc) + sizeof(Ctx) + 64 <= 4096, "static LDS of the solver kernel must fit the 4 KB left beside the dynamic part");

enum { CHD_SLOT_WAIT_SENTINEL = 1 << 28 };      // added to a launch's queue counter by a workgroup that gave up waiting for a workspace slot (launch_drained reports it)
#ifndef CHD_HOST_EMU
// Workspace slots.  A resident workgroup needs a workspace (~35 MB at 90 frames) only while it is resident, and at most `n_slots` workgroups are (one per
// compute unit: a workgroup needs the whole LDS) -- however many launches are in flight.  So the handle owns ONE set of n_slots workspaces, and a workgroup
// claims a free one when it starts and gives it back when its launch's queue is drained.  Two launches that overlap in time hand a slot from a workgroup on
// one XCD to a workgroup on another without a kernel boundary in between: the release (agent scope) writes the first owner's L2 back before the flag
// clears, the acquire pairs with it.  (Until round 4 a launch indexed a pool of its own by blockIdx: four pools, four launches in flight, and a pool could not
// be reused before the last straggler of its previous launch had finished.)
__device__ inline int claim_slot(int* slot_busy, int n_slots) {
  // A slot holder always makes progress (it never waits for anything a waiting workgroup owns), so a free slot turns up as soon as one launch's queue is
  // drained: the wait is bounded only against a slot flag that was LEFT SET -- a faulted or aborted earlier launch on the same handle -- which would otherwise
  // hang the device: after ~10 minutes of wall clock (100 MHz counter; far beyond any launch of this library) the workgroup gives up, returns -1 and the
  // kernel marks the launch's queue counter so that the host fails the call (launch_drained).  (Until round 5 the loop gave up after 4096 sweeps and the
  // workgroup left WITHOUT a trace: sequences were never solved and their zeroed result slots read as "converged".)
  int s = (int)(blockIdx.x % (unsigned)n_slots);
  const long long t0 = (long long)wall_clock64();
  for (unsigned sweep = 0;; ++sweep) {
    int expected = 0;
    if (__hip_atomic_compare_exchange_strong(&slot_busy[s], &expected, 1, __ATOMIC_ACQUIRE, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) return s;
    s = s + 1 == n_slots ? 0 : s + 1;
    if ((sweep & 63u) == 63u) {
      __builtin_amdgcn_s_sleep(64);
      if ((long long)wall_clock64() - t0 > 600LL * 100000000LL) return -1;
    }
  }
}
// (returns false when the queue is empty.  Not a pointer: the descriptor sits at LDS address 0, which is what a null
// local-address-space pointer compares equal to.)
__device__ inline bool take_sequence(SeqDesc* s_desc, int* s_item, const SeqDesc* descs, const int* order, int n_items, int* counter,
                                   double* wd_pool, long long wd_stride, int* wi_pool, long long wi_stride, int slot) {
  __syncthreads();                                  // everybody is done with the previous descriptor
  if (threadIdx.x == 0) *s_item = atomicAdd(counter, 1);
  __syncthreads();
  const int item = *s_item;
  if (item >= n_items) return false;
  const int* src = (const int*)(descs + order[item]);
  LdsI* dst = (LdsI*)s_desc;
  for (int i = threadIdx.x; i < (int)(sizeof(SeqDesc) / 4); i += blockDim.x) dst[i] = src[i];
  __syncthreads();
  if (threadIdx.x == 0) {
    s_desc->wd = (GD*)(wd_pool + (long long)slot * wd_stride);
    s_desc->wi = (GI*)(wi_pool + (long long)slot * wi_stride);
  }
  __syncthreads();
  return true;
}

__global__ __launch_bounds__(CHD_MAX_THREADS) void chd_solve_kernel(const SeqDesc* descs, const int* order, int n_items, int* counter,
                                                                    double* wd_pool, long long wd_stride, int* wi_pool, long long wi_stride, int* slot_busy, int n_slots,
                                                                    int lds_doubles, double tol, int stall_window, int stage_first, int stage_last) {
  extern __shared__ double lds[];
  __shared__ SeqDesc s_desc;
  __shared__ Ctx s_ctx;
  __shared__ int s_item;
  __shared__ int s_slot;
  if (threadIdx.x == 0) s_slot = claim_slot(slot_busy, n_slots);
  __syncthreads();
  const int slot = s_slot;
  if (slot < 0) {                                   // no workspace: nothing can be solved here; the host sees the mark and fails the call
    if (threadIdx.x == 0) atomicAdd(counter, (int)CHD_SLOT_WAIT_SENTINEL);
    return;
  }
  for (;;) {
    if (!take_sequence(&s_desc, &s_item, descs, order, n_items, counter, wd_pool, wd_stride, wi_pool, wi_stride, slot)) break;
    run_sequence((QP)&s_desc, *(LCtx*)&s_ctx, (LdsD*)lds, lds_doubles, tol, stall_window, stage_first, stage_last);
  }
  __syncthreads();                                  // every wavefront's stores to the workspace are issued and complete (take_sequence ends on a barrier too)
  if (threadIdx.x == 0) __hip_atomic_store(&slot_busy[slot], 0, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
}

__global__ __launch_bounds__(CHD_MAX_THREADS) void chd_debug_eval_kernel(const SeqDesc* descs, const int* order, int* counter, double* wd_pool, int* wi_pool,
                                                                         int stage, const double* xin, int lds_doubles, double* f_out) {
  extern __shared__ double lds[];
  __shared__ SeqDesc s_desc;
  __shared__ Ctx s_ctx;
  __shared__ int s_item;
  if (!take_sequence(&s_desc, &s_item, descs, order, 1, counter, wd_pool, 0, wi_pool, 0, 0)) return;      // (alone on the device: slot 0, not claimed)
  debug_eval((QP)&s_desc, *(LCtx*)&s_ctx, stage, xin != nullptr, (LdsD*)lds, lds_doubles, (const GD*)xin, (const GD*)nullptr, f_out);
}

__global__ __launch_bounds__(CHD_MAX_THREADS) void chd_debug_linsolve_kernel(const SeqDesc* descs, const int* order, int* counter, double* wd_pool, int* wi_pool,
                                                                             int stage, int lds_doubles, double dw, double dval, int which, int reps,
                                                                             const double* rhs, double* x, double* out) {
  extern __shared__ double lds[];
  __shared__ SeqDesc s_desc;
  __shared__ Ctx s_ctx;
  __shared__ int s_item;
  if (!take_sequence(&s_desc, &s_item, descs, order, 1, counter, wd_pool, 0, wi_pool, 0, 0)) return;
  debug_linsolve((QP)&s_desc, *(LCtx*)&s_ctx, stage, (LdsD*)lds, lds_doubles, dw, dval, which, reps, (const GD*)rhs, (GD*)x, out);
}

#else
// ---- host emulation of the three kernels (CHD_HOST_EMU: chd_kernels.hpp compiles as a single-"thread" host function).  One std::thread per resident
// workgroup, the same protocol as the device code: claim a workspace slot (compare-and-swap, acquire), take sequences from the launch's queue until it is
// drained, release the slot.  Used by the sanitizer builds of the host side only.
static int claim_slot_emu(int* slot_busy, int n_slots, unsigned block) {
  int s = (int)(block % (unsigned)n_slots);
  for (;;) {
    int expected = 0;
    if (__atomic_compare_exchange_n(&slot_busy[s], &expected, 1, false, __ATOMIC_ACQUIRE, __ATOMIC_RELAXED)) return s;
    s = s + 1 == n_slots ? 0 : s + 1;
    std::this_thread::yield();
  }
}
static void chd_solve_kernel_emu(unsigned grid, const SeqDesc* descs, const int* order, int n_items, int* counter, double* wd_pool, long long wd_stride, int* wi_pool, long long wi_stride,
                                 int* slot_busy, int n_slots, int lds_doubles, double tol, int stall_window, int stage_first, int stage_last) {
  std::vector<std::thread> wg;
  for (unsigned blk = 0; blk < grid; ++blk)
    wg.emplace_back([=]() {
      const int slot = claim_slot_emu(slot_busy, n_slots, blk);
      std::vector<double> lds((size_t)lds_doubles, 0.0);
      for (;;) {
        const int item = __atomic_fetch_add(counter, 1, __ATOMIC_RELAXED);
        if (item >= n_items) break;
        SeqDesc d = descs[order[item]];
        d.wd = wd_pool + (long long)slot * wd_stride; d.wi = wi_pool + (long long)slot * wi_stride;
        Ctx ctx;
        run_sequence(&d, ctx, lds.data(), lds_doubles, tol, stall_window, stage_first, stage_last);
      }
      __atomic_store_n(&slot_busy[slot], 0, __ATOMIC_RELEASE);
    });
  for (auto& t : wg) t.join();
}
static void chd_debug_eval_kernel_emu(unsigned, const SeqDesc* descs, const int* order, int*, double* wd_pool, int* wi_pool, int stage, const double* xin, int lds_doubles, double* f_out) {
  SeqDesc d = descs[order[0]];
  d.wd = wd_pool; d.wi = wi_pool;
  Ctx ctx; std::vector<double> lds((size_t)lds_doubles, 0.0);
  debug_eval(&d, ctx, stage, xin != nullptr, lds.data(), lds_doubles, xin, nullptr, f_out);
}
static void chd_debug_linsolve_kernel_emu(unsigned, const SeqDesc* descs, const int* order, int*, double* wd_pool, int* wi_pool, int stage, int lds_doubles, double dw, double dval, int which, int reps,
                                          const double* rhs, double* x, double* out) {
  SeqDesc d = descs[order[0]];
  d.wd = wd_pool; d.wi = wi_pool;
  Ctx ctx; std::vector<double> lds((size_t)lds_doubles, 0.0);
  debug_linsolve(&d, ctx, stage, lds.data(), lds_doubles, dw, dval, which, reps, rhs, x, out);
}
#endif

// A handle owns ONE set of workspaces -- one per workgroup that can be resident, claimed and released by the workgroups themselves (claim_slot above) --
// and CHD_N_POOLS "lanes": a stream and a set of reusable device / page-locked buffers for one chunk of sequences each.  A launch uses one lane, so up to
// CHD_N_POOLS persistent launches can be queued: in a pipelined call the host builds and launches chunk after chunk as fast as it can, and the
// device works through them in order -- a workgroup of chunk k + 1 starts as soon as a compute unit has no sequence of chunk k left to take.  The split
// interface (chd_batch_solve) uses lane 0.  (Rounds 3-4 had a workspace pool per lane: four launches in flight at 4 x the memory, and a lane could not be
// reused before the last straggler of its previous launch had finished -- 0.84 of the solve-only rate; profiles/r04_pipeline.md.)
#define CHD_N_POOLS 4          // = the hardware queues a process gets by default (GPU_MAX_HW_QUEUES): streams beyond that share a queue, and a persistent launch blocks the queue it is in until its last straggler ends (measured: eight lanes 0.60 of the solve-only rate)
struct chd_handle {
  int device = 0;
  hipStream_t stream[CHD_N_POOLS] = {};
  chd_config cfg;
  std::string err;
  int lds_bytes = 0;
  int threads = CHD_MAX_THREADS;
  int n_wg = 0;                        // resident workgroups of a launch = workspace slots
  // workgroup workspaces (grow-only; reallocated only while no launch is in flight) and their busy flags
  double* d_wd = nullptr; int* d_wi = nullptr; int* d_slots = nullptr;
  long long wd_stride = 0, wi_stride = 0;
  chd_call_stats call{};               // accounting of the last chd_phys_solve_batch / chd_phys_solve_dirs
  // device buffers of the pipelined path, one set per pool, grow-only and reused chunk after chunk: no hipMalloc / hipFree while launches are in flight
  // (measured, round 4: allocating per chunk serialised the chunks -- 214 ms of "upload" per chunk, the whole call at 0.72 of the solve-only rate)
  struct PoolBufs {
    double* d_cd = nullptr; int* d_ci = nullptr; double* d_od = nullptr; int* d_oi = nullptr; SeqDesc* d_descs = nullptr; int* d_order = nullptr; int* d_counter = nullptr; double* d_f = nullptr;
    long long cap_cd = 0, cap_ci = 0, cap_od = 0, cap_oi = 0, cap_seq = 0;
    // page-locked host staging: copies from / to it run on the DMA engines.  (A copy from pageable memory is done by a copy KERNEL, and the persistent
    // workgroups hold every register of every compute unit: measured in round 4, the upload of chunk k + 1 then waits until chunk k's queue drains.)
    void* pin[6] = {}; size_t pin_cap[6] = {};      // 0 cd, 1 ci, 2 descs + order, 3 od, 4 oi, 5 scratch
  } pb[CHD_N_POOLS];
};

struct chd_batch {
  int B = 0;
  int pool = 0;                          // workspace pool / stream of this batch's launches
  std::vector<SeqModel> models;
  std::vector<SeqDesc> descs;            // host copy with DEVICE pointers
  std::vector<char> ok;                  // 0: rejected at set-up (build_err), never queued
  std::vector<std::string> build_err;
  std::vector<int> order;                // queue order of the solvable sequences (longest first)
  std::vector<int> fallback;             // the sequences of the stage-4 launch (kept here: the copy to the device is asynchronous)
  std::vector<long long> off_cd, off_ci;
  long long tot_cd = 0, tot_ci = 0, od_stride = 0, oi_stride = 0;      // results: one fixed-size slot per sequence (strided copies of the statistics)
  long long wd_need = 0, wi_need = 0;
  double *d_cd = nullptr, *d_od = nullptr, *d_f = nullptr, *d_x = nullptr;
  int *d_ci = nullptr, *d_oi = nullptr, *d_order = nullptr, *d_counter = nullptr;
  long long x_cap = 0;
  SeqDesc* d_descs = nullptr;
  std::vector<double> h_od;
  std::vector<int> h_oi;
  bool solved = false, fetched = false;
  bool owns_device = true;               // false: the device buffers belong to the handle's pool (pipelined path)
  double build_cpu_ms = 0;               // host time of the table builder, summed over the sequences (thread time, not wall)
  chd_batch_stats stats{};
  hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};
};

// Error text.  The finisher threads of a pipelined call work on the SHARED handle (its pools' buffers must be the real ones: a finisher that grew a page-locked
// buffer in a private copy of the handle left the real handle with a dangling pointer -- round 4's use-after-free); what they must not share is the error
// string, so a thread can redirect its error text to a string of its own (the chunk's).
static thread_local std::string* tl_err_sink = nullptr;
static int fail(chd_handle* h, const std::string& msg) { if (tl_err_sink) *tl_err_sink = msg; else if (h) h->err = msg; return -1; }
#define HIP_TRY(h, call)                                                                                         \
  do { hipError_t e_ = (call); if (e_ != hipSuccess) return fail(h, std::string(#call) + ": " + hipGetErrorString(e_)); } while (0)

// run fn(i) for i in [0, n) on the host's cores (text parsing / formatting of thousands of directories would otherwise take
// as long as the solve itself: ~1 ms per file set)
// Host threads one process may use: at most 32, and -- one process per GPU on a shared host -- the host's hardware threads divided by the ranks the launcher
// started on it (LOCAL_WORLD_SIZE: set by torch.distributed.run and by bench.py's own launcher; 8 ranks x 32 builder threads would otherwise oversubscribe a
// 128-thread host).  CHD_HOST_THREADS overrides.
static unsigned host_thread_cap() {
  if (const char* e = std::getenv("CHD_HOST_THREADS")) { const int v = std::atoi(e); if (v > 0) return (unsigned)v; }
  unsigned hc = std::thread::hardware_concurrency();
  if (hc == 0) hc = 4;
  int lw = 1;
  if (const char* e = std::getenv("LOCAL_WORLD_SIZE")) { lw = std::atoi(e); if (lw < 1) lw = 1; }
  unsigned nt = hc / (unsigned)lw;
  if (nt < 2) nt = 2;
  if (nt > 32) nt = 32;
  return nt;
}
template <class F>
static void host_parallel_for(int n, F fn) {
  unsigned nt = host_thread_cap();
  if ((int)nt > n) nt = (unsigned)n;
  if (nt <= 1) { for (int i = 0; i < n; ++i) fn(i); return; }
  std::vector<std::thread> pool;
  for (unsigned t = 0; t < nt; ++t) pool.emplace_back([&, t]() { for (int i = (int)t; i < n; i += (int)nt) fn(i); });
  for (auto& th : pool) th.join();
}

static double now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
static unsigned host_threads(int n) {
  unsigned nt = host_thread_cap();
  if ((int)nt > n) nt = (unsigned)(n > 0 ? n : 1);
  return nt;
}

// workspaces for `n_wg` resident workgroups of at least (wd_need, wi_need) elements each (+ `headroom` eighths when they have to be (re)allocated: the
// pipelined path asks for 2/8, because a sequence that needs a few per cent more than its predecessors must not stall the pipeline -- measured in round 4:
// every new maximum drained all launches, 1.2 s each time).  The caller makes sure NO launch is in flight when the workspaces have to grow
// (workspace_fits tells).
static bool workspace_fits(const chd_handle* h, long long wd_need, long long wi_need) {
  auto al = [](long long v) { return (v + 63) & ~63LL; };
  return h->d_wd && al(wd_need) <= h->wd_stride && al(wi_need) <= h->wi_stride;
}
static int ensure_workspace(chd_handle* h, long long wd_need, long long wi_need, int headroom = 0) {
  auto al = [](long long v) { return (v + 63) & ~63LL; };
  if (workspace_fits(h, wd_need, wi_need)) return 0;
  if (h->d_wd) {
    HIP_TRY(h, hipDeviceSynchronize());
    (void)hipFree(h->d_wd); (void)hipFree(h->d_wi);
    h->d_wd = nullptr; h->d_wi = nullptr;
  }
  h->wd_stride = std::max(h->wd_stride, al(wd_need + wd_need / 8 * headroom)); h->wi_stride = std::max(h->wi_stride, al(wi_need + wi_need / 8 * headroom));
  hipError_t e = hipMalloc((void**)&h->d_wd, (size_t)h->wd_stride * 8 * h->n_wg);
  if (e == hipSuccess) e = hipMalloc((void**)&h->d_wi, (size_t)h->wi_stride * 4 * h->n_wg);
  if (e != hipSuccess && headroom > 0) {          // not with head-room: exactly what is needed
    (void)hipFree(h->d_wd); h->d_wd = nullptr; h->d_wi = nullptr;
    h->wd_stride = al(wd_need); h->wi_stride = al(wi_need);
    e = hipMalloc((void**)&h->d_wd, (size_t)h->wd_stride * 8 * h->n_wg);
    if (e == hipSuccess) e = hipMalloc((void**)&h->d_wi, (size_t)h->wi_stride * 4 * h->n_wg);
  }
  if (e == hipSuccess && !h->d_slots) e = hipMalloc((void**)&h->d_slots, sizeof(int) * (size_t)h->n_wg);
  if (e != hipSuccess) {
    (void)hipFree(h->d_wd); h->d_wd = nullptr; h->d_wi = nullptr;
    const long long mib = (h->wd_stride * 8 + h->wi_stride * 4) * h->n_wg >> 20;
    h->wd_stride = h->wi_stride = 0;
    return fail(h, std::string("workspace allocation (") + std::to_string(mib) + " MiB): " + hipGetErrorString(e));
  }
  // (no clearing needed for correctness: a workgroup zeroes / initialises what it reads, sequence by sequence and stage by stage; the slot flags must be clear)
  HIP_TRY(h, hipMemsetAsync(h->d_wd, 0, (size_t)h->wd_stride * 8 * h->n_wg, h->stream[0]));
  HIP_TRY(h, hipMemsetAsync(h->d_wi, 0, (size_t)h->wi_stride * 4 * h->n_wg, h->stream[0]));
  HIP_TRY(h, hipMemsetAsync(h->d_slots, 0, sizeof(int) * (size_t)h->n_wg, h->stream[0]));
  HIP_TRY(h, hipStreamSynchronize(h->stream[0]));          // (the launches of the other lanes do not wait for stream 0)
  return 0;
}

// ---- host half of an upload: the per-sequence NLP structure tables (phys_optim.cpp:428-540, nlp_formulation.cpp:79-203), on `nt` threads.  A sequence
// whose set-up fails (too short, inconsistent contact schedule, degenerate floor normal ...) is rejected on its own -- the reference runs one process per
// video, so a bad video only loses itself (run_phys_mocap.py:159-174) -- and the rest of the batch is solved.  No HIP call in here.
static chd_batch* batch_build(const chd_config& cfg, int B, const chd_seq_in* in, unsigned nt) {
  chd_batch* b = new chd_batch();
  b->B = B;
  b->models.resize(B);
  b->ok.assign(B, 1); b->build_err.assign(B, std::string());
  if (nt < 1) nt = 1;
  if ((int)nt > B) nt = (unsigned)B;
  std::vector<double> cpu(nt, 0.0);
  auto body = [&](unsigned t) {
    const double t0 = now_ms();
    for (int i = (int)t; i < B; i += (int)nt) {
      try { b->models[i].build(in[i], cfg); }
      catch (const std::exception& e) { b->ok[i] = 0; b->build_err[i] = e.what(); }
      catch (...) { b->ok[i] = 0; b->build_err[i] = "set-up failed"; }
    }
    cpu[t] = now_ms() - t0;
  };
  if (nt == 1) body(0);
  else {
    std::vector<std::thread> pool;
    for (unsigned t = 0; t < nt; ++t) pool.emplace_back(body, t);
    for (auto& th : pool) th.join();
  }
  for (double v : cpu) b->build_cpu_ms += v;
  for (int i = 0; i < B; ++i) if (b->ok[i]) b->order.push_back(i);
  // queue order: longest sequences first, and among equal lengths the ones with the most contact phases first (a launch lasts as long as its last
  // sequence: the expensive ones must not start last.  Iterations of the duration stage grow with the number of duration variables.)
  std::stable_sort(b->order.begin(), b->order.end(), [&](int a, int c2) {
    const SeqDesc& da = b->models[a].d; const SeqDesc& dc = b->models[c2].d;
    if (da.F != dc.F) return da.F > dc.F;
    return da.tot_phases > dc.tot_phases;
  });
  auto al = [](long long v) { return (v + 31) & ~31LL; };
  b->off_cd.assign(B, 0); b->off_ci.assign(B, 0);
  for (int i = 0; i < B; ++i) {
    if (!b->ok[i]) continue;
    SeqModel& M = b->models[i];
    b->off_cd[i] = b->tot_cd; b->tot_cd += al((long long)M.cd.size());
    b->off_ci[i] = b->tot_ci; b->tot_ci += al((long long)M.ci.size());
    b->wd_need = std::max(b->wd_need, M.wd_size); b->wi_need = std::max(b->wi_need, M.wi_size);
    b->od_stride = std::max(b->od_stride, al(out_d_size(M.d.cap, M.d.tot_entries + M.d.tot_phases)));
    b->oi_stride = std::max(b->oi_stride, al(out_i_size(M.d.cap)));
  }
  return b;
}

// ---- device half of an upload: pools, descriptors, result slots (on the stream of the batch's pool)
// make sure pool `pool`'s reusable buffers hold a batch of these sizes (grow-only, 1/8 head-room; reallocation only while nothing runs on the pool)
static int ensure_pool_bufs(chd_handle* h, int pool, long long n_cd, long long n_ci, long long n_od, long long n_oi, long long n_seq) {
  chd_handle::PoolBufs& P = h->pb[pool];
  auto grow = [&](void** p, long long& cap, long long need, size_t elem) -> hipError_t {
    if (need <= cap && *p) return hipSuccess;
    if (*p) (void)hipFree(*p);
    *p = nullptr;
    cap = need + need / 8 + 64;
    return hipMalloc(p, (size_t)cap * elem);
  };
  hipError_t e;
  if ((e = grow((void**)&P.d_cd, P.cap_cd, n_cd, 8)) != hipSuccess || (e = grow((void**)&P.d_ci, P.cap_ci, n_ci, 4)) != hipSuccess ||
      (e = grow((void**)&P.d_od, P.cap_od, n_od, 8)) != hipSuccess || (e = grow((void**)&P.d_oi, P.cap_oi, n_oi, 4)) != hipSuccess)
    return fail(h, std::string("pool buffers: ") + hipGetErrorString(e));
  if (n_seq > P.cap_seq || !P.d_descs) {
    (void)hipFree(P.d_descs); (void)hipFree(P.d_order); P.d_descs = nullptr; P.d_order = nullptr;
    P.cap_seq = n_seq + n_seq / 8 + 8;
    if ((e = hipMalloc((void**)&P.d_descs, sizeof(SeqDesc) * (size_t)P.cap_seq)) != hipSuccess || (e = hipMalloc((void**)&P.d_order, sizeof(int) * (size_t)P.cap_seq)) != hipSuccess)
      return fail(h, std::string("pool buffers: ") + hipGetErrorString(e));
  }
  if (!P.d_counter && ((e = hipMalloc((void**)&P.d_counter, 64)) != hipSuccess || (e = hipMalloc((void**)&P.d_f, 64)) != hipSuccess)) return fail(h, std::string("pool buffers: ") + hipGetErrorString(e));
  return 0;
}

static void* pin_get(chd_handle* h, int pool, int which, size_t bytes) {
  chd_handle::PoolBufs& P = h->pb[pool];
  if (bytes <= P.pin_cap[which] && P.pin[which]) return P.pin[which];
  if (P.pin[which]) (void)hipHostFree(P.pin[which]);
  P.pin[which] = nullptr; P.pin_cap[which] = bytes + bytes / 8 + 4096;
  if (hipHostMalloc(&P.pin[which], P.pin_cap[which], hipHostMallocDefault) != hipSuccess) { P.pin[which] = nullptr; P.pin_cap[which] = 0; }
  return P.pin[which];
}
// small copies of a batch's finisher / fallback path: through the pool's page-locked scratch when the batch lives in a pool (DMA engine, see above)
static hipError_t copy_d2h(chd_handle* h, chd_batch* b, void* dst, const void* src, size_t bytes) {
  if (b->owns_device) return hipMemcpy(dst, src, bytes, hipMemcpyDeviceToHost);
  void* sc = pin_get(h, b->pool, 5, bytes);
  if (!sc) return hipErrorOutOfMemory;
  hipError_t e = hipMemcpyAsync(sc, src, bytes, hipMemcpyDeviceToHost, h->stream[b->pool]);
  if (e == hipSuccess) e = hipStreamSynchronize(h->stream[b->pool]);
  if (e == hipSuccess) std::memcpy(dst, sc, bytes);
  return e;
}
static hipError_t copy_d2h_2d(chd_handle* h, chd_batch* b, void* dst, size_t dpitch, const void* src, size_t spitch, size_t width, size_t height) {
  if (b->owns_device) return hipMemcpy2D(dst, dpitch, src, spitch, width, height, hipMemcpyDeviceToHost);
  void* sc = pin_get(h, b->pool, 5, width * height);
  if (!sc) return hipErrorOutOfMemory;
  hipError_t e = hipMemcpy2DAsync(sc, width, src, spitch, width, height, hipMemcpyDeviceToHost, h->stream[b->pool]);
  if (e == hipSuccess) e = hipStreamSynchronize(h->stream[b->pool]);
  if (e == hipSuccess) for (size_t r = 0; r < height; ++r) std::memcpy((char*)dst + r * dpitch, (char*)sc + r * width, width);
  return e;
}
static hipError_t copy_h2d(chd_handle* h, chd_batch* b, void* dst, const void* src, size_t bytes) {
  if (b->owns_device) return hipMemcpy(dst, src, bytes, hipMemcpyHostToDevice);
  void* sc = pin_get(h, b->pool, 5, bytes);
  if (!sc) return hipErrorOutOfMemory;
  std::memcpy(sc, src, bytes);
  hipError_t e = hipMemcpyAsync(dst, sc, bytes, hipMemcpyHostToDevice, h->stream[b->pool]);
  if (e == hipSuccess) e = hipStreamSynchronize(h->stream[b->pool]);
  return e;
}

// ---- device half of an upload: pools, descriptors, result slots (on the stream of the batch's pool).  `pooled`: the buffers are the handle's reusable ones
static int batch_to_device(chd_handle* h, chd_batch* b, int pool, bool pooled = false) {
  const int B = b->B;
  b->pool = pool;
  hipStream_t st = h->stream[pool];
  auto bail = [&](const char* what, hipError_t e) { return fail(h, std::string(what) + ": " + hipGetErrorString(e)); };
  hipError_t e;
  if (pooled) {
    if (ensure_pool_bufs(h, pool, std::max<long long>(b->tot_cd, 1), std::max<long long>(b->tot_ci, 1), b->od_stride * B, b->oi_stride * B, B) != 0) return -1;
    const chd_handle::PoolBufs& P = h->pb[pool];
    b->owns_device = false;
    b->d_cd = P.d_cd; b->d_ci = P.d_ci; b->d_od = P.d_od; b->d_oi = P.d_oi; b->d_descs = P.d_descs; b->d_order = P.d_order; b->d_counter = P.d_counter; b->d_f = P.d_f;
  } else {
  if ((e = hipMalloc((void**)&b->d_cd, std::max<long long>(b->tot_cd, 1) * 8)) != hipSuccess) return bail("hipMalloc cd", e);
  if ((e = hipMalloc((void**)&b->d_ci, std::max<long long>(b->tot_ci, 1) * 4)) != hipSuccess) return bail("hipMalloc ci", e);
  if ((e = hipMalloc((void**)&b->d_od, b->od_stride * 8 * B)) != hipSuccess) return bail("hipMalloc od", e);
  if ((e = hipMalloc((void**)&b->d_oi, b->oi_stride * 4 * B)) != hipSuccess) return bail("hipMalloc oi", e);
  if ((e = hipMalloc((void**)&b->d_descs, sizeof(SeqDesc) * B)) != hipSuccess) return bail("hipMalloc descs", e);
  if ((e = hipMalloc((void**)&b->d_order, sizeof(int) * B)) != hipSuccess) return bail("hipMalloc order", e);
  if ((e = hipMalloc((void**)&b->d_counter, 64)) != hipSuccess) return bail("hipMalloc counter", e);
  if ((e = hipMalloc((void**)&b->d_f, 64)) != hipSuccess) return bail("hipMalloc f", e);
  }
  if ((e = hipMemsetAsync(b->d_od, 0, b->od_stride * 8 * B, st)) != hipSuccess) return bail("memset od", e);
  if ((e = hipMemsetAsync(b->d_oi, 0, b->oi_stride * 4 * B, st)) != hipSuccess) return bail("memset oi", e);
  // ---- stage pools through one staging buffer each (page-locked and owned by the pool when `pooled`: the copies are then asynchronous)
  {
    std::vector<double> hcd_v; std::vector<int> hci_v;
    double* hcd; int* hci; SeqDesc* hdesc;
    if (pooled) {
      hcd = (double*)pin_get(h, pool, 0, (size_t)std::max<long long>(b->tot_cd, 1) * 8); hci = (int*)pin_get(h, pool, 1, (size_t)std::max<long long>(b->tot_ci, 1) * 4);
      hdesc = (SeqDesc*)pin_get(h, pool, 2, sizeof(SeqDesc) * (size_t)B + sizeof(int) * (size_t)B);
      if (!hcd || !hci || !hdesc) return fail(h, "page-locked staging buffers: allocation failed");
      // the staging the chunk's FINISHER thread will need (results, statistics, the stage-4 fallback's table regions) is sized here, on the calling thread and
      // from this chunk's real strides, so that the finisher never reallocates page-locked memory while other lanes have launches in flight
      size_t scratch = std::max<size_t>(sizeof(SeqDesc), (size_t)B * N_STAGES * RS_STRIDE * 8);
      for (int i = 0; i < B; ++i) {
        if (!b->ok[i]) continue;
        const SeqModel& M = b->models[i];
        const StageDesc& S5 = M.d.st[5];
        scratch = std::max(scratch, (size_t)(S5.o_rcnt + (S5.n + M.stage_m_cap[5]) - S5.o_pos_var) * 4);
        scratch = std::max(scratch, (size_t)(S5.o_task_t + M.stage_task_cap[5] - S5.o_cl) * 8);
        scratch = std::max(scratch, (size_t)M.d.tot_phases * 8);
      }
      if (!pin_get(h, pool, 3, (size_t)b->od_stride * B * 8) || !pin_get(h, pool, 4, (size_t)b->oi_stride * B * 4) || !pin_get(h, pool, 5, scratch))
        return fail(h, "page-locked result buffers: allocation failed");
    } else { hcd_v.assign(b->tot_cd, 0.0); hci_v.assign(b->tot_ci, 0); hcd = hcd_v.data(); hci = hci_v.data(); hdesc = nullptr; }
    for (int i = 0; i < B; ++i) {
      if (!b->ok[i]) continue;
      std::copy(b->models[i].cd.begin(), b->models[i].cd.end(), hcd + b->off_cd[i]);
      std::copy(b->models[i].ci.begin(), b->models[i].ci.end(), hci + b->off_ci[i]);
    }
    if ((e = hipMemcpyAsync(b->d_cd, hcd, b->tot_cd * 8, hipMemcpyHostToDevice, st)) != hipSuccess) return bail("copy cd", e);
    if ((e = hipMemcpyAsync(b->d_ci, hci, b->tot_ci * 4, hipMemcpyHostToDevice, st)) != hipSuccess) return bail("copy ci", e);
    b->descs.resize(B);
    for (int i = 0; i < B; ++i) {
      SeqDesc dd = b->models[i].d;
      dd.cd = (const GD*)(b->d_cd + b->off_cd[i]); dd.ci = (const GI*)(b->d_ci + b->off_ci[i]);
      dd.wd = nullptr; dd.wi = nullptr;            // the workgroup that takes the sequence fills in its own workspace
      dd.out_d = (GD*)(b->d_od + b->od_stride * i); dd.out_i = (GI*)(b->d_oi + b->oi_stride * i);
      b->descs[i] = dd;
    }
    const SeqDesc* dsrc = b->descs.data();
    if (pooled) { std::memcpy(hdesc, b->descs.data(), sizeof(SeqDesc) * (size_t)B); dsrc = hdesc; }
    if ((e = hipMemcpyAsync(b->d_descs, dsrc, sizeof(SeqDesc) * B, hipMemcpyHostToDevice, st)) != hipSuccess) return bail("copy descs", e);
    if (!pooled && (e = hipStreamSynchronize(st)) != hipSuccess) return bail("sync", e);          // (the pageable staging buffers go out of scope)
  }
  for (int k = 0; k < 4; ++k) if ((e = hipEventCreate(&b->ev[k])) != hipSuccess) return bail("hipEventCreate", e);
  return 0;
}

// one persistent launch over `items` (indices into the batch), asynchronous: e1 is recorded behind the kernel
static int launch_queue(chd_handle* h, chd_batch* b, const std::vector<int>& items, int stage_first, int stage_last, hipEvent_t e0, hipEvent_t e1) {
  hipStream_t st = h->stream[b->pool];
  const int* isrc = items.data();
  if (!b->owns_device) {            // (behind the descriptors in the pool's page-locked staging buffer 2)
    int* po = (int*)((char*)pin_get(h, b->pool, 2, sizeof(SeqDesc) * (size_t)b->B + sizeof(int) * (size_t)b->B) + sizeof(SeqDesc) * (size_t)b->B);
    std::memcpy(po, items.data(), items.size() * sizeof(int)); isrc = po;
  }
  HIP_TRY(h, hipMemcpyAsync(b->d_order, isrc, items.size() * sizeof(int), hipMemcpyHostToDevice, st));
  HIP_TRY(h, hipMemsetAsync(b->d_counter, 0, sizeof(int), st));
  const unsigned grid = (unsigned)std::min<size_t>(items.size(), (size_t)h->n_wg);
  HIP_TRY(h, hipEventRecord(e0, st));
  (void)hipGetLastError();      // an error another library of the process left behind in this thread is not this launch's
  hipLaunchKernelGGL(chd_solve_kernel, dim3(grid), dim3(h->threads), h->lds_bytes, st, b->d_descs, (const int*)b->d_order, (int)items.size(),
                     b->d_counter, h->d_wd, h->wd_stride, h->d_wi, h->wi_stride, h->d_slots, h->n_wg, h->lds_bytes / 8, h->cfg.tol, h->cfg.stall_window, stage_first, stage_last);
  HIP_TRY(h, hipGetLastError());
  HIP_TRY(h, hipEventRecord(e1, st));
  return 0;          // (asynchronous: the kernel is waited for through e1; `items` must stay alive until then -- the callers pass vectors owned by the batch)
}

// A finished launch must have drained its queue: every workgroup leaves after exactly one failed take, so the counter ends at n_items + grid.  Anything else
// means sequences were never solved (their zeroed result slots would read as "converged"): fail loudly.
static int launch_drained(chd_handle* h, chd_batch* b, size_t n_items) {
  int cnt = -1;
  HIP_TRY(h, copy_d2h(h, b, &cnt, b->d_counter, sizeof(int)));
  const int grid = (int)std::min<size_t>(n_items, (size_t)h->n_wg);
  if (cnt >= (int)CHD_SLOT_WAIT_SENTINEL)
    return fail(h, "a workgroup of the solver launch waited 10 minutes for a workspace slot: a slot flag was left set by a faulted or aborted earlier launch on this handle -- destroy the handle and create a new one");
  if (cnt != (int)n_items + grid)
    return fail(h, "solver launch left its queue undrained: counter " + std::to_string(cnt) + ", expected " + std::to_string((int)n_items + grid) + " (" + std::to_string(n_items) + " sequences, " + std::to_string(grid) + " workgroups)");
  return 0;
}

// per-sequence statistics block (stages + phase timers are fetched with two strided copies)
static int fetch_stats(chd_handle* h, chd_batch* b, std::vector<double>& st) {
  const size_t w = (size_t)N_STAGES * RS_STRIDE;
  st.resize(w * b->B);
  HIP_TRY(h, copy_d2h_2d(h, b, st.data(), w * 8, b->d_od, (size_t)b->od_stride * 8, w * 8, (size_t)b->B));
  return 0;
}

// ---- launch 1: stages 1.1, 1.2, 2.1, 2.2, 3 for every sequence
static int solve_launch_main(chd_handle* h, chd_batch* b) {
  b->stats = chd_batch_stats{};
  b->fetched = false; b->solved = false;
  HIP_TRY(h, hipMemsetAsync(b->d_od, 0, b->od_stride * 8 * b->B, h->stream[b->pool]));
  return launch_queue(h, b, b->order, 0, 4, b->ev[0], b->ev[1]);
}

// ---- waits for launch 1, runs the stage-4 fallback where stage 3 failed (phys_optim.cpp:714), accounting
static int solve_finish(chd_handle* h, chd_batch* b) {
  HIP_TRY(h, hipEventSynchronize(b->ev[1]));
  float ms = 0;
  HIP_TRY(h, hipEventElapsedTime(&ms, b->ev[0], b->ev[1]));
  b->stats.kernel_ms[0] = ms;
  if (launch_drained(h, b, b->order.size()) != 0) return -1;
  auto t0 = std::chrono::steady_clock::now();
  std::vector<double> st;
  if (fetch_stats(h, b, st) != 0) return -1;
  const size_t sw = (size_t)N_STAGES * RS_STRIDE;
  std::vector<int>& idx = b->fallback;
  idx.clear();
  for (int i : b->order) if ((int)st[sw * i + 4 * RS_STRIDE + RS_STATUS] != 0) idx.push_back(i);
  std::sort(idx.begin(), idx.end());
  b->stats.n_fallback = (int)idx.size();
  if (!idx.empty()) {
    // durations left by stage 3 (kept with the sequence's results) -> rebuild the tables of the fallback stage on the host
    std::vector<std::vector<double>> ph(idx.size());
    for (size_t k = 0; k < idx.size(); ++k) {
      const SeqModel& M = b->models[idx[k]];
      ph[k].resize(M.d.tot_phases);
      HIP_TRY(h, copy_d2h(h, b, ph[k].data(), b->d_od + b->od_stride * idx[k] + out_d_state_off(M.d.cap) + 2LL * (M.d.tot_entries + M.d.tot_phases) + M.d.tot_entries, ph[k].size() * 8));      // (state slot 2: what stage 3 left)
    }
    const unsigned nt = host_threads((int)idx.size());
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
        const long long i0 = S.o_pos_var, i1 = S.o_rcnt + (S.n + M.stage_m_cap[5]);
        const long long d0 = S.o_cl, d1 = S.o_task_t + (long long)M.stage_task_cap[5];
        HIP_TRY(h, copy_h2d(h, b, b->d_ci + b->off_ci[i] + i0, M.ci.data() + i0, (i1 - i0) * 4));
        HIP_TRY(h, copy_h2d(h, b, b->d_cd + b->off_cd[i] + d0, M.cd.data() + d0, (d1 - d0) * 8));
      }
      HIP_TRY(h, copy_h2d(h, b, b->d_descs + i, &b->descs[i], sizeof(SeqDesc)));
    }
    b->stats.host_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    if (launch_queue(h, b, idx, 5, 5, b->ev[2], b->ev[3]) != 0) return -1;
    HIP_TRY(h, hipEventSynchronize(b->ev[3]));
    HIP_TRY(h, hipEventElapsedTime(&ms, b->ev[2], b->ev[3]));
    b->stats.kernel_ms[1] = ms;
    if (launch_drained(h, b, idx.size()) != 0) return -1;
    if (fetch_stats(h, b, st) != 0) return -1;
  } else {
    b->stats.host_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
  }
  b->solved = true;
  // ---- accounting
  std::vector<double> tm((size_t)24 * b->B);
  {
    // the timers sit behind the snapshots, whose size depends on the sequence: equal capacities (the usual case) take one strided copy
    bool same = true;
    for (int i : b->order) same = same && b->models[i].d.cap == b->models[b->order[0]].d.cap;
    if (same) {
      const long long off = N_STAGES * RS_STRIDE + 3LL * 10 * b->models[b->order[0]].d.cap * 3;
      HIP_TRY(h, copy_d2h_2d(h, b, tm.data(), 24 * 8, b->d_od + off, (size_t)b->od_stride * 8, 24 * 8, (size_t)b->B));
    } else {
      for (int i : b->order)
        HIP_TRY(h, copy_d2h(h, b, tm.data() + 24 * (size_t)i, b->d_od + b->od_stride * i + N_STAGES * RS_STRIDE + 3LL * 10 * b->models[i].d.cap * 3, 24 * 8));
    }
  }
  for (int i : b->order) {
    const double* s = st.data() + sw * i;
    for (int stg = 0; stg < N_STAGES; ++stg) {
      if (stg == 5 && (int)s[4 * RS_STRIDE + RS_STATUS] == 0) continue;
      const double it = s[stg * RS_STRIDE + RS_ITERS];
      b->stats.total_iters += (long long)it;
      b->stats.total_factorizations += (long long)s[stg * RS_STRIDE + RS_NFACT];
      b->stats.alg_bytes += it * b->models[i].alg_bytes_iter[stg];
      if ((int)s[stg * RS_STRIDE + RS_AUX] != 0) b->stats.n_stalled += 1;
    }
    const double* t = tm.data() + 24 * (size_t)i;     // 100 MHz ticks
    for (int k = 0; k < 24; ++k) b->stats.phase_ms[k] += t[k] * 1e-5;
    if (t[5] * 1e-5 > b->stats.max_seq_ms) b->stats.max_seq_ms = t[5] * 1e-5;
  }
  b->stats.n_rejected = b->B - (int)b->order.size();
  b->stats.n_workgroups = (int)std::min<size_t>(b->order.size(), (size_t)h->n_wg);
  return 0;
}

// results of a solved batch -> caller arrays
static int batch_fetch(chd_handle* h, chd_batch* b, chd_seq_out* out) {
  const double* h_od = nullptr; const int* h_oi = nullptr;
  if (!b->fetched && !b->owns_device) {            // pool: straight into the page-locked buffers (read below before the pool is handed on)
    double* pod = (double*)pin_get(h, b->pool, 3, (size_t)b->od_stride * b->B * 8); int* poi = (int*)pin_get(h, b->pool, 4, (size_t)b->oi_stride * b->B * 4);
    if (!pod || !poi) return fail(h, "page-locked result buffers: allocation failed");
    HIP_TRY(h, hipMemcpyAsync(pod, b->d_od, (size_t)b->od_stride * b->B * 8, hipMemcpyDeviceToHost, h->stream[b->pool]));
    HIP_TRY(h, hipMemcpyAsync(poi, b->d_oi, (size_t)b->oi_stride * b->B * 4, hipMemcpyDeviceToHost, h->stream[b->pool]));
    HIP_TRY(h, hipStreamSynchronize(h->stream[b->pool]));
    h_od = pod; h_oi = poi;
  } else {
    if (!b->fetched) {
      b->h_od.resize((size_t)b->od_stride * b->B); b->h_oi.resize((size_t)b->oi_stride * b->B);
      HIP_TRY(h, hipMemcpy(b->h_od.data(), b->d_od, b->h_od.size() * 8, hipMemcpyDeviceToHost));
      HIP_TRY(h, hipMemcpy(b->h_oi.data(), b->d_oi, b->h_oi.size() * 4, hipMemcpyDeviceToHost));
      b->fetched = true;
    }
    h_od = b->h_od.data(); h_oi = b->h_oi.data();
  }
  for (int i = 0; i < b->B; ++i) {
    chd_seq_out& o = out[i];
    if (!b->ok[i]) {                 // rejected at set-up: nothing was solved
      for (int s = 0; s < N_STAGES; ++s) { o.stage_status[s] = -4; o.stage_iters[s] = 0; o.stage_stalled[s] = 0; o.stage_factorizations[s] = 0; o.stage_kkt_error[s] = o.stage_constr_viol[s] = o.stage_objective[s] = 0.0; }
      o.dynamics_succeed = 0; o.durations_succeed = 0;
      o.n_vars = o.n_rows = o.kkt_dim = o.kkt_halfband = o.kkt_border = 0; o.nnz_jac = 0;
      for (int s = 0; s < CHD_N_SNAPSHOTS; ++s) { o.snap[s].n_samples = 0; o.snap[s].num_frames_header = 0; }
      continue;
    }
    const SeqModel& M = b->models[i];
    const int cap = M.d.cap;
    const double* od = h_od + b->od_stride * i;
    const int* oi = h_oi + b->oi_stride * i;
    const bool fb = (int)od[4 * RS_STRIDE + RS_STATUS] != 0;
    for (int s = 0; s < N_STAGES; ++s) {
      const double* r = od + s * RS_STRIDE;
      const bool ran = (s < 5) || fb;
      o.stage_status[s] = ran ? (int)r[RS_STATUS] : 9;
      o.stage_iters[s] = ran ? (int)r[RS_ITERS] : 0;
      o.stage_stalled[s] = ran ? (int)r[RS_AUX] : 0;
      o.stage_factorizations[s] = ran ? (int)r[RS_NFACT] : 0;
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

// ---- the pipelined whole-call path (chd_phys_solve_batch, chd_phys_solve_dirs).
// The B sequences are cut into chunks.  For chunk k, in this order: `prep` (e.g. parse the input files) and the table builder on all host cores;
// upload + persistent launch on pool k mod CHD_N_POOLS, behind the launch of chunk k - CHD_N_POOLS; a finisher thread that waits for the launch, runs the
// stage-4 fallback, fetches the results and calls `fin` (e.g. write the output files).  While the device works on chunks k and k - 1, the host
// prepares chunk k + 1: the set-up (4.5 ms per 90-frame sequence and core) and the file I/O disappear behind the solve, and a chunk's last sequences
// share the device with the next chunk's first ones.
struct PipeChunk {
  int c0 = 0, c1 = 0;
  chd_batch* b = nullptr;
  int rc = 0;
  std::string err;
  std::thread fin;
  std::mutex mu; std::condition_variable cv; bool device_done = false;
  double t_built = 0, t_launched = 0, t_solved = 0, t_fetched = 0, t_finished = 0, kernel_ms0 = 0, kernel_ms1 = 0, t_step[3] = {0, 0, 0};      // ms since the start of the call (CHD_PIPE_TRACE)
};
// (`caller_index`: position j of `in` is the caller's sequence caller_index[j] -- the call works through the batch in its own order, messages name the caller's)
template <class Prep, class Fin>
static int solve_pipelined(chd_handle* h, int B, const chd_seq_in* in, chd_seq_out* out, Prep prep, Fin fin, const std::vector<int>* caller_index = nullptr) {
  HIP_TRY(h, hipSetDevice(h->device));
  const double t_begin = now_ms();
  h->call = chd_call_stats{};
  // chunk plan: a small first chunk (one sequence per compute unit: the device starts after ~30 ms of host work instead of after the set-up of a full chunk), then the rest in
  // CHD_N_POOLS - 1 equal chunks, so that every chunk has a lane (a hardware queue) of its own and all of them are in flight together: a lane is blocked until
  // the LAST straggler of its launch has ended (~1.5 s for 90-frame walks), so reusing lanes with chunks of a few hundred sequences costs more than it hides
  // (measured: ~8 chunks on four lanes 0.84 of the solve-only rate).  Chunks are capped at 4 096 sequences (device and page-locked buffers of a lane:
  // ~0.6 MB per 90-frame sequence each); beyond 256 + 3 x 4 096 sequences lanes are reused, with launches long enough to make the straggler's share small.
  int chunk = h->cfg.pipeline_chunk;
  if (chunk == 0) { chunk = (B - 256 + CHD_N_POOLS - 2) / (CHD_N_POOLS - 1); if (chunk < 256) chunk = 256; if (chunk > 4096) chunk = 4096; }
  if (chunk < 0 || chunk > B) chunk = B;                                                                // < 0: one chunk, i.e. upload, solve, fetch in turn
  std::vector<std::unique_ptr<PipeChunk>> ch;
  {
    int c0 = 0;
    const int first = (h->cfg.pipeline_chunk >= 0 && B > 2 * 256 && chunk > 256) ? 256 : chunk;          // (one sequence for every compute unit)
    while (c0 < B) {
      int n = ch.empty() ? first : chunk;
      if (B - c0 - n < std::min(64, (chunk + 1) / 2)) n = B - c0;            // (no crumbs at the end: less than half a chunk -- at most 64 sequences -- joins the chunk before)
      ch.emplace_back(new PipeChunk()); ch.back()->c0 = c0; ch.back()->c1 = std::min(B, c0 + n); c0 = ch.back()->c1;
    }
  }
  const int K = (int)ch.size();
  int n_pools = std::min(CHD_N_POOLS, K);
  const unsigned nt = host_threads(B);
  std::string first_err;
  int n_solved_chunks = 0;
  std::mutex agg_mu;
  double prep_ms = 0, build_wall_ms = 0, upload_ms = 0, wait_ms = 0;
  for (int k = 0; k < K; ++k) {
    PipeChunk& c = *ch[k];
    const int n = c.c1 - c.c0;
    double t0 = now_ms();
    prep(c.c0, c.c1);
    prep_ms += now_ms() - t0; t0 = now_ms();
    { TraceRange tr("chd set-up: tables of chunk " + std::to_string(k) + " (" + std::to_string(n) + " sequences)"); c.b = batch_build(h->cfg, n, in + c.c0, nt); }
    build_wall_ms += now_ms() - t0;
    c.t_built = now_ms() - t_begin;
    h->call.setup_cpu_ms += c.b->build_cpu_ms;
    for (int i = 0; i < n; ++i) if (!c.b->ok[i]) { std::lock_guard<std::mutex> lk(agg_mu); first_err = "sequence " + std::to_string(caller_index ? (*caller_index)[c.c0 + i] : c.c0 + i) + " rejected: " + c.b->build_err[i]; }
    const int pool = k % n_pools;
    t0 = now_ms();
    if (k >= n_pools) {            // the pool's previous launch (and its fallback) must be over
      PipeChunk& p = *ch[k - n_pools];
      std::unique_lock<std::mutex> lk(p.mu);
      p.cv.wait(lk, [&] { return p.device_done; });
    }
    if (c.b->order.empty()) {            // nothing solvable in this chunk: results are the rejection marks
      wait_ms += now_ms() - t0;
      c.rc = 1;
      { std::lock_guard<std::mutex> lk(c.mu); c.device_done = true; }
      c.cv.notify_all();
      c.b->fetched = true;               // (nothing on the device: the results are the rejection marks)
      batch_fetch(h, c.b, out + c.c0);
      fin(c.c0, c.c1);
      continue;
    }
    wait_ms += now_ms() - t0; t0 = now_ms();
    int rc = 0;
    const long long scale_ = (chunk + n - 1) / n;          // (the first chunk is the small one: the pools are sized for a full chunk of sequences like its own)
    if (k == 0)          // before the first launch: workspaces and reusable buffers of every pool this call will use, sized by this chunk (+ head-room)
      for (int p = 0; p < n_pools && rc == 0; ++p) {
        if (p == 0) rc = ensure_workspace(h, c.b->wd_need, c.b->wi_need, 2);
        if (rc == 0) rc = ensure_pool_bufs(h, p, std::max<long long>(c.b->tot_cd, 1) * scale_, std::max<long long>(c.b->tot_ci, 1) * scale_, c.b->od_stride * chunk, c.b->oi_stride * chunk, chunk);
        if (rc == 0) {
          const size_t need[6] = {(size_t)c.b->tot_cd * 8 * scale_, (size_t)c.b->tot_ci * 4 * scale_, (sizeof(SeqDesc) + sizeof(int)) * (size_t)chunk, (size_t)c.b->od_stride * chunk * 8, (size_t)c.b->oi_stride * chunk * 4, (size_t)chunk * 1024};
          for (int q = 0; q < 6; ++q) if (!pin_get(h, p, q, need[q] + need[q] / 8)) rc = fail(h, "page-locked staging buffers: allocation failed");
        }
        if (rc != 0 && p >= 1) { n_pools = p; rc = 0; h->err.clear(); break; }          // (long sequences: the memory holds fewer pools -- fewer launches in flight)
      }
    c.t_step[0] = now_ms() - t_begin;
    if (rc == 0 && !workspace_fits(h, c.b->wd_need, c.b->wi_need)) {          // this chunk holds a larger sequence than any before: every launch in flight has to end first
      for (int j = 0; j < k; ++j) { PipeChunk& pj = *ch[j]; std::unique_lock<std::mutex> lk(pj.mu); pj.cv.wait(lk, [&] { return pj.device_done; }); }
      rc = ensure_workspace(h, c.b->wd_need, c.b->wi_need, 2);
    }
    c.t_step[1] = now_ms() - t_begin;
    { TraceRange tr("chd upload + launch: chunk " + std::to_string(k) + " on lane " + std::to_string(pool));
    if (rc == 0) rc = batch_to_device(h, c.b, pool, true);
    c.t_step[2] = now_ms() - t_begin;
    if (rc == 0) rc = solve_launch_main(h, c.b); }
    upload_ms += now_ms() - t0;
    c.t_launched = now_ms() - t_begin;
    if (rc != 0) {
      c.rc = -1; c.err = h->err;
      { std::lock_guard<std::mutex> lk(c.mu); c.device_done = true; }
      c.cv.notify_all();
      continue;
    }
    c.fin = std::thread([h, &c, out, &fin, &agg_mu, &n_solved_chunks, t_begin]() {
      (void)hipSetDevice(h->device);
#ifdef CHD_STRESS_INJECT_RACE          // (tests/test_sanitizers.py: the harness must SEE a race when there is one -- an unsynchronised write to the shared handle from every finisher)
      h->call.finish_ms += 1.0;
#endif
      tl_err_sink = &c.err;               // (error text of this thread's calls goes to the chunk; the handle itself is shared: this thread owns its lane's result / scratch staging until device_done)
      int rc2;
      { TraceRange tr("chd finisher: wait for the launch, stage-4 fallback (" + std::to_string(c.c1 - c.c0) + " sequences)"); rc2 = solve_finish(h, c.b); }
      c.t_solved = now_ms() - t_begin;
      if (rc2 == 0) { TraceRange tr("chd finisher: fetch results"); rc2 = batch_fetch(h, c.b, out + c.c0); }
      c.t_fetched = now_ms() - t_begin;
      { std::lock_guard<std::mutex> lk(c.mu); c.device_done = true; }          // (the pool's buffers are free again: the results are on the host)
      c.cv.notify_all();
      tl_err_sink = nullptr;
      if (rc2 != 0) { c.rc = -1; return; }
      { TraceRange tr("chd finisher: output files"); fin(c.c0, c.c1); }
      c.t_finished = now_ms() - t_begin;
      std::lock_guard<std::mutex> lk(agg_mu);
      ++n_solved_chunks;
    });
  }
  for (auto& c : ch) if (c->fin.joinable()) c->fin.join();
  // ---- accounting over the chunks, then the device memory goes back
  chd_call_stats& cs = h->call;
  cs.n_chunks = K; cs.chunk = chunk; cs.host_threads = (int)nt; cs.n_sequences = B;
  cs.prep_ms = prep_ms; cs.setup_wall_ms = build_wall_ms; cs.upload_ms = upload_ms; cs.wait_for_pool_ms = wait_ms;
  std::string err;
  for (auto& c : ch) {
    if (c->rc < 0 && err.empty()) err = c->err;
    if (c->b) {
      const chd_batch_stats& s = c->b->stats;
      c->kernel_ms0 = s.kernel_ms[0]; c->kernel_ms1 = s.kernel_ms[1];
      cs.kernel_ms += s.kernel_ms[0] + s.kernel_ms[1]; cs.total_iters += s.total_iters; cs.total_factorizations += s.total_factorizations; cs.alg_bytes += s.alg_bytes;
      cs.n_fallback += s.n_fallback; cs.n_stalled += s.n_stalled; cs.n_rejected += c->b->B - (int)c->b->order.size();
      cs.sequence_ms += s.phase_ms[5]; if (s.max_seq_ms > cs.max_seq_ms) cs.max_seq_ms = s.max_seq_ms;
      chd_batch_free(h, c->b);
    }
  }
  cs.wall_ms = now_ms() - t_begin;
  if (std::getenv("CHD_PIPE_TRACE"))
    for (int k = 0; k < K; ++k)
      std::fprintf(stderr, "[chd pipeline] chunk %d (%d sequences, pool %d): built %.0f ms (pool ready %.0f, workspace %.0f, uploaded %.0f), launched %.0f, solved %.0f (kernel %.0f + %.0f ms), fetched %.0f, finished %.0f\n", k, ch[k]->c1 - ch[k]->c0, k % n_pools,
                   ch[k]->t_built, ch[k]->t_step[0], ch[k]->t_step[1], ch[k]->t_step[2], ch[k]->t_launched, ch[k]->t_solved, ch[k]->kernel_ms0, ch[k]->kernel_ms1, ch[k]->t_fetched, ch[k]->t_finished);
  if (!err.empty()) return fail(h, err);
  if (n_solved_chunks == 0) return fail(h, "no solvable sequence in the batch (" + first_err + ")");
  h->err = first_err;          // (a rejected sequence: the rest was solved)
  return 0;
}

extern "C" {

int chd_phys_version(void) { return CHD_PHYS_ABI_VERSION; }

void chd_config_default(chd_config* c) {
  c->w_com_lin = 0.4; c->w_com_ang = 1.7; c->w_ee = 0.3; c->w_smooth = 0.1; c->w_dur = 0.1;     // phys_optim.cpp:27-31
  const int mi[CHD_N_STAGES] = {7000, 7000, 7000, 2500, 2000, 7000};                              // :571, :640, :652, :706, :743
  for (int i = 0; i < CHD_N_STAGES; ++i) c->max_iter[i] = mi[i];
  c->tol = 1e-3;                                                                                   // :578
  c->threads_per_sequence = 0;
  c->stall_window = 0;
  c->max_workgroups = 0;
  c->lds_kilobytes = 0;
  c->factorisation = 0;
  c->pipeline_chunk = 0;
  c->damping_rule = 0;
  c->reserved[0] = 0;
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
  if (hipSetDevice(device_id) != hipSuccess) { delete h; return -4; }
  for (int p = 0; p < CHD_N_POOLS; ++p)
    if (hipStreamCreateWithFlags(&h->stream[p], hipStreamNonBlocking) != hipSuccess) { for (int q = 0; q < p; ++q) (void)hipStreamDestroy(h->stream[q]); delete h; return -4; }
  auto drop = [&]() { for (int p = 0; p < CHD_N_POOLS; ++p) (void)hipStreamDestroy(h->stream[p]); delete h; };
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, device_id) != hipSuccess) { drop(); return -5; }
  size_t lds = prop.maxSharedMemoryPerMultiProcessor;
  if (lds > 160 * 1024) lds = 160 * 1024;
  if (lds < 64 * 1024) lds = 64 * 1024;
  h->lds_bytes = (int)lds - 12288;    // the 12 KB hold the kernel's static LDS: sequence descriptor + solver context (2.4 KB), cumulative-time tables (5.6 KB)
  if (h->cfg.lds_kilobytes > 0 && h->cfg.lds_kilobytes * 1024 < h->lds_bytes) h->lds_bytes = std::max(32, h->cfg.lds_kilobytes) * 1024;
#ifndef CHD_HOST_EMU
  hipFuncSetAttribute((const void*)chd_solve_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, h->lds_bytes);
  hipFuncSetAttribute((const void*)chd_debug_eval_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, h->lds_bytes);
  hipFuncSetAttribute((const void*)chd_debug_linsolve_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, h->lds_bytes);
#endif
  // the factorisation / substitution phases are written for eight wavefronts (wave-specialised look-ahead, register prefetch
  // by lane group): other workgroup sizes are refused rather than silently mis-solved
  // (experiment, profiles/r02k_final/two_workgroups.md: with CHD_EXPERIMENTAL_256 set, 256-thread workgroups -- two per compute unit with
  //  76 KB of LDS each -- solve correctly through the generic substitution, at 0.8x the throughput)
  const bool exp256 = h->cfg.threads_per_sequence == 256 && std::getenv("CHD_EXPERIMENTAL_256") != nullptr;
  if (h->cfg.threads_per_sequence != 0 && h->cfg.threads_per_sequence != CHD_MAX_THREADS && !exp256) {
    std::fprintf(stderr, "chd_phys_create: threads_per_sequence must be 0 or %d\n", CHD_MAX_THREADS);
    drop(); return -7;
  }
  h->threads = exp256 ? 256 : CHD_MAX_THREADS;
  h->n_wg = h->cfg.max_workgroups > 0 ? h->cfg.max_workgroups : prop.multiProcessorCount;
  if (h->n_wg < 1) h->n_wg = 1;
  *out = h;
  return 0;
}

void chd_phys_destroy(chd_handle* h) {
  if (!h) return;
  (void)hipSetDevice(h->device);
  for (int p = 0; p < CHD_N_POOLS; ++p) {
    if (p == 0) { (void)hipFree(h->d_wd); (void)hipFree(h->d_wi); (void)hipFree(h->d_slots); }
    chd_handle::PoolBufs& P = h->pb[p];
    (void)hipFree(P.d_cd); (void)hipFree(P.d_ci); (void)hipFree(P.d_od); (void)hipFree(P.d_oi); (void)hipFree(P.d_descs); (void)hipFree(P.d_order); (void)hipFree(P.d_counter); (void)hipFree(P.d_f);
    for (int q = 0; q < 6; ++q) if (P.pin[q]) (void)hipHostFree(P.pin[q]);
    if (h->stream[p]) (void)hipStreamDestroy(h->stream[p]);
  }
  delete h;
}

const char* chd_phys_last_error(const chd_handle* h) { return h ? h->err.c_str() : "null handle"; }

void chd_batch_free(chd_handle* h, chd_batch* b) {
  if (!b) return;
  if (h) (void)hipSetDevice(h->device);
  if (b->owns_device) {
    (void)hipFree(b->d_cd); (void)hipFree(b->d_ci); (void)hipFree(b->d_od); (void)hipFree(b->d_oi);
    (void)hipFree(b->d_descs); (void)hipFree(b->d_order); (void)hipFree(b->d_counter); (void)hipFree(b->d_f);
  }
  (void)hipFree(b->d_x);
  for (int k = 0; k < 4; ++k) if (b->ev[k]) (void)hipEventDestroy(b->ev[k]);
  delete b;
}

int chd_batch_upload(chd_handle* h, int B, const chd_seq_in* in, chd_batch** out) {
  if (!h || !in || !out || B <= 0) return fail(h, "chd_batch_upload: bad arguments");
  *out = nullptr;
  HIP_TRY(h, hipSetDevice(h->device));
  chd_batch* b = batch_build(h->cfg, B, in, host_threads(B));
  if (b->order.empty()) { std::string m = "no solvable sequence in the batch (sequence 0: " + b->build_err[0] + ")"; chd_batch_free(h, b); return fail(h, m); }
  for (int i = 0; i < B; ++i) if (!b->ok[i]) h->err = "sequence " + std::to_string(i) + " rejected: " + b->build_err[i];
  if (batch_to_device(h, b, 0) != 0 || ensure_workspace(h, b->wd_need, b->wi_need) != 0) { std::string m = h->err; chd_batch_free(h, b); return fail(h, m); }
  hipError_t e = hipStreamSynchronize(h->stream[0]);
  if (e != hipSuccess) { chd_batch_free(h, b); return fail(h, std::string("sync: ") + hipGetErrorString(e)); }
  *out = b;
  return 0;
}

int chd_batch_solve(chd_handle* h, chd_batch* b) {
  if (!h || !b) return fail(h, "chd_batch_solve: bad arguments");
  HIP_TRY(h, hipSetDevice(h->device));
  TraceRange tr("chd_batch_solve (" + std::to_string(b->B) + " sequences)");
  if (ensure_workspace(h, b->wd_need, b->wi_need) != 0) return -1;
  if (solve_launch_main(h, b) != 0) return -1;
  return solve_finish(h, b);
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
  return batch_fetch(h, b, out);
}

// The order a whole call works through its sequences: longest first, and among equal lengths the ones with the most contact phases first -- over the WHOLE batch, before it is cut into
// chunks.  (Until round 5 only each chunk's own queue was ordered: the expensive sequences of the last chunk started last, and the call's tail was theirs.)  Both keys are in the
// inputs; no table has to be built for them.
static std::vector<int> call_order(int B, const chd_seq_in* in) {
  std::vector<int> perm(B);
  for (int i = 0; i < B; ++i) perm[i] = i;
  if (std::getenv("CHD_CALL_ORDER_OFF")) return perm;          // (A/B switch of the study in profiles/r05_experiments.md)
  auto phases = [&](int i) { return in[i].n_phases[0] + in[i].n_phases[1] + in[i].n_phases[2] + in[i].n_phases[3]; };
  std::stable_sort(perm.begin(), perm.end(), [&](int a, int b) { if (in[a].F != in[b].F) return in[a].F > in[b].F; return phases(a) > phases(b); });
  return perm;
}

int chd_phys_solve_batch(chd_handle* h, int B, const chd_seq_in* in, chd_seq_out* out) {
  if (!h || !in || !out || B <= 0) return fail(h, "chd_phys_solve_batch: bad arguments");
  const std::vector<int> perm = call_order(B, in);
  std::vector<chd_seq_in> in_p(B); std::vector<chd_seq_out> out_p(B);
  for (int j = 0; j < B; ++j) { in_p[j] = in[perm[j]]; out_p[j] = out[perm[j]]; }          // (the snapshot arrays are the caller's: results land there directly)
  const int rc = solve_pipelined(h, B, in_p.data(), out_p.data(), [](int, int) {}, [](int, int) {}, &perm);
  for (int j = 0; j < B; ++j) out[perm[j]] = out_p[j];
  return rc;
}

int chd_phys_get_call_stats(chd_handle* h, chd_call_stats* out) {
  if (!h || !out) return fail(h, "chd_phys_get_call_stats: bad arguments");
  *out = h->call;
  return 0;
}

int chd_phys_solve_dirs(chd_handle* h, int B, const char* const* in_dirs, const char* const* out_dirs, const int* nframes, int* status) {
  if (!h || B <= 0 || !in_dirs || !out_dirs || !nframes) return fail(h, "chd_phys_solve_dirs: bad arguments");
  // Unreadable directories drop out before the batch is formed (the reference's child process would have died on that video alone); the readable ones are
  // parsed, solved and written chunk by chunk: reading chunk k + 1 and writing chunk k - 1 run on the host while the device solves chunk k.
  std::vector<io::SeqFiles> files(B);
  std::vector<std::string> errs(B);
  std::vector<char> readable(B, 0);
  // (the token count of the four files is not known before they are parsed: every directory is read once, here, in parallel -- ~1 ms each -- and the
  //  pipeline below starts with the table builder.  Reading is `prep` for the accounting.)
  const double t_read0 = now_ms();
  { TraceRange tr("chd read input files (" + std::to_string(B) + " directories)");
  host_parallel_for(B, [&](int i) { readable[i] = io::read_inputs(in_dirs[i], nframes[i], files[i], errs[i]) ? 1 : 0; }); }
  const double read_ms = now_ms() - t_read0;
  std::vector<int> good;
  std::string first_err;
  for (int i = 0; i < B; ++i) {
    if (status) status[i] = readable[i] ? 0 : -1;
    if (readable[i]) good.push_back(i); else if (first_err.empty()) first_err = std::string(in_dirs[i]) + ": " + errs[i];
  }
  if (good.empty()) return fail(h, "chd_phys_solve_dirs: no readable input directory (" + first_err + ")");
  std::vector<chd_seq_in> in(good.size());
  std::vector<chd_seq_out> out(good.size());
  std::vector<io::SnapStore> store(good.size());
  for (size_t k = 0; k < good.size(); ++k) files[good[k]].fill(in[k]);
  {          // the call's order over the whole batch (call_order): `good` is the only index everything below goes through
    const std::vector<int> perm = call_order((int)good.size(), in.data());
    std::vector<int> g2(good.size());
    for (size_t j = 0; j < good.size(); ++j) g2[j] = good[perm[j]];
    good.swap(g2);
  }
  for (size_t k = 0; k < good.size(); ++k) {
    files[good[k]].fill(in[k]);
    store[k].bind(out[k], nframes[good[k]] + 4);
  }
  h->err.clear();
  std::vector<std::string> werr(good.size());
  std::vector<int> wst(good.size(), 0);
  std::atomic<long long> write_us{0};
  auto fin = [&](int c0, int c1) {          // the chunk's output files (called from the chunk's finisher thread)
    const double t0 = now_ms();
    host_parallel_for(c1 - c0, [&](int j) {
      const int k = c0 + j;
      if (out[k].stage_status[0] == -4) { wst[k] = -3; return; }      // rejected at set-up: no output files, as when the reference's child process dies
      if (!io::write_outputs(out_dirs[good[k]], files[good[k]].dt, out[k], werr[k])) wst[k] = -2;
    });
    write_us += (long long)((now_ms() - t0) * 1e3);
  };
  int rc = solve_pipelined(h, (int)good.size(), in.data(), out.data(), [](int, int) {}, fin, &good);
  h->call.prep_ms += read_ms; h->call.wall_ms += read_ms; h->call.finish_ms = write_us.load() * 1e-3;
  if (rc != 0) return rc;
  if (first_err.empty()) first_err = h->err;           // a sequence rejected at set-up (the rest was solved)
  for (size_t k = 0; k < good.size(); ++k) {
    if (wst[k] != 0 && status) status[good[k]] = wst[k];
    if (wst[k] == -2 && first_err.empty()) first_err = werr[k];
  }
  h->err = first_err;
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
  if (!h || !b || seq < 0 || seq >= b->B || stage < 0 || stage >= N_STAGES || !b->ok[seq]) return fail(h, "chd_debug_eval: bad arguments");
  HIP_TRY(h, hipSetDevice(h->device));
  if (ensure_workspace(h, b->wd_need, b->wi_need) != 0) return -1;
  const SeqModel& M = b->models[seq];
  const SeqDesc& dd = b->descs[seq];
  const StageDesc& S = M.d.st[stage];
  const int n = S.n, m = S.m;
  if (x) {
    if (b->x_cap < n) { (void)hipFree(b->d_x); b->d_x = nullptr; HIP_TRY(h, hipMalloc((void**)&b->d_x, (size_t)dd.max_n * 8)); b->x_cap = dd.max_n; }
    HIP_TRY(h, hipMemcpy(b->d_x, x, n * 8, hipMemcpyHostToDevice));
  }
  HIP_TRY(h, hipMemcpyAsync(b->d_order, &seq, sizeof(int), hipMemcpyHostToDevice, h->stream[0]));
  HIP_TRY(h, hipMemsetAsync(b->d_counter, 0, sizeof(int), h->stream[0]));
  (void)hipGetLastError();
  hipLaunchKernelGGL(chd_debug_eval_kernel, dim3(1), dim3(h->threads), h->lds_bytes, h->stream[0], b->d_descs, (const int*)b->d_order, b->d_counter, h->d_wd, h->d_wi,
                     stage, x ? (const double*)b->d_x : (const double*)nullptr, h->lds_bytes / 8, b->d_f);
  HIP_TRY(h, hipGetLastError());
  HIP_TRY(h, hipStreamSynchronize(h->stream[0]));
  double fo[2];
  HIP_TRY(h, hipMemcpy(fo, b->d_f, 16, hipMemcpyDeviceToHost));
  if (f) *f = fo[0];
  const double* wd = h->d_wd;          // workgroup 0's workspace
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

// The point behind output snapshot `snapshot` (0 after stage 1.2, 1 after 2.2, 2 after 3 or its stage-4 fallback) of sequence `seq` of a solved batch, in the
// NLP's own variables: the node variables (variable-set order, phys_optim.cpp:483-540 -- the first n_node_vars entries of every stage's x) and the phase
// durations of the four end-effectors (NLP order, all phases).  For tests: an evaluator that shares nothing with the kernel recomputes objective and
// constraint violation there.
int chd_debug_get_state(chd_handle* h, chd_batch* b, int seq, int snapshot, double* node_vars, int* n_node_vars, double* phase_durations, int* n_phases /*4*/) {
  if (!h || !b || seq < 0 || seq >= b->B || snapshot < 0 || snapshot >= CHD_N_SNAPSHOTS || !b->ok[seq] || !b->solved) return fail(h, "chd_debug_get_state: bad arguments");
  HIP_TRY(h, hipSetDevice(h->device));
  const SeqModel& M = b->models[seq];
  const SeqDesc& d = M.d;
  const long long ns = d.tot_entries + d.tot_phases;
  std::vector<double> st((size_t)ns);
  HIP_TRY(h, copy_d2h(h, b, st.data(), b->d_od + b->od_stride * seq + out_d_state_off(d.cap) + snapshot * ns, (size_t)ns * 8));
  if (n_node_vars) *n_node_vars = d.n_nodesvars;
  if (node_vars)
    for (int sp = 0; sp < N_SPLINES; ++sp)
      for (int k = 0; k < d.sp[sp].n_nodes * 6; ++k) {
        const int v = M.ci[d.o_varof + d.sp[sp].node_off + k];
        if (v >= 0) node_vars[d.sp[sp].var_off + v] = st[d.sp[sp].node_off + k];
      }
  for (int e = 0; e < N_EE; ++e) {
    if (n_phases) n_phases[e] = d.n_phase[e];
    if (phase_durations) for (int k = 0; k < d.n_phase[e]; ++k) phase_durations[d.phase_off[e] + k] = st[d.tot_entries + d.phase_off[e] + k];
  }
  return 0;
}

// Factor / solve self test of one sequence's KKT matrix (stage `stage` at the initial state, diagonal dw Dw / -dval):
// (`which` is ignored since round 4: one factorisation is left.)  info: [0] replaced pivots,
// [1] / [2] clock ticks (100 MHz) of `reps` factorisations / of the solve, [3] the factorisation that actually ran.
int chd_debug_linsolve(chd_handle* h, chd_batch* b, int seq, int stage, double dw, double dval, int which, int reps, const double* rhs, double* x, double* info) {
  if (!h || !b || seq < 0 || seq >= b->B || stage < 0 || stage >= N_STAGES || !b->ok[seq] || !rhs || !x) return fail(h, "chd_debug_linsolve: bad arguments");
  HIP_TRY(h, hipSetDevice(h->device));
  if (ensure_workspace(h, b->wd_need, b->wi_need) != 0) return -1;
  const StageDesc& S = b->models[seq].d.st[stage];
  const int N = S.n + S.m;
  double* d_buf = nullptr;
  HIP_TRY(h, hipMalloc((void**)&d_buf, (size_t)(2 * N + 16) * 8));
  hipError_t e = hipMemcpy(d_buf, rhs, (size_t)N * 8, hipMemcpyHostToDevice);          // (every later error path frees d_buf)
  if (e == hipSuccess) e = hipMemcpyAsync(b->d_order, &seq, sizeof(int), hipMemcpyHostToDevice, h->stream[0]);
  if (e == hipSuccess) e = hipMemsetAsync(b->d_counter, 0, sizeof(int), h->stream[0]);
  if (e == hipSuccess) {
    (void)hipGetLastError();
    hipLaunchKernelGGL(chd_debug_linsolve_kernel, dim3(1), dim3(h->threads), h->lds_bytes, h->stream[0], b->d_descs, (const int*)b->d_order, b->d_counter, h->d_wd, h->d_wi,
                       stage, h->lds_bytes / 8, dw, dval, which, reps < 1 ? 1 : reps, (const double*)d_buf, d_buf + N, d_buf + 2 * N);
    e = hipGetLastError();
  }
  if (e == hipSuccess) e = hipStreamSynchronize(h->stream[0]);
  if (e == hipSuccess) e = hipMemcpy(x, d_buf + N, (size_t)N * 8, hipMemcpyDeviceToHost);
  if (e == hipSuccess && info) e = hipMemcpy(info, d_buf + 2 * N, 14 * 8, hipMemcpyDeviceToHost);
  (void)hipFree(d_buf);
  if (e != hipSuccess) return fail(h, std::string("chd_debug_linsolve: ") + hipGetErrorString(e));
  return 0;
}

}  // extern "C"

// hip_stub.hpp — a stand-in for the HIP runtime, TEST INFRASTRUCTURE ONLY (SURVEY 5 "sanitizers").
//
// The host side of libchd_phys.so -- table builder threads, the four lanes of a pipelined call, finisher threads, page-locked staging, workspace slots
// claimed by compare-and-swap -- is multi-threaded, and nothing checked it for races or memory errors.  tests/host_emu/pipeline_stress.cpp compiles
// contact-human-dynamics_amd/csrc/chd_phys.hip as plain C++ with CHD_HOST_EMU against THIS header under -fsanitize=thread / -fsanitize=address: every HIP call
// the library makes is implemented here with the same ordering semantics the library relies on --
//   * a stream is a worker thread that executes its operations in order, asynchronously to the caller (copies, memsets, kernel launches, event records);
//   * an event completes when the stream reaches it; hipEventSynchronize / hipStreamSynchronize / hipDeviceSynchronize block the caller;
//   * "device" and page-locked memory are heap blocks (so ASan sees overruns, use after free and double frees; TSan sees a host thread writing a staging
//     buffer that an asynchronous copy is still reading);
//   * a kernel launch runs <kernel>_emu(grid, args...) on the stream's thread: chd_phys.hip's emulation of its kernel starts `grid` threads, one per
//     resident workgroup, which claim workspace slots and drain the launch's queue with the same atomics as the device code.
// Nothing under contact-human-dynamics_amd/ includes this file unless the test build defines CHD_HOST_EMU_HIP_STUB.
#pragma once
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <functional>
#include <mutex>
#include <set>
#include <thread>
#include <vector>

typedef int hipError_t;
enum { hipSuccess = 0, hipErrorInvalidValue = 1, hipErrorOutOfMemory = 2, hipErrorNoDevice = 100 };
enum hipMemcpyKind { hipMemcpyHostToHost = 0, hipMemcpyHostToDevice = 1, hipMemcpyDeviceToHost = 2, hipMemcpyDeviceToDevice = 3, hipMemcpyDefault = 4 };
enum { hipStreamNonBlocking = 1, hipHostMallocDefault = 0 };
enum hipFuncAttribute { hipFuncAttributeMaxDynamicSharedMemorySize = 8 };
struct dim3 { unsigned x, y, z; dim3(unsigned a = 1, unsigned b = 1, unsigned c = 1) : x(a), y(b), z(c) {} };
struct hipDeviceProp_t { int multiProcessorCount; size_t maxSharedMemoryPerMultiProcessor; char name[64]; };

struct StubStream {
  std::mutex mu; std::condition_variable cv;
  std::deque<std::function<void()>> q;
  bool stop = false; int busy = 0;
  std::thread worker;
  StubStream() : worker([this] { run(); }) {}
  ~StubStream() { { std::lock_guard<std::mutex> lk(mu); stop = true; } cv.notify_all(); worker.join(); }
  void run() {
    for (;;) {
      std::function<void()> f;
      { std::unique_lock<std::mutex> lk(mu); cv.wait(lk, [&] { return stop || !q.empty(); }); if (q.empty()) return; f = std::move(q.front()); q.pop_front(); busy = 1; }
      f();
      { std::lock_guard<std::mutex> lk(mu); busy = 0; }
      cv.notify_all();
    }
  }
  void enqueue(std::function<void()> f) { { std::lock_guard<std::mutex> lk(mu); q.push_back(std::move(f)); } cv.notify_all(); }
  void sync() { std::unique_lock<std::mutex> lk(mu); cv.wait(lk, [&] { return q.empty() && !busy; }); }
};
typedef StubStream* hipStream_t;

struct StubEvent { std::mutex mu; std::condition_variable cv; long long pending = 0, done = 0; std::chrono::steady_clock::time_point t; };
typedef StubEvent* hipEvent_t;

namespace hipstub {
inline std::mutex& reg_mu() { static std::mutex m; return m; }
inline std::set<StubStream*>& streams() { static std::set<StubStream*> s; return s; }
inline int n_cus() { const char* e = std::getenv("CHD_STUB_CUS"); int v = e ? std::atoi(e) : 6; return v > 0 ? v : 6; }
}  // namespace hipstub

inline const char* hipGetErrorString(hipError_t e) { return e == hipSuccess ? "no error" : e == hipErrorOutOfMemory ? "out of memory (stub)" : e == hipErrorNoDevice ? "no device (stub)" : "invalid value (stub)"; }
inline hipError_t hipGetLastError() { return hipSuccess; }
inline hipError_t hipGetDeviceCount(int* n) { if (std::getenv("CHD_STUB_NO_DEVICE")) { *n = 0; return hipErrorNoDevice; } *n = 1; return hipSuccess; }
inline hipError_t hipSetDevice(int) { return hipSuccess; }
inline hipError_t hipGetDeviceProperties(hipDeviceProp_t* p, int) { p->multiProcessorCount = hipstub::n_cus(); p->maxSharedMemoryPerMultiProcessor = 160 * 1024; std::snprintf(p->name, sizeof(p->name), "host emulation"); return hipSuccess; }
template <class F> inline hipError_t hipFuncSetAttribute(F, int, int) { return hipSuccess; }

inline hipError_t hipMalloc(void** p, size_t n) { *p = std::malloc(n ? n : 1); return *p ? hipSuccess : hipErrorOutOfMemory; }
inline hipError_t hipFree(void* p) { std::free(p); return hipSuccess; }
inline hipError_t hipHostMalloc(void** p, size_t n, unsigned) { *p = std::malloc(n ? n : 1); return *p ? hipSuccess : hipErrorOutOfMemory; }
inline hipError_t hipHostFree(void* p) { std::free(p); return hipSuccess; }

inline hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned) { *s = new StubStream(); std::lock_guard<std::mutex> lk(hipstub::reg_mu()); hipstub::streams().insert(*s); return hipSuccess; }
inline hipError_t hipStreamDestroy(hipStream_t s) { if (!s) return hipSuccess; s->sync(); { std::lock_guard<std::mutex> lk(hipstub::reg_mu()); hipstub::streams().erase(s); } delete s; return hipSuccess; }
inline hipError_t hipStreamSynchronize(hipStream_t s) { s->sync(); return hipSuccess; }
inline hipError_t hipDeviceSynchronize() {
  std::vector<StubStream*> v;
  { std::lock_guard<std::mutex> lk(hipstub::reg_mu()); v.assign(hipstub::streams().begin(), hipstub::streams().end()); }
  for (StubStream* s : v) s->sync();
  return hipSuccess;
}

inline hipError_t hipMemcpy(void* d, const void* s, size_t n, hipMemcpyKind) { std::memcpy(d, s, n); return hipSuccess; }
inline hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, hipMemcpyKind, hipStream_t st) { st->enqueue([=] { std::memcpy(d, s, n); }); return hipSuccess; }
inline void hipstub_copy2d(void* d, size_t dp, const void* s, size_t sp, size_t w, size_t h) { for (size_t r = 0; r < h; ++r) std::memcpy((char*)d + r * dp, (const char*)s + r * sp, w); }
inline hipError_t hipMemcpy2D(void* d, size_t dp, const void* s, size_t sp, size_t w, size_t h, hipMemcpyKind) { hipstub_copy2d(d, dp, s, sp, w, h); return hipSuccess; }
inline hipError_t hipMemcpy2DAsync(void* d, size_t dp, const void* s, size_t sp, size_t w, size_t h, hipMemcpyKind, hipStream_t st) { st->enqueue([=] { hipstub_copy2d(d, dp, s, sp, w, h); }); return hipSuccess; }
inline hipError_t hipMemsetAsync(void* d, int v, size_t n, hipStream_t st) { st->enqueue([=] { std::memset(d, v, n); }); return hipSuccess; }

inline hipError_t hipEventCreate(hipEvent_t* e) { *e = new StubEvent(); return hipSuccess; }
inline hipError_t hipEventDestroy(hipEvent_t e) { delete e; return hipSuccess; }
inline hipError_t hipEventRecord(hipEvent_t e, hipStream_t st) {
  long long id;
  { std::lock_guard<std::mutex> lk(e->mu); id = ++e->pending; }
  st->enqueue([e, id] { { std::lock_guard<std::mutex> lk(e->mu); e->t = std::chrono::steady_clock::now(); e->done = id; } e->cv.notify_all(); });
  return hipSuccess;
}
inline hipError_t hipEventSynchronize(hipEvent_t e) { std::unique_lock<std::mutex> lk(e->mu); e->cv.wait(lk, [&] { return e->done == e->pending; }); return hipSuccess; }
inline hipError_t hipEventElapsedTime(float* ms, hipEvent_t a, hipEvent_t b) {
  std::scoped_lock lk(a->mu, b->mu);
  *ms = std::chrono::duration<float, std::milli>(b->t - a->t).count();
  return hipSuccess;
}

// a kernel launch: the arguments are evaluated NOW (as the real runtime copies them at launch time), the emulation runs when the stream gets there
template <class F, class... A>
inline void hipstub_launch(hipStream_t st, F f, unsigned grid, A... a) { st->enqueue([=] { f(grid, a...); }); }
#define hipLaunchKernelGGL(kern, grid, block, lds, stream, ...) hipstub_launch((stream), kern##_emu, (grid).x, __VA_ARGS__)

// chd_prepare.hip -- libchd_prepare.so: the C ABI of include/chd_prepare.h (producer of the physics stage's inputs).
//
//   chd_prep_frames     one HIP kernel launch over the frames of all clips of a run, one thread per frame (chd_prepare_kernels.hpp: two forward-kinematics passes,
//                       centre of mass, hip offsets, inertia, toe / heel trajectories).  A frame reads (7 J) doubles and writes 28: HBM bound, 1.8 KB per frame at
//                       31 joints; the chain products run out of the thread's private arrays.  No CPU path: without a HIP device the call fails.
//   chd_bvh_load_batch  native BVH reader on the host's cores (chd_bvh.hpp).
#include <hip/hip_runtime.h>

#include <atomic>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <sstream>
#include <string>
#include <thread>
#include <vector>

#include "chd_bvh.hpp"
#include "chd_prepare_kernels.hpp"

static thread_local std::string g_err;
static thread_local double g_kernel_ms = 0.0;

__global__ __launch_bounds__(64) void chd_prep_kernel(const chd_prep_skeleton* __restrict__ skel, long long n, const double* __restrict__ rot, const double* __restrict__ pos,
                                                      double* __restrict__ out) {
  __shared__ chd_prep_skeleton S;                // the tables are read ~100 times per frame: one LDS copy per workgroup
  for (int i = threadIdx.x; i < (int)(sizeof(chd_prep_skeleton) / 4); i += blockDim.x) ((int*)&S)[i] = ((const int*)skel)[i];
  __syncthreads();
  const long long f = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (f >= n) return;
  double R[9 * CHD_PREP_MAX_JOINTS], P[3 * CHD_PREP_MAX_JOINTS], o[CHD_PREP_OUT_STRIDE];
  const int J = S.n_joints;
  chd_prep::prep_frame(S, rot + f * J * 4, pos + f * J * 3, o, R, P);
  for (int k = 0; k < CHD_PREP_OUT_STRIDE; ++k) out[f * CHD_PREP_OUT_STRIDE + k] = o[k];
}

static_assert(sizeof(chd_prep_skeleton) % 4 == 0, "copied word by word");

extern "C" {

int chd_prep_version(void) { return CHD_PREP_ABI_VERSION; }
const char* chd_prep_last_error(void) { return g_err.c_str(); }
double chd_prep_last_kernel_ms(void) { return g_kernel_ms; }

#define PREP_TRY(call)                                                                                     \
  do { hipError_t e_ = (call); if (e_ != hipSuccess) { g_err = std::string(#call) + ": " + hipGetErrorString(e_); rc = -1; goto done; } } while (0)

int chd_prep_frames(const chd_prep_skeleton* skel, int device, long long n, const double* rot, const double* pos, double* out) {
  g_err.clear(); g_kernel_ms = 0.0;
  if (!skel || n < 0 || (n > 0 && (!rot || !pos || !out))) { g_err = "chd_prep_frames: bad arguments"; return -1; }
  const int J = skel->n_joints;
  if (J < 1 || J > CHD_PREP_MAX_JOINTS || skel->n_joints_body < 1 || skel->n_joints_body > J || skel->n_segments < 1 || skel->n_segments > CHD_PREP_MAX_SEGMENTS ||
      skel->seg_first[skel->n_segments] > CHD_PREP_MAX_SEGMENT_JOINTS) { g_err = "chd_prep_frames: skeleton tables out of range"; return -1; }
  for (int j = 0; j < J; ++j) if (skel->parents[j] >= j) { g_err = "chd_prep_frames: joints must follow their parents"; return -1; }
  for (int s = 0; s < skel->n_segments; ++s) {
    if (skel->seg_first[s + 1] <= skel->seg_first[s]) { g_err = "chd_prep_frames: empty segment"; return -1; }
    for (int k = skel->seg_first[s]; k < skel->seg_first[s + 1]; ++k) if (skel->seg_joint[k] < 0 || skel->seg_joint[k] >= skel->n_joints_body) { g_err = "chd_prep_frames: segment joint out of range"; return -1; }
  }
  for (int k = 0; k < 2; ++k)
    if (skel->hip_inds[k] < 0 || skel->hip_inds[k] >= skel->n_joints_body || skel->toe_inds[k] < 0 || skel->toe_inds[k] >= J || skel->heel_inds[k] < 0 || skel->heel_inds[k] >= J) { g_err = "chd_prep_frames: hip / toe / heel index out of range"; return -1; }
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) { g_err = "chd_prep_frames: no HIP device available (this library has no CPU path)"; return -2; }
  if (device < 0 || device >= ndev) { g_err = "chd_prep_frames: bad device id"; return -3; }
  if (n == 0) return 0;
  int rc = 0;
  chd_prep_skeleton* d_s = nullptr; double *d_rot = nullptr, *d_pos = nullptr, *d_out = nullptr;
  hipStream_t st = nullptr; hipEvent_t e0 = nullptr, e1 = nullptr;
  float ms = 0.0f;
  PREP_TRY(hipSetDevice(device));
  PREP_TRY(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
  PREP_TRY(hipEventCreate(&e0)); PREP_TRY(hipEventCreate(&e1));
  PREP_TRY(hipMalloc((void**)&d_s, sizeof(chd_prep_skeleton)));
  PREP_TRY(hipMalloc((void**)&d_rot, (size_t)n * J * 4 * 8));
  PREP_TRY(hipMalloc((void**)&d_pos, (size_t)n * J * 3 * 8));
  PREP_TRY(hipMalloc((void**)&d_out, (size_t)n * CHD_PREP_OUT_STRIDE * 8));
  PREP_TRY(hipMemcpyAsync(d_s, skel, sizeof(chd_prep_skeleton), hipMemcpyHostToDevice, st));
  PREP_TRY(hipMemcpyAsync(d_rot, rot, (size_t)n * J * 4 * 8, hipMemcpyHostToDevice, st));
  PREP_TRY(hipMemcpyAsync(d_pos, pos, (size_t)n * J * 3 * 8, hipMemcpyHostToDevice, st));
  PREP_TRY(hipEventRecord(e0, st));
  (void)hipGetLastError();
  hipLaunchKernelGGL(chd_prep_kernel, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, st, (const chd_prep_skeleton*)d_s, n, (const double*)d_rot, (const double*)d_pos, d_out);
  PREP_TRY(hipGetLastError());
  PREP_TRY(hipEventRecord(e1, st));
  PREP_TRY(hipMemcpyAsync(out, d_out, (size_t)n * CHD_PREP_OUT_STRIDE * 8, hipMemcpyDeviceToHost, st));
  PREP_TRY(hipStreamSynchronize(st));
  PREP_TRY(hipEventElapsedTime(&ms, e0, e1));
  g_kernel_ms = ms;
done:
  (void)hipFree(d_s); (void)hipFree(d_rot); (void)hipFree(d_pos); (void)hipFree(d_out);
  if (e0) (void)hipEventDestroy(e0);
  if (e1) (void)hipEventDestroy(e1);
  if (st) (void)hipStreamDestroy(st);
  return rc;
}

static char* dup_str(const std::string& s) { char* p = (char*)malloc(s.size() + 1); if (p) memcpy(p, s.c_str(), s.size() + 1); return p; }

int chd_bvh_load_batch(int n, const char* const* paths, int n_threads, chd_bvh_clip* out) {
  if (n < 0 || (n > 0 && (!paths || !out))) return -1;
  for (int i = 0; i < n; ++i) memset(&out[i], 0, sizeof(chd_bvh_clip));
  unsigned nt = n_threads > 0 ? (unsigned)n_threads : std::thread::hardware_concurrency();
  if (nt == 0) nt = 4;
  if (nt > 64) nt = 64;
  if ((int)nt > n) nt = (unsigned)(n > 0 ? n : 1);
  std::atomic<int> next{0}, failed{0};
  auto body = [&]() {
    for (;;) {
      const int i = next.fetch_add(1);
      if (i >= n) return;
      std::string err;
      chd_bvh::Clip c;
      try {
        std::ifstream f(paths[i], std::ios::binary);
        if (!f) err = std::string(paths[i]) + ": cannot open";
        else {
          std::stringstream ss; ss << f.rdbuf();
          err = chd_bvh::parse(paths[i], ss.str(), c);
        }
      } catch (const std::exception& e) { err = std::string(paths[i]) + ": " + e.what(); }
      chd_bvh_clip& o = out[i];
      if (!err.empty()) { o.error = dup_str(err); failed.fetch_add(1); continue; }
      o.n_frames = c.n_frames; o.n_joints = c.n_joints; o.channels = c.channels; o.frame_time = c.frame_time;
      memset(o.order, 0, 4); memcpy(o.order, c.order.c_str(), 3);
      std::string names;
      for (size_t k = 0; k < c.names.size(); ++k) { if (k) names += '\n'; names += c.names[k]; }
      o.names = dup_str(names);
      o.parents = (int*)malloc(sizeof(int) * c.parents.size()); memcpy(o.parents, c.parents.data(), sizeof(int) * c.parents.size());
      o.offsets = (double*)malloc(8 * c.offsets.size()); memcpy(o.offsets, c.offsets.data(), 8 * c.offsets.size());
      o.positions = (double*)malloc(8 * (c.positions.size() + 1)); memcpy(o.positions, c.positions.data(), 8 * c.positions.size());
      o.rotations = (double*)malloc(8 * (c.rotations.size() + 1)); memcpy(o.rotations, c.rotations.data(), 8 * c.rotations.size());
    }
  };
  std::vector<std::thread> pool;
  for (unsigned t = 1; t < nt; ++t) pool.emplace_back(body);
  body();
  for (auto& th : pool) th.join();
  return failed.load();
}

void chd_bvh_free(int n, chd_bvh_clip* clips) {
  if (!clips) return;
  for (int i = 0; i < n; ++i) {
    free(clips[i].names); free(clips[i].parents); free(clips[i].offsets); free(clips[i].positions); free(clips[i].rotations); free(clips[i].error);
    memset(&clips[i], 0, sizeof(chd_bvh_clip));
  }
}

}  // extern "C"

// chd_prepare.hip -- libchd_prepare.so: the C ABI of include/chd_prepare.h (producer of the physics stage's inputs).
//
//   chd_prep_frames     one HIP kernel launch over the frames of all clips of a run, one thread per frame (chd_prepare_kernels.hpp: two forward-kinematics passes,
//                       centre of mass, hip offsets, inertia, toe / heel trajectories).  A frame reads (7 J) doubles and writes 28: HBM bound, 1.8 KB per frame at
//                       31 joints; the chain products run out of the thread's private arrays.  No CPU path: without a HIP device the call fails.
//   chd_bvh_load_batch  native BVH reader on the host's cores (chd_bvh.hpp).
//   chd_openpose_load_dirs / chd_totalcap_load_batch  the JSON inputs in front of the kinematic optimisation and the contact network (chd_json.hpp), all videos of a run on the host's cores.
#include <hip/hip_runtime.h>

#include <atomic>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <sstream>
#include <string>
#include <thread>
#include <vector>

#include <dirent.h>

#include <algorithm>

#include "chd_bvh.hpp"
#include "chd_json.hpp"
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
  if (skel->parents[0] != -1) { g_err = "chd_prep_frames: parents[0] must be -1 (the root)"; return -1; }
  for (int j = 1; j < J; ++j) if (skel->parents[j] >= j || skel->parents[j] < 0) { g_err = "chd_prep_frames: joints must follow their parents (0 <= parents[j] < j)"; return -1; }
  if (skel->seg_first[0] != 0) { g_err = "chd_prep_frames: seg_first[0] must be 0"; return -1; }
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
      o.parents = (int*)malloc(sizeof(int) * (c.parents.size() + 1));
      o.offsets = (double*)malloc(8 * (c.offsets.size() + 1));
      o.positions = (double*)malloc(8 * (c.positions.size() + 1));
      o.rotations = (double*)malloc(8 * (c.rotations.size() + 1));
      if (!o.names || !o.parents || !o.offsets || !o.positions || !o.rotations) {      // out of memory: this clip fails on its own, with a message
        free(o.names); free(o.parents); free(o.offsets); free(o.positions); free(o.rotations);
        memset(&o, 0, sizeof(chd_bvh_clip));
        o.error = dup_str(std::string(paths[i]) + ": out of memory");
        failed.fetch_add(1);
        continue;
      }
      memcpy(o.parents, c.parents.data(), sizeof(int) * c.parents.size());
      memcpy(o.offsets, c.offsets.data(), 8 * c.offsets.size());
      memcpy(o.positions, c.positions.data(), 8 * c.positions.size());
      memcpy(o.rotations, c.rotations.data(), 8 * c.rotations.size());
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

// ---- JSON ingest (host) ------------------------------------------------------------------------------------------------------------------------------------------
namespace {
bool read_file(const std::string& path, std::string& text) {
  FILE* f = fopen(path.c_str(), "rb");
  if (!f) return false;
  char buf[1 << 16];
  size_t n;
  text.clear();
  while ((n = fread(buf, 1, sizeof(buf), f)) > 0) text.append(buf, n);
  const bool ok = !ferror(f);
  fclose(f);
  return ok;
}
double* dup_doubles(const std::vector<double>& v) { double* p = (double*)malloc(8 * (v.size() + 1)); if (p && !v.empty()) memcpy(p, v.data(), 8 * v.size()); return p; }

// contact_net.load_keypoint_dir (openpose_utils.py:48-76)
std::string load_openpose_dir(const std::string& dir, int num_joints, int& n_frames, std::vector<double>& data) {
  DIR* d = opendir(dir.c_str());
  if (!d) return dir + ": cannot open the directory";
  std::vector<std::string> files;
  while (dirent* en = readdir(d)) {
    const std::string name = en->d_name;
    if (name == "." || name == "..") continue;
    const size_t dot = name.rfind('.');
    if ((dot == std::string::npos ? name : name.substr(dot + 1)) == "json") files.push_back(name);      // f.split('.')[-1] == 'json'
  }
  closedir(d);
  std::sort(files.begin(), files.end());
  if (files.empty()) return dir + ": no .json result files";
  for (const std::string& name : files) {
    const std::string path = dir + "/" + name;
    std::string text;
    if (!read_file(path, text)) return path + ": cannot open";
    chd_json::Value root;
    const std::string err = chd_json::parse(text, root);
    if (!err.empty()) return path + ": " + err;
    const chd_json::Value* people = root.get("people");
    if (!people || people->kind != chd_json::Value::Array) return path + ": no \"people\" array";
    if (people->size() == 0) { data.insert(data.end(), (size_t)num_joints * 3, 0.0); continue; }
    if (people->all_numbers) return path + ": people[0] is not an object";
    if (!chd_json::numbers(people->items[0].get("pose_keypoints_2d"), data, (size_t)num_joints * 3))
      return path + ": people[0].pose_keypoints_2d is not an array of " + std::to_string(num_joints * 3) + " numbers";
  }
  n_frames = (int)files.size();
  return "";
}

// totalcap_io.load_totalcap_results (totalcap_utils.py:33-79)
std::string load_totalcap(const std::string& path, chd_totalcap_clip& o) {
  std::string text;
  if (!read_file(path, text)) return path + ": cannot open";
  chd_json::Value root;
  const std::string err = chd_json::parse(text, root);
  if (!err.empty()) return path + ": " + err;
  const chd_json::Value* frames = root.get("totalcapResults");
  if (!frames || frames->kind != chd_json::Value::Array) return path + ": no \"totalcapResults\" array";
  std::vector<double> trans, j3, s3, sr, bc, fc;
  int nj = -1, ns = -1, nb = -1, nf = -1;
  if (frames->all_numbers) return path + ": \"totalcapResults\" is not an array of frames";
  for (size_t k = 0; k < frames->items.size(); ++k) {
    const chd_json::Value& fr = frames->items[k];
    const std::string at = path + ": frame " + std::to_string(k) + ": ";
    if (!chd_json::xyz(fr.get("trans"), trans)) return at + "\"trans\" is not {x, y, z}";
    const chd_json::Value *joints = fr.get("joints"), *smpl = fr.get("SMPLJoints");
    if (!joints || joints->kind != chd_json::Value::Array || joints->all_numbers || !smpl || smpl->kind != chd_json::Value::Array || smpl->all_numbers) return at + "\"joints\" / \"SMPLJoints\" missing";
    if (nj < 0) { nj = (int)joints->items.size(); ns = (int)smpl->items.size(); }
    if ((int)joints->items.size() != nj || (int)smpl->items.size() != ns) return at + "another number of joints than frame 0";
    for (const chd_json::Value& j : joints->items) if (!chd_json::xyz(j.get("pos"), j3)) return at + "a joint without \"pos\": {x, y, z}";
    for (const chd_json::Value& j : smpl->items) if (!chd_json::xyz(j.get("pos"), s3) || !chd_json::xyz(j.get("rot"), sr)) return at + "an SMPL joint without \"pos\" / \"rot\": {x, y, z}";
    const size_t b0 = bc.size(), f0 = fc.size();
    if (!chd_json::numbers(fr.get("bodyCoeffs"), bc) || !chd_json::numbers(fr.get("faceCoeffs"), fc)) return at + "\"bodyCoeffs\" / \"faceCoeffs\" are not arrays of numbers";
    if (nb < 0) { nb = (int)(bc.size() - b0); nf = (int)(fc.size() - f0); }
    if ((int)(bc.size() - b0) != nb || (int)(fc.size() - f0) != nf) return at + "another number of coefficients than frame 0";
  }
  o.n_frames = (int)frames->items.size(); o.n_joints = nj < 0 ? 0 : nj; o.n_smpl_joints = ns < 0 ? 0 : ns; o.n_body_coeffs = nb < 0 ? 0 : nb; o.n_face_coeffs = nf < 0 ? 0 : nf;
  o.root_trans = dup_doubles(trans); o.joint3d = dup_doubles(j3); o.smpl_joint3d = dup_doubles(s3); o.smpl_joint_angles = dup_doubles(sr);
  o.body_coeffs = dup_doubles(bc); o.face_coeffs = dup_doubles(fc);
  return "";
}

template <class Body> int run_on_threads(int n, int n_threads, Body body) {
  unsigned nt = n_threads > 0 ? (unsigned)n_threads : std::thread::hardware_concurrency();
  if (nt == 0) nt = 4;
  if (nt > 64) nt = 64;
  if ((int)nt > n) nt = (unsigned)(n > 0 ? n : 1);
  std::atomic<int> next{0}, failed{0};
  auto loop = [&]() { for (;;) { const int i = next.fetch_add(1); if (i >= n) return; if (!body(i)) failed.fetch_add(1); } };
  std::vector<std::thread> pool;
  for (unsigned t = 1; t < nt; ++t) pool.emplace_back(loop);
  loop();
  for (auto& th : pool) th.join();
  return failed.load();
}
}  // namespace

extern "C" {

int chd_openpose_load_dirs(int n, const char* const* dirs, int num_joints, int n_threads, chd_keypoint_clip* out) {
  if (n < 0 || num_joints < 1 || (n > 0 && (!dirs || !out))) return -1;
  for (int i = 0; i < n; ++i) memset(&out[i], 0, sizeof(chd_keypoint_clip));
  return run_on_threads(n, n_threads, [&](int i) {
    std::string err; std::vector<double> data; int nf = 0;
    try { err = load_openpose_dir(dirs[i], num_joints, nf, data); } catch (const std::exception& e) { err = std::string(dirs[i]) + ": " + e.what(); }
    if (!err.empty()) { out[i].error = dup_str(err); return false; }
    out[i].n_frames = nf; out[i].data = dup_doubles(data);
    return true;
  });
}
void chd_openpose_free(int n, chd_keypoint_clip* clips) {
  if (!clips) return;
  for (int i = 0; i < n; ++i) { free(clips[i].data); free(clips[i].error); memset(&clips[i], 0, sizeof(chd_keypoint_clip)); }
}

int chd_totalcap_load_batch(int n, const char* const* paths, int n_threads, chd_totalcap_clip* out) {
  if (n < 0 || (n > 0 && (!paths || !out))) return -1;
  for (int i = 0; i < n; ++i) memset(&out[i], 0, sizeof(chd_totalcap_clip));
  return run_on_threads(n, n_threads, [&](int i) {
    std::string err;
    try { err = load_totalcap(paths[i], out[i]); } catch (const std::exception& e) { err = std::string(paths[i]) + ": " + e.what(); }
    if (!err.empty()) { out[i].error = dup_str(err); return false; }
    return true;
  });
}
void chd_totalcap_free(int n, chd_totalcap_clip* clips) {
  if (!clips) return;
  for (int i = 0; i < n; ++i) {
    free(clips[i].root_trans); free(clips[i].joint3d); free(clips[i].smpl_joint3d); free(clips[i].smpl_joint_angles); free(clips[i].body_coeffs); free(clips[i].face_coeffs); free(clips[i].error);
    memset(&clips[i], 0, sizeof(chd_totalcap_clip));
  }
}

}  // extern "C"

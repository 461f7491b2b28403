// chd_bvh.hpp -- native BVH reader (host), the semantics of BVH.load (src/skeleton_fitting/ik/BVH.py:25-168) for the files this pipeline handles.
//
// Written from the file format: a token stream over HIERARCHY (ROOT / JOINT / End Site / { } / OFFSET / CHANNELS) and MOTION (Frames:, Frame Time:, numbers).
//   * End Sites are not joints (their OFFSET is skipped);
//   * the rotation order comes from the FIRST CHANNELS line (its last three entries when it has six);
//   * every joint is taken to have as many channels as the LAST joint declares: 3 = only the root carries a translation, 6 = every joint does,
//     9 = translation, rotation, scale per non-root joint (BVH.py:156-160);
//   * translations default to the joint offsets; Euler angles are degrees, composed in local order q0 (q1 q2)
//     (Quaternions.from_euler(world=False), Quaternions.py:401-414, with from_angle_axis' `axis / (|axis| + 1e-10)`).
// The Python mirror is skeleton_io.load_bvh; tests/test_prepare_native.py holds the two together (names, parents, offsets and translations exactly, rotations to rounding:
// NumPy's vectorised sin / cos and libm's differ in the last bit).
#pragma once
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include <string>
#include <vector>

namespace chd_bvh {

struct Clip {
  int n_frames = 0, n_joints = 0, channels = 0;
  double frame_time = 0.0;
  std::string order;
  std::vector<std::string> names;
  std::vector<int> parents;
  std::vector<double> offsets, positions, rotations;
};

inline void quat_mul(const double* a, const double* b, double* o) {      // Hamilton product, the sums in Quaternions.__mul__'s order
  const double w = a[0] * b[0] - ((a[1] * b[1] + a[2] * b[2]) + a[3] * b[3]);
  const double x = (a[0] * b[1] + b[0] * a[1]) + (a[2] * b[3] - a[3] * b[2]);
  const double y = (a[0] * b[2] + b[0] * a[2]) + (a[3] * b[1] - a[1] * b[3]);
  const double z = (a[0] * b[3] + b[0] * a[3]) + (a[1] * b[2] - a[2] * b[1]);
  o[0] = w; o[1] = x; o[2] = y; o[3] = z;
}

inline bool is_space(char c) { return c == ' ' || c == '\t' || c == '\n' || c == '\r' || c == '\f' || c == '\v'; }

// whitespace-separated tokens of [b, e) as (begin, length) pairs
inline void tokenize(const char* b, const char* e, std::vector<std::pair<const char*, int>>& out) {
  const char* p = b;
  while (p < e) {
    while (p < e && is_space(*p)) ++p;
    if (p >= e) break;
    const char* s = p;
    while (p < e && !is_space(*p)) ++p;
    out.emplace_back(s, (int)(p - s));
  }
}

inline bool tok_is(const std::pair<const char*, int>& t, const char* s) { return (int)strlen(s) == t.second && memcmp(t.first, s, t.second) == 0; }
inline double tok_double(const std::pair<const char*, int>& t, bool& ok) {
  char buf[64];
  if (t.second <= 0 || t.second >= 63) { ok = false; return 0.0; }
  memcpy(buf, t.first, t.second); buf[t.second] = 0;
  char* end = nullptr;
  const double v = strtod(buf, &end);
  if (end != buf + t.second) ok = false;
  return v;
}

// returns "" on success, else the reason
inline std::string parse(const std::string& path, const std::string& text, Clip& c) {
  const size_t cut = text.find("MOTION");
  if (cut == std::string::npos) return path + ": no MOTION section";
  std::vector<std::pair<const char*, int>> tok;
  tokenize(text.data(), text.data() + cut, tok);
  std::vector<int> stack;
  int active = -1, pending = -2;            // pending: -2 none, -3 an End Site, >= 0 a joint whose '{' has not been seen
  bool in_end_site = false, ok = true;
  size_t i = 0;
  while (i < tok.size()) {
    if (tok_is(tok[i], "ROOT") || tok_is(tok[i], "JOINT")) {
      if (i + 1 >= tok.size()) return path + ": joint without a name";
      c.names.emplace_back(tok[i + 1].first, tok[i + 1].second);
      c.offsets.insert(c.offsets.end(), {0.0, 0.0, 0.0});
      c.parents.push_back(active);
      pending = (int)c.names.size() - 1;
      i += 2;
    } else if (tok_is(tok[i], "End")) { pending = -3; i += 2; }
    else if (tok_is(tok[i], "{")) {
      stack.push_back(active);
      if (pending == -3) in_end_site = true; else if (pending >= 0) active = pending;
      pending = -2; ++i;
    } else if (tok_is(tok[i], "}")) {
      if (stack.empty()) return path + ": unbalanced braces";
      const int prev = stack.back(); stack.pop_back();
      if (in_end_site) in_end_site = false; else active = prev;
      ++i;
    } else if (tok_is(tok[i], "OFFSET")) {
      if (i + 3 >= tok.size()) return path + ": truncated OFFSET";
      if (!in_end_site) {
        if (active < 0) return path + ": OFFSET outside a joint";
        for (int k = 0; k < 3; ++k) c.offsets[3 * active + k] = tok_double(tok[i + 1 + k], ok);
      }
      i += 4;
    } else if (tok_is(tok[i], "CHANNELS")) {
      if (i + 1 >= tok.size()) return path + ": truncated CHANNELS";
      const int n = atoi(std::string(tok[i + 1].first, tok[i + 1].second).c_str());
      if (n < 0 || i + 2 + n > tok.size()) return path + ": truncated CHANNELS";
      c.channels = n;
      if (c.order.empty() && (n == 3 || n >= 6)) {
        std::string o;
        const size_t first = i + 2 + (n == 3 ? 0 : 3);
        for (size_t k = first; k < first + 3; ++k) {
          if (tok_is(tok[k], "Xrotation")) o += 'x'; else if (tok_is(tok[k], "Yrotation")) o += 'y'; else if (tok_is(tok[k], "Zrotation")) o += 'z';
        }
        if (o.size() == 3) c.order = o;
      }
      i += 2 + n;
    } else ++i;
  }
  if (!ok) return path + ": malformed number in the hierarchy";
  if (c.names.empty() || c.order.empty()) return path + ": no joints / rotation channels found";
  const int J = c.n_joints = (int)c.names.size();
  std::vector<std::pair<const char*, int>> mt;
  tokenize(text.data() + cut, text.data() + text.size(), mt);
  size_t kf = 0, kt = 0;
  for (size_t k = 0; k < mt.size(); ++k) { if (!kf && tok_is(mt[k], "Frames:")) kf = k + 1; if (!kt && tok_is(mt[k], "Time:")) kt = k + 1; }
  if (!kf || !kt || kf >= mt.size() || kt >= mt.size()) return path + ": no Frames: / Frame Time: line";
  const int nf = c.n_frames = atoi(std::string(mt[kf].first, mt[kf].second).c_str());
  c.frame_time = tok_double(mt[kt], ok);
  const int per = c.channels == 3 ? 3 + 3 * J : c.channels == 6 ? 6 * J : c.channels == 9 ? 3 + 9 * (J - 1) : -1;
  if (per < 0) return path + ": " + std::to_string(c.channels) + " channels per joint are not supported";
  const size_t first = kt + 1;
  if (nf < 0 || mt.size() < first + (size_t)nf * per)
    return path + ": expected " + std::to_string((long long)nf * per) + " motion values (" + std::to_string(nf) + " frames x " + std::to_string(per) + "), found " + std::to_string((long long)mt.size() - (long long)first);
  c.positions.resize((size_t)nf * J * 3); c.rotations.resize((size_t)nf * J * 4);
  int ax[3];
  for (int k = 0; k < 3; ++k) ax[k] = c.order[k] == 'x' ? 0 : c.order[k] == 'y' ? 1 : 2;
  const double deg = M_PI / 180.0, kk = 1.0 / (1.0 + 1e-10);          // np.radians; from_angle_axis' axis normalisation (a unit axis divided by 1 + 1e-10)
  std::vector<double> row((size_t)per);
  for (int f = 0; f < nf; ++f) {
    for (int k = 0; k < per; ++k) row[k] = tok_double(mt[first + (size_t)f * per + k], ok);
    for (int j = 0; j < J; ++j) {
      double* p = &c.positions[((size_t)f * J + j) * 3];
      double e[3] = {0.0, 0.0, 0.0};
      for (int k = 0; k < 3; ++k) p[k] = c.offsets[3 * j + k];
      if (c.channels == 3) { if (j == 0) for (int k = 0; k < 3; ++k) p[k] = row[k]; for (int k = 0; k < 3; ++k) e[k] = row[3 + 3 * j + k]; }
      else if (c.channels == 6) { for (int k = 0; k < 3; ++k) { p[k] = row[6 * j + k]; e[k] = row[6 * j + 3 + k]; } }
      else {
        if (j == 0) { for (int k = 0; k < 3; ++k) p[k] = row[k]; }
        else { const double* d = &row[3 + 9 * (j - 1)]; for (int k = 0; k < 3; ++k) { e[k] = d[3 + k]; p[k] += d[k] * d[6 + k]; } }
      }
      double q[3][4];
      for (int k = 0; k < 3; ++k) {
        const double h = 0.5 * (e[k] * deg), s = sin(h) * kk;
        q[k][0] = cos(h); q[k][1] = q[k][2] = q[k][3] = 0.0 * s;      // (the zero components are sin x 0 as in the array form: -0.0 stays -0.0)
        q[k][1 + ax[k]] = s;
      }
      double t[4];
      quat_mul(q[1], q[2], t);
      quat_mul(q[0], t, &c.rotations[((size_t)f * J + j) * 4]);
    }
  }
  if (!ok) return path + ": malformed number in the motion section";
  return "";
}

}  // namespace chd_bvh

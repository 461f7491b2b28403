// chd_ik_host.hpp -- host side shared by the HIP library (chd_ik.hip) and the CPU emulation used in tests
// (tests/host_emu/ik_emu.cpp): packs a batch of chd_ik_seq into flat pools + per-sequence descriptors.
#pragma once
#include <cstring>
#include <string>
#include <vector>

#include "../../include/chd_ik.h"
#include "chd_ik_kernels.hpp"

namespace chd_ik {

struct IkBatch {
  std::vector<IkSeq> seqs;
  std::vector<int> ipool;
  std::vector<double> dpool;
  std::vector<double> state;            // initial state (both device buffers start from it)
  std::vector<int> frame_seq, frame_idx;   // one entry per (sequence, frame) workgroup
  int max_J = 0, max_T = 0;             // the workgroup scratch is carved for these
  std::string err;

  bool build(int B, const chd_ik_seq* in) {
    for (int b = 0; b < B; ++b) {
      const chd_ik_seq& q = in[b];
      if (q.n_frames < 1 || q.n_joints < 1 || q.n_joints > MAXJ || q.n_targets < 1 || q.n_targets > MAXT) { err = "sequence " + std::to_string(b) + ": sizes out of range"; return false; }
      if (!q.parents || !q.target_joints || !q.targets || !q.rot_in || !q.pos_in || !q.rot_out || !q.pos_out) { err = "sequence " + std::to_string(b) + ": null pointer"; return false; }
      IkSeq s;
      s.F = q.n_frames; s.J = q.n_joints; s.T = q.n_targets;
      s.o_parents = (int)ipool.size();
      for (int j = 0; j < s.J; ++j) {
        if (q.parents[j] >= j || (j == 0 && q.parents[j] != -1) || (j > 0 && q.parents[j] < 0)) { err = "sequence " + std::to_string(b) + ": parents must satisfy parents[0] = -1, 0 <= parents[j] < j"; return false; }
        ipool.push_back(q.parents[j]);
      }
      s.o_tj = (int)ipool.size();
      for (int t = 0; t < s.T; ++t) {
        if (q.target_joints[t] < 0 || q.target_joints[t] >= s.J) { err = "sequence " + std::to_string(b) + ": target joint out of range"; return false; }
        ipool.push_back(q.target_joints[t]);
      }
      // descendants_mask (AnimationStructure.py:129-150, 217) restricted to the targeted joints, as bit masks both ways
      s.o_masks = (int)ipool.size();                     // == o_parents + J + T: the kernel stages the three tables with one copy
      std::vector<unsigned> jt_strict(s.J, 0u), jt_self(s.J, 0u);
      std::vector<unsigned long long> ta_strict(s.T, 0ull), ta_self(s.T, 0ull);
      for (int t = 0; t < s.T; ++t) {
        int a = q.target_joints[t];
        jt_self[a] |= 1u << t; ta_self[t] |= 1ull << a;
        for (a = q.parents[a]; a >= 0; a = q.parents[a]) { jt_strict[a] |= 1u << t; jt_self[a] |= 1u << t; ta_strict[t] |= 1ull << a; ta_self[t] |= 1ull << a; }
      }
      for (int j = 0; j < s.J; ++j) ipool.push_back((int)jt_strict[j]);
      for (int j = 0; j < s.J; ++j) ipool.push_back((int)jt_self[j]);
      for (int t = 0; t < s.T; ++t) { ipool.push_back((int)(unsigned)(ta_strict[t] & 0xffffffffull)); ipool.push_back((int)(unsigned)(ta_strict[t] >> 32)); }
      for (int t = 0; t < s.T; ++t) { ipool.push_back((int)(unsigned)(ta_self[t] & 0xffffffffull)); ipool.push_back((int)(unsigned)(ta_self[t] >> 32)); }
      s.o_targets = (long long)dpool.size();
      dpool.insert(dpool.end(), q.targets, q.targets + (size_t)s.T * s.F * 3);
      s.o_state = (long long)state.size();
      for (int f = 0; f < s.F; ++f) {
        state.insert(state.end(), q.rot_in + (size_t)f * s.J * 4, q.rot_in + (size_t)(f + 1) * s.J * 4);
        state.insert(state.end(), q.pos_in + (size_t)f * s.J * 3, q.pos_in + (size_t)(f + 1) * s.J * 3);
      }
      if (s.J > max_J) max_J = s.J;
      if (s.T > max_T) max_T = s.T;
      for (int f = 0; f < s.F; ++f) { frame_seq.push_back(b); frame_idx.push_back(f); }
      seqs.push_back(s);
    }
    return true;
  }
  void scatter(const double* final_state, const chd_ik_seq* out) const {
    for (size_t b = 0; b < seqs.size(); ++b) {
      const IkSeq& s = seqs[b];
      for (int f = 0; f < s.F; ++f) {
        const double* x = final_state + s.o_state + (long long)f * 7 * s.J;
        std::memcpy(out[b].rot_out + (size_t)f * s.J * 4, x, sizeof(double) * 4 * s.J);
        std::memcpy(out[b].pos_out + (size_t)f * s.J * 3, x + 4 * s.J, sizeof(double) * 3 * s.J);
      }
    }
  }
};

inline IkParams params_of(const chd_ik_config* cfg) {
  IkParams P;
  P.iterations = cfg->iterations; P.translate = cfg->translate; P.damping = cfg->damping; P.smoothness = cfg->smoothness; P.gamma = cfg->gamma;
  return P;
}

}  // namespace chd_ik

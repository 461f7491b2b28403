// chd_kinopt_host.hpp -- host side shared by the HIP library (chd_kinopt.hip) and the CPU emulation used in tests
// (tests/host_emu/kin_emu.cpp): packs a batch of chd_kin_seq into flat pools + per-video descriptors, and groups the videos by the
// number of workgroups that solve one (chd_kinopt_kernels.hpp: a cluster of G workgroups per clip, each owning a run of frames).
#pragma once
#include <algorithm>
#include <string>
#include <vector>

#include "../../include/chd_kinopt.h"
#include "chd_kinopt_kernels.hpp"

namespace chd_kin {

enum { KIN_LDS_DOUBLES_DEFAULT = 19760 };       // 154.4 KB of the compute unit's 160 (3 KB more are static: the small structs): slices of 13 frames

inline void config_default(chd_kin_config* cfg) {
  static const int parents[NJ] = {-1, 0, 1, 2, 3, 3, 3, 0, 7, 8, 9, 9, 9, 0, 13, 14, 15, 16, 16, 16, 16, 16, 15, 22, 23, 15, 25, 26};      // combined_body_25.bvh
  cfg->max_nfev = 50; cfg->ftol = 1e-8; cfg->xtol = 1e-8; cfg->gtol = 1e-12;
  cfg->lsmr_atol = 1e-6; cfg->lsmr_btol = 1e-6; cfg->lsmr_conlim = 1e8; cfg->lsmr_maxiter = 0;
  for (int j = 0; j < NJ; ++j) cfg->parents[j] = parents[j];
  for (int k = 0; k < 4; ++k) cfg->reserved[k] = 0;
}

struct KinGroup { int G; std::vector<int> clips; };      // the clips solved by clusters of G workgroups, longest first

struct KinBatch {
  KinParams P;
  std::vector<KinSeq> seqs;
  std::vector<int> cluster;                     // workgroups per clip
  std::vector<KinGroup> groups;
  std::vector<double> dpool, state;
  std::vector<int> ipool;
  long long work_total = 0;
  int lds_doubles = KIN_LDS_DOUBLES_DEFAULT, frames_cap = 0;
  std::string err;

  bool build(const chd_kin_config* cfg, int B, const chd_kin_seq* in) {
    for (int j = 0; j < NJ; ++j) {
      const int p = cfg->parents[j];
      if ((j == 0 && p != -1) || (j > 0 && (p < 0 || p >= j))) { err = "parents must satisfy parents[0] = -1, 0 <= parents[j] < j"; return false; }
      P.parents[j] = p; P.desc[j] = 0u;
    }
    for (int t = 1; t < NJ; ++t) for (int a = P.parents[t]; a >= 0; a = P.parents[a]) P.desc[a] |= 1u << t;
    for (int j = 0; j < NJ; ++j) { P.fwd[j] = FWD[j]; P.bwd[j] = BWD[j]; P.smooth_w[j] = SMOOTH_W[j]; }
    bool dfs = true;                            // depth-first order: the descendants of j are j + 1 .. j + (their number)
    int nd[NJ];
    for (int j = 0; j < NJ; ++j) {
      nd[j] = 0;
      for (int t = j + 1; t < NJ; ++t) nd[j] += (P.desc[j] >> t) & 1u;
      for (int t = j + 1; t <= j + nd[j]; ++t) if (!((P.desc[j] >> t) & 1u)) dfs = false;
      int na = 0;
      for (int a = P.parents[j]; a >= 0; a = P.parents[a]) { if (na < 8) P.anc[j][na] = (unsigned char)a; ++na; }
      P.anc_n[j] = na;
      for (int q = na; q < 8; ++q) P.anc[j][q] = (unsigned char)j;
    }
    // the walks over the descendants: the shortest piece length T with which the four idle lanes of a frame can take every piece beyond a joint's first
    for (int j = 0; j < 32; ++j) { P.walk_of[j] = j < NJ ? j : 0; P.walk_t0[j] = j < NJ ? j + 1 : 1; P.walk_t1[j] = j < NJ ? (dfs ? j + nd[j] : NJ - 1) : 0; P.walk_help[j] = 0; }
    if (dfs) {
      int T = 1;
      for (;; ++T) { int extra = 0; for (int j = 0; j < NJ; ++j) extra += nd[j] > T ? (nd[j] + T - 1) / T - 1 : 0; if (extra <= 4) break; }
      int h = 0;
      for (int j = 0; j < NJ; ++j)
        if (nd[j] > T) {
          P.walk_t1[j] = j + T;
          for (int t0 = j + 1 + T; t0 <= j + nd[j]; t0 += T, ++h) {
            P.walk_of[NJ + h] = j; P.walk_t0[NJ + h] = t0; P.walk_t1[NJ + h] = t0 + T - 1 < j + nd[j] ? t0 + T - 1 : j + nd[j];
            P.walk_help[j] |= 1 << h;
          }
        }
    }
    if (cfg->max_nfev < 1) { err = "max_nfev must be positive"; return false; }
    P.max_nfev = cfg->max_nfev; P.ftol = cfg->ftol; P.xtol = cfg->xtol; P.gtol = cfg->gtol;
    P.atol = cfg->lsmr_atol; P.btol = cfg->lsmr_btol; P.conlim = cfg->lsmr_conlim; P.lsmr_maxiter = cfg->lsmr_maxiter;
    lds_doubles = cfg->reserved[1] > 0 ? cfg->reserved[1] : KIN_LDS_DOUBLES_DEFAULT;
    if (lds_doubles < HALO_V + KC_HALO) { err = "reserved[1] (LDS doubles per workgroup) must hold the received halos: >= 598"; return false; }
    // frames per workgroup: what the LDS block holds, or reserved[2] (a smaller value forces more workgroups per clip; a larger one slices that stay in device memory)
    frames_cap = cfg->reserved[2] > 0 ? cfg->reserved[2] : lds_frames(lds_doubles);
    if (frames_cap < 2) frames_cap = 2;
    for (int b = 0; b < B; ++b) {
      const chd_kin_seq& q = in[b];
      const std::string who = "video " + std::to_string(b) + ": ";
      if (q.n_frames < 3 || q.n_frames > 100000) { err = who + "needs at least 3 frames"; return false; }
      if (!q.offsets || !q.pose3d || !q.root_trans || !q.pose2d_n || !q.proj_w || !q.data_w || !q.contact || !q.x) { err = who + "null pointer"; return false; }
      const size_t F = (size_t)q.n_frames;
      KinSeq s;
      s.F = q.n_frames; s.n = NV * s.F; s.m = rows_of(s.F);
      s.o_const = (long long)dpool.size();
      dpool.insert(dpool.end(), q.offsets, q.offsets + 84);
      dpool.insert(dpool.end(), q.pose3d, q.pose3d + 84 * F);
      dpool.insert(dpool.end(), q.root_trans, q.root_trans + 3 * F);
      dpool.insert(dpool.end(), q.pose2d_n, q.pose2d_n + 56 * F);
      dpool.insert(dpool.end(), q.proj_w, q.proj_w + 28 * F);
      dpool.insert(dpool.end(), q.data_w, q.data_w + 28 * F);
      s.o_contact = (long long)ipool.size();
      ipool.insert(ipool.end(), q.contact, q.contact + 28 * F);
      const int G = cluster_size(s.F, frames_cap);
      cluster.push_back(G);
      s.o_work = work_total; work_total += work_doubles(s.F, G);
      s.o_x = (long long)state.size();
      state.insert(state.end(), q.x, q.x + NV * F);
      for (int k = 0; k < 3; ++k) { s.floor_n[k] = q.floor_n[k]; s.floor_p[k] = q.floor_p[k]; }
      s.w[0] = q.w_proj; s.w[1] = q.w_smooth_vel; s.w[2] = q.w_smooth_acc; s.w[3] = q.w_data; s.w[4] = q.w_vel; s.w[5] = q.w_floor;
      seqs.push_back(s);
    }
    for (int b = 0; b < B; ++b) {
      auto it = std::find_if(groups.begin(), groups.end(), [&](const KinGroup& g) { return g.G == cluster[b]; });
      if (it == groups.end()) { groups.push_back(KinGroup{cluster[b], {}}); it = groups.end() - 1; }
      it->clips.push_back(b);
    }
    std::sort(groups.begin(), groups.end(), [](const KinGroup& a, const KinGroup& b) { return a.G > b.G; });
    for (auto& g : groups) std::stable_sort(g.clips.begin(), g.clips.end(), [&](int a, int b) { return seqs[a].F > seqs[b].F; });
    return true;
  }
  // stats: KIN_STATS doubles per video (kin_solve)
  void scatter(const double* final_state, const double* stats, chd_kin_seq* out) const {
    for (size_t b = 0; b < seqs.size(); ++b) {
      const KinSeq& s = seqs[b];
      for (long long i = 0; i < s.n; ++i) out[b].x[i] = final_state[s.o_x + i];
      const double* st = stats + (size_t)KIN_STATS * b;
      out[b].cost = st[0]; out[b].nfev = (int)st[1]; out[b].njev = (int)st[2]; out[b].status = (int)st[3]; out[b].lsmr_iterations = (int)st[4]; out[b].optimality = st[5]; out[b].jv_fraction = st[6]; out[b].jtu_fraction = st[7];
    }
  }
};

}  // namespace chd_kin

// chd_device.hpp — plain-old-data descriptors shared by the host-side table builder
// (chd_model.hpp) and the solver kernel (chd_kernels.hpp).
//
// One sequence = one NLP (phys_optim.cpp:375-762).  Everything a workgroup needs is reachable
// from a SeqDesc: four base pointers (constant doubles / ints, workspace doubles / ints) plus
// 32-bit offsets.  All arithmetic data is fp64.
#pragma once
#include <stdint.h>

namespace chd {

// Pointers into HBM carry the global address space in device code: a pointer *loaded from memory* is otherwise
// generic and every access through it becomes a flat_load / flat_store (which also ties up the LDS counter).
#if defined(__HIP_DEVICE_COMPILE__)
typedef __attribute__((address_space(1))) double GD;
typedef __attribute__((address_space(1))) int GI;
typedef __attribute__((address_space(1))) unsigned long long GU;
#else
typedef double GD;
typedef int GI;
typedef unsigned long long GU;
#endif

enum { N_SPLINES = 10, N_EE = 4, N_STAGES = 6 };
// spline ids: 0 base-lin, 1 base-ang, 2..5 ee-motion (NLP ee order), 6..9 ee-force

// constraint families (bit flags) — names follow Parameters::ConstraintName (parameters.h)
enum {
  FAM_BASEACC = 1, FAM_TERRAIN = 2, FAM_ROM = 4, FAM_HEELDIST = 8,
  FAM_DYNAMIC = 16, FAM_FORCE = 32, FAM_HEIGHT = 64, FAM_TOTALTIME = 128
};

// row tasks: one task = the rows one thread evaluates together
enum {
  T_BASEACC = 0,   // a = spline (0/1), b = junction j       -> 3 rows
  T_TERRAIN = 1,   // a = ee, b = node                        -> 1 row
  T_ROM = 2,       // a = ee, b = sample index in t_rom       -> 1 row
  T_HEELDIST = 3,  // a = pair (0/1), b = sample index        -> 1 row
  T_DYN = 4,       // b = sample index in t_dyn               -> 6 rows
  T_FORCE = 5,     // a = ee, b = node                        -> 5 rows
  T_HEIGHT = 6,    // a = ee, t = sample time                 -> 1 row
  T_TOTALTIME = 7, // a = ee                                  -> 1 row
  T_DURBOUND = 8   // a = ee, b = duration index              -> 1 row
};

struct SplineDesc {
  int n_nodes, n_polys, n_var, var_off;
  int node_off;     // first entry of this spline in the node-entry arrays (6 entries per node: [deriv][dim])
  int poly_off;     // first polynomial of this spline in the polynomial arrays
  int phase_based;  // 0 for the two base splines
  int ee;           // end-effector of a phase-based spline, -1 otherwise
};

struct StageDesc {
  int stage, families, opt_dur, max_iter;
  int n, m, n_dur, dur_off[N_EE];
  int Nb, bc, w;            // KKT layout: banded part, border, half-bandwidth
  int n_tasks;
  int dyn_first, n_dyn;     // the dynamics samples are a contiguous run of tasks (a lane group works on each)
  int nnz_jac;
  int valid;
  int heel_row0;            // first row of the heel-distance family (rows heel_row0 + pair * n_trom + sample), -1 if the stage has none
  double w_data[3], w_vel[3], w_acc[3], w_dur;
  // offsets into ci
  int o_pos_var, o_pos_row, o_task;     // task: 4 ints (type, a, b, row0)
  int o_env;                            // 2 ints per band position: first / last coupled band position (envelope of the KKT matrix)
  int o_rcnt;                           // per band position k: number of leading border positions whose first coupled band position is <= k (the border is sorted by it)
  // offsets into cd
  int o_cl, o_cu, o_Dw, o_task_t;
};

// cached spline sample used by the cost terms (one per cost spline per data frame)
// (SC_OME / SC_OMC: d2 p / d(node coefficient j) d(duration) for a duration of an earlier phase / of the current phase -- the derivative of the Hermite weight
//  itself; SC_CF: d f / d p_i of all cost terms, scaled.  The first SC_LDS fields are what the entry tasks read: they are staged in LDS)
enum { SC_WP = 0, SC_WV = 4, SC_P = 8, SC_V = 11, SC_DXDT = 14, SC_POLY = 17, SC_PHASE = 18, SC_LAST = 19,
       SC_OME = 20, SC_OMC = 24, SC_LDS = 28, SC_QEE = 28, SC_QEC = 31, SC_QCC = 34, SC_CF = 37, SC_STRIDE = 40 };
// second-order duration tables (stage 3): per end-effector one slot per row sample; 4 doubles = (cur phase, S_ee, S_ec, S_cc)
enum { D2_STRIDE = 4, X2_STRIDE = 6 };
// cache of the ee-motion splines at the range-of-motion sample times (exact curvature of the heel-distance rows): per (end-effector,
// sample) the four position weights at SC_WP, the row's scaled multiplier at RC_MU and the polynomial at SC_POLY (same offsets as the
// cost-sample cache, so the same weight look-up serves both)
enum { RC_MU = 4, RC_STRIDE = 18 };
// Exact node x duration block of the Lagrangian Hessian (duration stage): one record per (row sample, spline whose node values the entry differentiates by,
// end-effector whose durations it differentiates by).  The entry for coefficient j (dimension dim) of the spline's polynomial XR_POLY and a duration T_k is
//   w_j A_class[dim] + om_class,j B[dim],  class = earlier phase (k < XR_CUR) / current phase (k == XR_CUR, unless XR_LAST),
// with w the position weights of the sample (XR_W), om the derivatives of those weights by the duration (XR_OME / XR_OMC).
enum { XR_POLY = 0, XR_CUR = 1, XR_LAST = 2, XR_W = 3, XR_AE = 7, XR_AC = 10, XR_B = 13, XR_OME = 16, XR_OMC = 20, XR_STRIDE = 24 };
// record blocks per duration end-effector e, in this order: HEIGHT (n_tdyn, ee-motion spline of e), ROM_M (n_trom, the same spline), HEEL_OWN (n_trom, the same),
// DYN_P (n_tdyn, the same), HEEL_X (n_trom, the ee-motion spline of the other contact point of e's foot), DYN_F (n_tdyn, ee-force spline of e),
// ROM_C (n_trom, base-lin), DYN_C (n_tdyn, base-lin), ROM_A (n_trom, base-ang)
enum { XB_HEIGHT = 0, XB_ROM_M, XB_HEEL_OWN, XB_DYN_P, XB_HEEL_X, XB_DYN_F, XB_ROM_C, XB_DYN_C, XB_ROM_A, XB_COUNT };

struct SeqDesc {
  const GD* cd;       // constant doubles
  const GI* ci;       // constant ints
  GD* wd;             // workspace doubles (state + solver vectors + KKT storage)
  GI* wi;             // workspace ints
  GD* out_d;          // results: snapshots + per-stage statistics
  GI* out_i;

  int F, cap;         // data frames, snapshot capacity
  double dt, T, mass, leg_len, heel_len, heel_dist;
  double ratio_low;                    // chd_config.damping_rule = 1: an accepted step that delivers less than this fraction of the predicted merit reduction raises the damping (0 = rule off; solve_stage)
  double normal[3], point[3], gdir[3], hx, hy, bn[3], bt1[3], bt2[3];
  SplineDesc sp[N_SPLINES];
  int n_phase[N_EE], phase_off[N_EE], start_contact[N_EE];
  int n_nodesvars, tot_entries, tot_polys, tot_phases, max_polys;
  int n_tdyn, n_trom;
  // cd offsets
  int o_data[6];      // com, euler, ee-motion targets (NLP ee order); F x 3 each
  int o_hip[2], o_inertia, o_tcost, o_tdyn, o_trom, o_phase_dur0, o_node0, o_phase_dur_in;
  // ci offsets
  int o_varof;        // tot_entries: local optimisation index or -1
  int o_pinfo;        // tot_polys x 4: phase, k_in_phase, n_in_phase, is_const
  int o_varnode;      // n_nodesvars: first node entry (spline-local entry index) of each node variable
  int o_varspl;       // n_nodesvars: spline of each node variable
  // wd offsets — state
  int o_node, o_poly_dur, o_pend, o_phase_dur, o_phend, o_ttot;
  // wd offsets — solver vectors (n-, m- and N-sized), see chd_kernels.hpp
  int o_vec_n, o_vec_m, o_vec_N, o_scache, o_d2tab, o_x2tab, d2_slots, o_rcache, o_xtab;
  int max_n, max_m, max_N;
  // wd offsets — KKT storage
  int o_K0b, o_K0x, o_Kfb, o_Kfx;      // full band, border rows (unfactored); lower band, border rows (factor)
  int o_pmb, o_pmx, o_pmt;             // occupancy masks of the unfactored matrix (64-bit words, one bit per stored entry; chd_kernels.hpp "KKT storage"):
                                       // band rows, border rows, and the border transposed (per band column: which border rows hold an entry in it)
  long long sz_K0b, sz_K0x, sz_Kfb, sz_Kfx;
  // wi offsets
  int o_flags, o_first, o_sign;
  int o_envw;           // working copy of the stage's envelope (2 ints per KKT position): widened when an entry lands outside it
  int o_rcntw;          // working copy of StageDesc::o_rcnt
  int o_side_v, o_side_pq, side_cap;   // the Gauss-Newton values of the KKT entries that the exact blocks of an evaluation changed (chd_kernels.hpp model_switch): side_cap doubles in wd, 2 side_cap ints in wi
  int o_csr_rp, o_csr_col, o_csr_row, csr_cap;      // the marked entries of the unfactored matrix as a row-sorted list (chd_kernels.hpp "KKT storage"): N + 1 row starts, csr_cap (column, row) pairs
  StageDesc st[N_STAGES];
};

// per-stage statistics written to out_d (8 doubles per stage)
enum { RS_STATUS = 0, RS_ITERS = 1, RS_KKT = 2, RS_VIOL = 3, RS_OBJ = 4, RS_MU = 5, RS_NFACT = 6, RS_AUX = 7 /* 1 = ended by the stall guard */, RS_STRIDE = 8 };
// out_d layout: [N_STAGES x RS_STRIDE] then 3 snapshots x 10 blocks (base_lin, base_ang_deg, 4 ee_pos, 4 ee_force) x cap x 3, then
// 24 phase timers, then three state slots (node values, phase durations) as the stages behind the three snapshots left them (slot 2 after stage 3 is what
// the stage-4 fallback launch starts from)
// out_i layout: [3 x (n_samples, header)] then 3 x 4 x cap contact flags
#if defined(__HIPCC__)
#define CHD_HD __host__ __device__
#else
#define CHD_HD
#endif
CHD_HD inline long long out_d_state_off(int cap) { return (long long)N_STAGES * RS_STRIDE + 3LL * 10 * cap * 3 + 24; }
CHD_HD inline long long out_d_size(int cap, int n_state) { return out_d_state_off(cap) + 3LL * n_state; }      // three state slots (one per snapshot): chd_kernels.hpp saved_state
inline long long out_i_size(int cap) { return 8 + 3LL * 4 * cap; }

}  // namespace chd

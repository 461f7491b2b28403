// chd_kinopt_kernels.hpp -- one `least_squares` solve of the kinematic optimisation for one video, on a CLUSTER of workgroups.
//
// Reference: optimize_trajectory.py:660-670 / :779-789 --
//     least_squares(fun_anim_for_projection, x0, jac=jac_anim_for_projection_sparse, max_nfev=50, gtol=1e-12, tr_solver='lsmr')
// i.e. SciPy's trust-region-reflective method without bounds in its 2-D subspace form (`trf_no_bounds`: Gauss-Newton direction
// from LSMR with the Cauchy-step regularisation, trust-region problem in span{g, gn}) on the residual of :324-483 with the
// Jacobian of :51-322.  The same source is compiled by hipcc for gfx950 and by g++ with -DCHD_HOST_EMU (the workgroups of a cluster
// emulated one after the other, phase by phase) for the CPU tests.
//
// Layout on the device (round 5).  Every row of the residual and every product touches the unknowns of at most three consecutive
// frames, and LSMR's state for one frame -- u (507 rows), v, h, the linearisation (positions, axes, projection coefficients: 420 doubles)
// and the products' per-joint intermediates -- is 1 353 doubles.  A 100-frame clip does not fit one compute unit's LDS; thirteen frames do.
// So a clip is solved by G = ceil(F / 13) workgroups (one per compute unit) that each OWN a run of consecutive frames:
//   * everything LSMR touches per iteration lives in the owner's LDS for the whole solve (flat pointers: the same code runs on a
//     slice in device memory when a clip is too long for 16 workgroups); the vectors only the outer iteration needs (x, g, the
//     residual, J g ...) are the owner's slices of device-memory arrays that no other workgroup ever touches;
//   * a workgroup handles (frame, joint) items of its slice, one per thread: no tile loop, each phase is one LDS round trip;
//   * what crosses a slice boundary is small and goes through one slot per workgroup in device memory: the two products need v of the
//     two frames AFTER the slice (174 doubles, from the right neighbour) and the smoothness / contact / Euler rows of u of the two
//     frames BEFORE it (423 doubles, from the left neighbour); every norm is a sum of G partial sums;
//   * LSMR therefore synchronises the cluster twice per iteration -- after the rows of u are written (|u|, u halo) and after v is
//     (|v|, v halo, and the three dot products that give |x| without a third round) -- by an all-gather of TAGGED values: every
//     8-byte granule a workgroup publishes carries 32 data bits and the number of the synchronisation it belongs to, written and
//     polled with agent-scope relaxed atomics (`sc1`: write-through / read at the memory side).  No flag, no fence: a release /
//     acquire pair at agent scope costs 17 us on this part (L2 write-back + invalidate), a flag behind acknowledged stores 2.5-3,
//     this 1.6-2.4 (tests/tools/cluster_sync_probe.hip, profiles/r05_experiments.md sections 7, 8b);
//   * a launch is persistent: as many clusters as the device holds resident (all workgroups of a cluster MUST be resident: they
//     spin on each other), each taking clips from a queue.  Clips of 13 frames or fewer are a cluster of one: no device-memory traffic at all.
//
// What is different from the reference, all exact in real arithmetic:
//  * the Jacobian is never formed (the reference allocates rows x (84 F) and rows x (87 F) dense arrays: 3.4 GB + 3.5 GB for
//    100 frames).  J = dE/dP * dP/dx: dE/dP is a handful of stencil coefficients per row, dP/dx per frame is
//    cross(axis_{j,a}, p_t - p_j) for joint j an ancestor of t (InverseKinematics.py:192-230), so
//        J v  : omega_j = sum_a v_{j,a} axis_{j,a};  dp_t = sum_{j anc t} omega_j x (p_t - p_j);  rows from dp
//        J^T u: lambda_t from the rows;  (J^T u)_{j,a} = axis_{j,a} . sum_{t desc j} (p_t - p_j) x lambda_t
//    with the linearisation cached per accepted point;
//  * rows are numbered frame by frame (507 per frame, the ones a clip's last frames do not have are kept at zero) instead of term by
//    term: a permutation of the reference's residual vector, invisible in every quantity the solve uses;
//  * every norm is a tree sum over the workgroup's 512 per-thread partial sums (KoAcc), then over the cluster in rank order: with a
//    single running sum per norm LSMR's iterate at its iteration limit drifts 5 % away from SciPy's and the solves end
//    1e-3 .. 3e-3 from the reference's instead of 1e-4 (tests/test_kinopt_emu.py);
//  * |x| in LSMR's stopping test is sqrt(x.x + 2 k2 x.hbar + k2^2 hbar.hbar) from three dot products taken before the step length k2 is
//    known (they ride on the synchronisation that yields it) instead of the norm of the updated x: equal up to rounding;
//  * forward kinematics with rotation matrices instead of quaternions;
//  * span{g, gn} is orthonormalised by Gram-Schmidt instead of Householder QR (same subspace, so the same step);
//  * the boundary solution of the 2-D trust-region problem is found on the angle parametrisation (scan + bisection of the
//    derivative) instead of the roots of the tangent-half-angle quartic (numpy.roots).
// Kept on purpose: the reference Jacobian's misplaced root column in the projection rows (`varIndex + 0`, :103-137) -- it
// decides which steps get rejected, i.e. where the solve stops.
#pragma once
#include <cmath>
#ifdef CHD_HOST_EMU
#include <cstdio>
#include <cstdlib>
#define KO_TRACE(...) do { if (std::getenv("KIN_TRACE")) std::fprintf(stderr, __VA_ARGS__); } while (0)
#else
#define KO_TRACE(...) ((void)0)
#endif

#ifdef CHD_HOST_EMU
#define KO_DEV static inline
#define KO_HD inline
#define KO_TID 0
#define KO_NT 1
#define KO_SYNC() ((void)0)
#define KO_CONST static const
#define KO_CLOCK() 0LL
#else
#include <hip/hip_runtime.h>
#define KO_DEV __device__ inline
#define KO_HD __host__ __device__ inline
#define KO_TID ((int)threadIdx.x)
#define KO_NT ((int)blockDim.x)
#define KO_SYNC() __syncthreads()
#define KO_CONST __constant__ const
#define KO_CLOCK() ((long long)wall_clock64())
#endif
#define KO_FOR(i, n) for (int i = KO_TID; i < (n); i += KO_NT)
// between two steps in which the 32 lanes that share a frame exchange values through LDS: the frame's lanes are in one wavefront, whose LDS operations execute
// in program order -- the compiler only has to keep them in that order
#ifdef CHD_HOST_EMU
#define KO_WSYNC() ((void)0)
#else
#define KO_WSYNC() do { __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront"); __builtin_amdgcn_wave_barrier(); } while (0)
#endif
// -DKIN_PROFILE: shader-clock ticks per segment of LSMR's iteration, summed per clip into stats[8 ..] (a study build; the shipped library does not read the clock here)
#if defined(KIN_PROFILE) && !defined(CHD_HOST_EMU)
#define KO_SEG(c, k) do { const long long t_ = (long long)clock64(); (c).seg[k] += t_ - (c).tlast; (c).tlast = t_; } while (0)
#else
#define KO_SEG(c, k) ((void)0)
#endif
#ifdef KIN_PROFILE
enum { KIN_STATS = 24 + 2 * 16 };               // + per rank of the cluster: ticks spent waiting in the two synchronisations of LSMR's iterations
#else
enum { KIN_STATS = 24 };                        // doubles of statistics per clip: 8 the ABI reports + 16 profile segments
#endif
// Pointers the compiler knows to be LDS (ds_read / ds_write instead of flat accesses that wait on both memory pipes): Sp<true>.  The host emulation has one kind.
#if defined(CHD_HOST_EMU) || defined(KIN_FLAT_LDS)
#define KO_LDSQ
template <bool L> struct Sp { typedef double* p; typedef const double* cp; typedef const int* ci; };
#else
#define KO_LDSQ __attribute__((address_space(3)))
template <bool L> struct Sp { typedef double* p; typedef const double* cp; typedef const int* ci; };
template <> struct Sp<true> { typedef KO_LDSQ double* p; typedef const KO_LDSQ double* cp; typedef const KO_LDSQ int* ci; };
#endif                        // doubles of statistics per clip: 8 the ABI reports + 16 profile segments

namespace chd_kin {

enum { NJ = 28, NV = 87, ROOT = 8, NR = 507 };
// the 507 rows of one frame: projection (2 per joint), velocity smoothness, acceleration smoothness, 3-D data, contact velocity (3 per joint),
// floor (1 per joint), Euler-angle smoothness (87).  Frames F-1 (velocity, contact velocity, Euler) and F-2, F-1 (acceleration) have no such rows.
enum { R_PROJ = 0, R_VEL = 56, R_ACC = 140, R_DATA = 224, R_CVEL = 308, R_FLOOR = 392, R_EUL = 420 };
enum { HALO_V = 2 * NV, HALO_U = 423, KC_PARTS = 4, KC_HALO = 424, KC_MAXG = 16 };
enum { N_X, N_XN, N_G, N_GN, N_V, N_H, N_HB, N_S0, N_S1, N_COUNT };          // the slices of the n-vectors a workgroup owns (87 per frame)
enum { M_FV, M_FN, M_U, M_T1, M_T2, M_COUNT };                                 // ... of the m-vectors (507 per frame)
// SkeletonDefinitions.py:64-137 (combined skeleton = body-25 + three spine joints).  Host tables: the kernel reads them from its LDS copy of KinParams (fwd, bwd, smooth_w)
static const int FWD[NJ] = {8, 12, 13, 14, 21, 19, 20, 9, 10, 11, 24, 22, 23, 25, 26, 27, 1, 0, 16, 18, 15, 17, 5, 6, 7, 2, 3, 4};      // skeleton joint -> data joint
static const int BWD[NJ] = {17, 16, 25, 26, 27, 22, 23, 24, 0, 7, 8, 9, 1, 2, 3, 20, 18, 21, 19, 5, 6, 4, 11, 12, 10, 13, 14, 15};       // data joint -> skeleton joint
static const double SMOOTH_W[NJ] = {2.5, 2.5, 2.5, 1.5, 1.0, 2.5, 1.5, 1.0, 1.0, 2.5, 1.5, 1.0, 2.5, 1.5, 1.0, 1.0, 1.0, 1.0, 1.0, 1.0, 1.0, 1.0, 1.0, 1.0, 1.0, 1.5, 1.5, 1.5};
KO_CONST double SMOOTH_VEL[3] = {1.0, 1.0, 2.0};          // optimize_trajectory.py:43-45
#define KO_SMOOTH_EULER 10.0                               // :46-48 (the same for the three angles)

struct KinParams {
  int parents[NJ];
  unsigned desc[NJ];             // bit t: joint t is a strict descendant of joint j (AnimationStructure.descendants_mask)
  int max_nfev;                  // 50
  double ftol, xtol, gtol;       // 1e-8, 1e-8, 1e-12
  double atol, btol, conlim;     // LSMR: 1e-6, 1e-6, 1e8
  int lsmr_maxiter;              // 0: min(m, n) as SciPy
  // the tree's walks as tables (built by the host from `parents`)
  // the walk over a joint's descendants, by lane of the frame (0 .. 27 the joints themselves, 28 .. 31 helpers that take over part of the longest walks): the joint
  // the walk belongs to, the candidates walk_t0 .. walk_t1 (descendants are contiguous when the joints are in depth-first order; else every later joint is a
  // candidate and the helpers stay idle), and for a joint's own lane the helpers whose partial sums it adds (bit h: lane 28 + h)
  int walk_of[32], walk_t0[32], walk_t1[32], walk_help[32];
  int anc_n[NJ]; unsigned char anc[NJ][8]; // strict ancestors of joint t, nearest first (the first eight; a deeper walk continues through `parents`)
  int fwd[NJ], bwd[NJ];          // FWD / BWD and SMOOTH_W below, for the kernel's copy of this struct in LDS (a per-lane index into constant memory is a device-memory load)
  double smooth_w[NJ];
};

// one video inside the batch pools
struct KinSeq {
  int F, n, m;                   // frames, unknowns 87 F, rows of the reference's residual (507 F - 423)
  long long o_const;             // double pool: offsets[84] | pose3d[F*84] | root_trans[F*3] | pose2d[F*56] | proj_w[F*28] | data_w[F*28]
  long long o_contact;           // int pool: contact[F*28]
  long long o_work;              // workspace pool (doubles), work_doubles(F, G)
  long long o_x;                 // state pool: x[F*87] (start point in, solution out)
  double floor_n[3], floor_p[3];
  double w[6];                   // projWeight, smoothWeightVel, smoothWeightAcc, dataWeight, velWeight, floorWeight
};

KO_HD int rows_of(int F) { return 507 * F - 423; }      // 56F + 84(F-1) + 84(F-2) + 84F + 84(F-1) + 28F + 87(F-1)
// device-memory workspace of one clip solved by G workgroups: 9 n-vectors, 5 m-vectors (frame-major, 507 per frame), and the per-frame arrays --
// P 84, E 252, RG 252, PN 84, LW 84, LD 84 with two halo frames per workgroup, C 84 and LL 84 without
// and the per-item constants DW 28 (doubles), CT 28 (ints)
KO_HD long long work_doubles(int F, int G) { return (long long)N_COUNT * NV * F + (long long)M_COUNT * NR * F + 840LL * (F + 2 * G) + 168LL * F + 42LL * F + 64; }
// frames per workgroup the LDS block holds: 1 395 doubles per frame + two halo frames of P / E / LW / LD + the received halos
enum { KIN_LDS_PER_FRAME = 1395, KIN_LDS_FIXED = 1008 + HALO_V + KC_HALO };
KO_HD int lds_frames(int lds_doubles) { const int f = (lds_doubles - KIN_LDS_FIXED) / KIN_LDS_PER_FRAME; return f < 0 ? 0 : f; }
// workgroups per clip: the fewest whose slices fit the LDS block (`cap` frames each), never more than 16, never slices of fewer than two frames
KO_HD int cluster_size(int F, int cap) {
  int G = cap > 0 ? (F + cap - 1) / cap : KC_MAXG;
  if (G > KC_MAXG) G = KC_MAXG;
  if (G > F / 2) G = F / 2;
  return G < 1 ? 1 : G;
}

// what a workgroup publishes at a synchronisation of its cluster (device memory; one per workgroup of the launch)
struct KinSlot {
  // 8-byte granules {32 data bits | 32-bit tag}, a double as two of them; the tag is the number of the synchronisation the value belongs to, so a reader needs no flag:
  // it polls the granule itself (an aligned 8-byte atomic is never torn).  By parity of the synchronisation's number: a workgroup can be one synchronisation
  // ahead of a neighbour that still reads.  First the KC_PARTS partial sums, then the halo.
  unsigned long long gran[2][2 * (KC_PARTS + KC_HALO)];
};

// a workgroup's share of a clip: frames [a, a + nf), local frame index l = f - a; arrays marked (+2) carry the two frames after the slice
struct KinWg {
  int g, a, nf, nh;              // rank in the cluster; first frame; own frames; own + halo frames that exist (min(nf + 2, F - a))
  double* nv[N_COUNT];           // nf x 87
  double* mv[M_COUNT];           // nf x 507
  double *P, *E, *RG, *PN;       // (+2) x 84 / 252 / 252 / 84: positions and axes of the linearisation, scratch of the forward kinematics, positions at a trial point
  double *C, *LW, *LD, *LL;      // projection coefficients nf x 84; the products' per-joint intermediates (+2) x 84, (+2) x 84, nf x 84
  double* DW; int* CT;           // per (frame, joint) item: dataWeight x data_w; bit 0 the joint is in contact in this frame, bit 1 it was in the frame before
  double *vh, *uh;               // received halos (LDS): 2 x 87 values of an n-vector from the right neighbour, HALO_U rows of an m-vector from the left one
  bool in_lds;                   // U, V, H, P, E, C, LW, LD, LL, DW, CT are in the LDS block (else: the slices in device memory)
#ifdef CHD_HOST_EMU
  double pub[KC_HALO];           // what this workgroup publishes at the next synchronisation
#endif
};

// a clip's constants and LSMR's scalar state: small structs the kernel keeps in LDS, out of the register file (what is live across the product phases
// otherwise ends up in scratch memory, and every phase then begins with a round trip to fetch its pointers)
struct KinClip {
  const KinSeq* q;
  int F; double wt[6], fn[3], fp[3];      // frame count; term weights (projWeight, smoothWeightVel, smoothWeightAcc, dataWeight, velWeight, floorWeight); floor
  const double *offs, *pose3d, *root_trans, *pose2d, *proj_w, *data_w;
  const int* contact;
};
struct KinLsmr {
  double normb, beta, alpha, su, sv, damp, ctol;
  double zetabar, alphabar, rho, rhobar, cbar, sbar, betadd, betad, rhodold, tautildeold, thetatilde, zeta, d, normA2, maxrbar, minrbar;
  double chat, shat, cc, s, rhoold, rhobarold, zetaold, thetabar, rhotemp;     // from the first half of an iteration for the second
  double k1, k2, k3, nx2;
  int itn, istop, maxiter, tested;        // `tested`: the last iteration whose stopping tests have run
};

struct KinCtx {
  KO_LDSQ KinClip* k; const KO_LDSQ KinParams* P;
  KO_LDSQ KinWg* wg;             // this workgroup's share (the emulation: all G of them)
  KO_LDSQ KinLsmr* S;
  KO_LDSQ double* gath;          // what the last synchronisation gathered: KC_PARTS values per workgroup of the cluster
  int G;                         // workgroups of the cluster
#ifndef CHD_HOST_EMU
  KinSlot* slots;                // the cluster's G slots
  unsigned long long epoch;      // synchronisations so far (the same in every workgroup of the cluster)
  long long patience;            // KC_PATIENCE (a test shortens it)
  int* abort_flag;               // device memory, one per launch: a workgroup that has waited KC_PATIENCE for a neighbour sets it, every workgroup that sees it gives up
  KO_LDSQ int* dead;             // this workgroup has given up (LDS; changes only inside kc_sync, before its last barrier)
  KO_LDSQ double* red;           // workgroup reduction scratch: KC_PARTS * 16 doubles
#endif
#ifndef CHD_HOST_EMU
  // what the tables say about THIS lane's joint (lane & 31: the same in every item a thread handles), fetched once per launch
  int lj_na, lj_f3, lj_b3, lj_wof, lj_wt0, lj_wt1, lj_whelp;
  unsigned lj_desc;
  unsigned long long lj_anc;                  // ancestors, eight bits each
#endif
  long long seg[16], tlast;      // KIN_PROFILE
  long long t_jv, t_jtu;         // wall-clock ticks in the two halves of LSMR's iterations (first thread's view; monitoring only)
};
// The thread that runs LSMR's scalar recurrences: the first lane of the LAST half wavefront, which has no (frame, joint) item while a slice of up to 14 frames is
// multiplied by J^T or its rows are written -- the recurrences then run beside those phases instead of between them.
#ifdef CHD_HOST_EMU
#define KO_IS_SCALAR() true
#else
#define KO_IS_SCALAR() ((int)threadIdx.x == (int)blockDim.x - 32)
#endif

// the tables' entries for joint j: the emulation looks them up, a device lane has its own joint's in registers
#ifdef CHD_HOST_EMU
KO_DEV int kj_na(const KinCtx& c, int j) { return c.P->anc_n[j]; }
KO_DEV int kj_f3(const KinCtx& c, int j) { return 3 * c.P->fwd[j]; }
KO_DEV int kj_b3(const KinCtx& c, int j) { return 3 * c.P->bwd[j]; }
KO_DEV int kj_wof(const KinCtx& c, int j) { return c.P->walk_of[j]; }
KO_DEV int kj_wt0(const KinCtx& c, int j) { return c.P->walk_t0[j]; }
KO_DEV int kj_wt1(const KinCtx& c, int j) { return c.P->walk_t1[j]; }
KO_DEV int kj_whelp(const KinCtx& c, int j) { return c.P->walk_help[j]; }
KO_DEV unsigned kj_desc(const KinCtx& c, int j) { return c.P->desc[c.P->walk_of[j]]; }
KO_DEV unsigned long long kj_anc(const KinCtx& c, int j) { unsigned long long v = 0; for (int q = 0; q < 8; ++q) v |= (unsigned long long)c.P->anc[j][q] << (8 * q); return v; }
#else
KO_DEV int kj_na(const KinCtx& c, int) { return c.lj_na; }
KO_DEV int kj_f3(const KinCtx& c, int) { return c.lj_f3; }
KO_DEV int kj_b3(const KinCtx& c, int) { return c.lj_b3; }
KO_DEV int kj_wof(const KinCtx& c, int) { return c.lj_wof; }
KO_DEV int kj_wt0(const KinCtx& c, int) { return c.lj_wt0; }
KO_DEV int kj_wt1(const KinCtx& c, int) { return c.lj_wt1; }
KO_DEV int kj_whelp(const KinCtx& c, int) { return c.lj_whelp; }
KO_DEV unsigned kj_desc(const KinCtx& c, int) { return c.lj_desc; }
KO_DEV unsigned long long kj_anc(const KinCtx& c, int) { return c.lj_anc; }
KO_DEV void kin_lane_tables(KinCtx& c) {
  const int j = threadIdx.x & 31, jj = j < NJ ? j : 0;
  c.lj_na = j < NJ ? c.P->anc_n[jj] : 0; c.lj_f3 = 3 * c.P->fwd[jj]; c.lj_b3 = 3 * c.P->bwd[jj];
  c.lj_wof = c.P->walk_of[j]; c.lj_wt0 = c.P->walk_t0[j]; c.lj_wt1 = c.P->walk_t1[j]; c.lj_whelp = c.P->walk_help[j]; c.lj_desc = c.P->desc[c.lj_wof];
  unsigned long long b = 0;
  for (int q = 0; q < 8; ++q) b |= (unsigned long long)c.P->anc[jj][q] << (8 * q);
  c.lj_anc = b;
}
#endif

// A launch whose workgroups are not all resident would wait for ever (its members spin on each other).  The host sizes the grid so that they are; should that
// fail all the same -- a partitioned device, another process holding compute units -- the wait is bounded: after KC_PATIENCE the launch winds down (every loop of
// the solve checks kc_dead) and the call returns an error instead of hanging the device.
#ifdef CHD_HOST_EMU
KO_DEV bool kc_dead(const KinCtx&) { return false; }
#else
#define KC_PATIENCE 500000000LL                 // wall_clock64 ticks (100 MHz): 5 s
KO_DEV bool kc_dead(const KinCtx& c) { return *c.dead != 0; }
KO_DEV unsigned long long kc_poll(KinCtx& c, const unsigned long long* g, unsigned tag, bool& ok) {      // the granule once it carries `tag`; ok = false: gave up
  unsigned long long v = __hip_atomic_load(g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if ((unsigned)v == tag) return v;
  const long long t0 = (long long)wall_clock64();
  unsigned spins = 0;
  for (;;) {
    __builtin_amdgcn_s_sleep(1);
    v = __hip_atomic_load(g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if ((unsigned)v == tag) return v;
    if ((++spins & 1023u) == 0 && (__hip_atomic_load(c.abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0 || (long long)wall_clock64() - t0 > c.patience)) {
      __hip_atomic_store(c.abort_flag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      *c.dead = 1;
      ok = false;
      return 0;
    }
  }
}
KO_DEV double kc_get(KinCtx& c, const unsigned long long* g2, unsigned tag, bool& ok) {
  const unsigned long long hi = kc_poll(c, g2, tag, ok), lo = ok ? kc_poll(c, g2 + 1, tag, ok) : 0ull;
  return __longlong_as_double((long long)((hi & 0xffffffff00000000ull) | (lo >> 32)));
}
#endif

// sum (or maximum) number i of the last synchronisation: the cluster's partial results in rank order -- the same bits in every workgroup
KO_DEV double kc_sum(const KinCtx& c, int i, bool mx = false) {
  double s = 0.0;
  for (int g = 0; g < c.G; ++g) { const double t = c.gath[KC_PARTS * g + i]; s = mx ? (t > s ? t : s) : s + t; }
  return s;
}

#ifdef CHD_HOST_EMU
enum { KC_NW = KC_MAXG };
#define KC_EACH(c, w) for (int wi = 0; wi < (c).G; ++wi) { KO_LDSQ KinWg& w = (c).wg[wi];
#else
enum { KC_NW = 1 };
#define KC_EACH(c, w) { const int wi = 0; KO_LDSQ KinWg& w = (c).wg[0];
#endif
#define KC_DONE }

// ---- sums over a workgroup's items and over the cluster (fixed trees: results do not depend on scheduling) -------------------------------
// Every thread adds its own items in loop order, the per-thread sums are combined by a butterfly inside each wavefront, then wavefront by
// wavefront, then workgroup by workgroup.  The host emulation keeps 512 lane accumulators per workgroup and combines them the same way, so that its
// sums round like the device's (a single running sum over ~50 000 terms is 1000 times less accurate, and LSMR's convergence over thousands of
// iterations feels that).
#ifdef CHD_HOST_EMU
enum { KO_LANES = 512 };
struct KoAcc {
  double l[KO_LANES];
  KoAcc() { for (int i = 0; i < KO_LANES; ++i) l[i] = 0.0; }
  void add(long long i, double v) { l[i & (KO_LANES - 1)] += v; }
  void hi(long long i, double v) { double& t = l[i & (KO_LANES - 1)]; t = v > t ? v : t; }
  double total(bool mx = false) const {
    double s = 0.0;
    for (int w0 = 0; w0 < KO_LANES; w0 += 64) {
      double a[64], t[64];
      for (int k = 0; k < 64; ++k) a[k] = l[w0 + k];
      for (int o = 32; o > 0; o >>= 1) { for (int k = 0; k < 64; ++k) t[k] = mx ? (a[k ^ o] > a[k] ? a[k ^ o] : a[k]) : a[k] + a[k ^ o]; for (int k = 0; k < 64; ++k) a[k] = t[k]; }
      s = mx ? (a[0] > s ? a[0] : s) : s + a[0];
    }
    return s;
  }
};
#define KC_PUB(c, w, i, val) ((w).pub[i] = (val))
#else
struct KoAcc {
  double s = 0.0;
  __device__ void add(long long, double v) { s += v; }
  __device__ void hi(long long, double v) { s = v > s ? v : s; }
};
KO_DEV void kc_put(unsigned long long* g2, double x, unsigned tag) {      // agent-scope relaxed atomics: write-through, visible to the other XCDs' sc1 loads
  const unsigned long long b = (unsigned long long)__double_as_longlong(x);
  __hip_atomic_store(g2, (b & 0xffffffff00000000ull) | tag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __hip_atomic_store(g2 + 1, (b << 32) | tag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
#define KC_PUB(c, w, i, val) kc_put(&(c).slots[(w).g].gran[((c).epoch + 1) & 1][2 * (KC_PARTS + (i))], (val), (unsigned)((c).epoch + 1))
#endif

// Sum over the joints of a frame of three values per joint, the joint `skip` left out: a butterfly over the frame's 32 lanes (the same bits in every lane).
// Device: every lane of the frame passes its own values (`mine`; lanes that are not joints, and `skip`, contribute zero).  Emulation: from the frame's 84 stored values.
#ifdef CHD_HOST_EMU
KO_DEV void ko_frame_sum3(const double* frame, const double*, int, int skip, double* out) {
  for (int k = 0; k < 3; ++k) {
    double a[32], t[32];
    for (int j = 0; j < 32; ++j) a[j] = (j < NJ && j != skip) ? frame[3 * j + k] : 0.0;
    for (int o = 16; o > 0; o >>= 1) { for (int j = 0; j < 32; ++j) t[j] = a[j] + a[j ^ o]; for (int j = 0; j < 32; ++j) a[j] = t[j]; }
    out[k] = a[0];
  }
}
#else
KO_DEV void ko_frame_sum3(const void*, const double* mine, int j, int skip, double* out) {
  for (int k = 0; k < 3; ++k) {
    double v = (j < NJ && j != skip) ? mine[k] : 0.0;
    for (int o = 16; o > 0; o >>= 1) v += __shfl_xor(v, o);
    out[k] = v;
  }
}
#endif

// One synchronisation of the cluster.  In: `np` accumulators per workgroup (acc[wi][0 .. np)), and whatever the workgroups have published with KC_PUB.
// Out: c.gath -- every workgroup's np partial sums (maxima with `mx`), read with kc_sum -- and, with dir = +1 / -1, the first `nrecv` values the right / left
// neighbour published in w.vh / w.uh (a workgroup without that neighbour keeps what it had).  Ends with a workgroup barrier.
KO_DEV void kc_sync(KinCtx& c, KoAcc (*acc)[KC_PARTS], int np, int dir = 0, int nrecv = 0, bool mx = false) {
#ifdef CHD_HOST_EMU
  for (int g = 0; g < c.G; ++g) for (int i = 0; i < np; ++i) c.gath[KC_PARTS * g + i] = acc[g][i].total(mx);
  if (dir != 0)
    for (int g = 0; g < c.G; ++g) {
      const int nb = g + dir;
      if (nb < 0 || nb >= c.G) continue;
      double* dst = dir > 0 ? c.wg[g].vh : c.wg[g].uh;
      for (int i = 0; i < nrecv; ++i) dst[i] = c.wg[nb].pub[i];
    }
#else
  if (kc_dead(c)) return;
  double v[KC_PARTS];
#pragma unroll
  for (int i = 0; i < KC_PARTS; ++i) {
    v[i] = i < np ? acc[0][i].s : 0.0;
    if (i < np) for (int o = 32; o > 0; o >>= 1) { const double t = __shfl_xor(v[i], o); v[i] = mx ? (t > v[i] ? t : v[i]) : v[i] + t; }
  }
  const int wv = threadIdx.x >> 6, nw = blockDim.x >> 6, ln = threadIdx.x & 63;
  __syncthreads();                              // the previous use of c.red / c.gath is over
  if (ln == 0)
    for (int i = 0; i < np; ++i) c.red[16 * i + wv] = v[i];
  __syncthreads();
  const int me = c.wg[0].g;
  if (c.G == 1) {                               // a cluster of one: nothing leaves the compute unit
    if ((int)threadIdx.x < np) {
      double s = 0.0;
      for (int k = 0; k < nw; ++k) { const double t = c.red[16 * threadIdx.x + k]; s = mx ? (t > s ? t : s) : s + t; }
      c.gath[threadIdx.x] = s;
    }
    __syncthreads();
    return;
  }
  // Every synchronisation gathers at least one value from EVERY rank (a zero when the caller has none): a workgroup thus never gets more than one synchronisation ahead
  // of any other, which is what the two-deep slots rely on.
  const unsigned long long e = ++c.epoch;
  const int par = (int)(e & 1);
  const unsigned tag = (unsigned)e;
  const int npp = np > 0 ? np : 1;
  KinSlot* mine = c.slots + me;
  if ((int)threadIdx.x < npp) {
    double s = 0.0;
    if ((int)threadIdx.x < np) for (int k = 0; k < nw; ++k) { const double t = c.red[16 * threadIdx.x + k]; s = mx ? (t > s ? t : s) : s + t; }
    kc_put(&mine->gran[par][2 * threadIdx.x], s, tag);
  }
  // first wavefront: lane g fetches workgroup g's partial sums; the other wavefronts fetch the neighbour's halo meanwhile.  No flag, no fence: a value is there when its tag is
  const int nb = me + dir;
  const bool want = dir != 0 && nb >= 0 && nb < c.G;
  bool ok = true;
  if (wv == 0) {
    if (ln < c.G) {
      const unsigned long long* o = c.slots[ln].gran[par];
#if defined(KIN_PROFILE)
      const long long tw_ = (long long)clock64();
#endif
      for (int i = 0; i < npp; ++i) { const double t = kc_get(c, o + 2 * i, tag, ok); if (i < np) c.gath[KC_PARTS * ln + i] = ok ? t : 0.0; }
#if defined(KIN_PROFILE)
      if (ln == 0) { const long long dw_ = (long long)clock64() - tw_; if (dir < 0) c.seg[12] += dw_; else c.seg[13] += dw_; }      // (lane 0 leaves the loop with the wavefront's last lane)
#endif
    }
  } else if (want) {
    const unsigned long long* o = c.slots[nb].gran[par] + 2 * KC_PARTS;
    KO_LDSQ double* dst = (KO_LDSQ double*)(dir > 0 ? c.wg[0].vh : c.wg[0].uh);
    for (int i = threadIdx.x - 64; i < nrecv && ok; i += blockDim.x - 64) { const double t = kc_get(c, o + 2 * i, tag, ok); if (ok) dst[i] = t; }
  }
  __syncthreads();
#endif
}

// entry k of local frame l of an n-vector slice: the slice itself, or for the two frames after it the received halo
template <bool VL> KO_DEV double nvf(const KO_LDSQ KinWg& w, typename Sp<VL>::cp v, int l, int k) {
  return l < w.nf ? v[NV * l + k] : ((typename Sp<true>::cp)w.vh)[NV * (l - w.nf) + k];
}
// row t of local frame l of an m-vector slice, l = -2, -1 being the last frames of the left neighbour (received halo: the acceleration rows of both, the
// velocity, contact-velocity and Euler rows of the nearer one)
KO_DEV int uh_index(int l, int t) { return l == -2 ? t - R_ACC : (t < R_ACC ? 84 + (t - R_VEL) : t < R_DATA ? 168 + (t - R_ACC) : t < R_FLOOR ? 252 + (t - R_CVEL) : 336 + (t - R_EUL)); }
template <bool VL> KO_DEV double mv_row(const KO_LDSQ KinWg& w, typename Sp<VL>::cp u, int l, int t) { return l >= 0 ? u[NR * l + t] : ((typename Sp<true>::cp)w.uh)[uh_index(l, t)]; }

// the same as a pointer, for the outer iteration's functions (generic addressing)
KO_DEV const double* nv_frame(const KO_LDSQ KinWg& w, const double* v, int l) { return l < w.nf ? v + NV * l : w.vh + NV * (l - w.nf); }

// publish the first two frames of n-vector `sel` (for the left neighbour) / the boundary rows of m-vector `sel` (for the right neighbour)
template <bool VL> KO_DEV void kin_pub_v(KinCtx& c, int sel) {
  KC_EACH(c, w)
    (void)wi;
    typename Sp<VL>::cp v = (typename Sp<VL>::cp)w.nv[sel];
    if (w.g > 0) KO_FOR(i, HALO_V) KC_PUB(c, w, i, v[i]);
  KC_DONE
}
template <bool VL> KO_DEV void kin_pub_u(KinCtx& c, int sel) {
  KC_EACH(c, w)
    (void)wi;
    if (w.g < c.G - 1) {
      typename Sp<VL>::cp u = (typename Sp<VL>::cp)w.mv[sel];
      KO_FOR(i, HALO_U) {
        const int l = i < 84 ? w.nf - 2 : w.nf - 1;
        const int t = i < 84 ? R_ACC + i : i < 168 ? R_VEL + (i - 84) : i < 252 ? R_ACC + (i - 168) : i < 336 ? R_CVEL + (i - 252) : R_EUL + (i - 336);
        KC_PUB(c, w, i, u[NR * l + t]);
      }
    }
  KC_DONE
}
KO_DEV void kin_exchange_v(KinCtx& c, int sel) { kin_pub_v<false>(c, sel); KoAcc none[KC_NW][KC_PARTS]; kc_sync(c, none, 0, +1, HALO_V); }
KO_DEV void kin_exchange_u(KinCtx& c, int sel) { kin_pub_u<false>(c, sel); KoAcc none[KC_NW][KC_PARTS]; kc_sync(c, none, 0, -1, HALO_U); }
// a function template picked by where the slice lives (L) and whether the vector arguments are LSMR's own, LDS-resident ones (lsmr)
#ifdef CHD_HOST_EMU
#define KIN_DISPATCH(fn, lsmr, ...) fn<false, false>(__VA_ARGS__)
#else
#define KIN_DISPATCH(fn, lsmr, ...) do { if (c.wg[0].in_lds) { if (lsmr) fn<true, true>(__VA_ARGS__); else fn<true, false>(__VA_ARGS__); } else fn<false, false>(__VA_ARGS__); } while (0)
#endif

// ---- forward kinematics of one frame (Animation.py:294-323, 379-414; Quaternions.from_euler(order='xyz', world=True)) -------
// R = Rz Ry Rx per joint; global rotation Rg_j = Rg_parent R_j; position p_j = p_parent + Rg_parent offset_j, root at 0
// (the fitted skeleton's root offset is zero; the root's translation is a separate unknown).  With `axes`: the rotation axes
// of the 84 angle unknowns (InverseKinematics.py:205-209): prs Rz Ry e_x, prs Rz e_y, prs e_z, prs = parent's global rotation.
KO_DEV void fk_frame(const KinCtx& c, const double* xf, double* Pf, double* RGf, double* Ef) {
  for (int j = 0; j < NJ; ++j) {
    const double ex = xf[3 + 3 * j], ey = xf[4 + 3 * j], ez = xf[5 + 3 * j];
    const double cx = std::cos(ex), sx = std::sin(ex), cy = std::cos(ey), sy = std::sin(ey), cz = std::cos(ez), sz = std::sin(ez);
    // Rz Ry Rx
    const double R[9] = {cz * cy, cz * sy * sx - sz * cx, cz * sy * cx + sz * sx,
                         sz * cy, sz * sy * sx + cz * cx, sz * sy * cx - cz * sx,
                         -sy, cy * sx, cy * cx};
    const int p = c.P->parents[j];
    double* Rg = RGf + 9 * j;
    if (p < 0) {
      for (int k = 0; k < 9; ++k) Rg[k] = R[k];
      Pf[0] = Pf[1] = Pf[2] = 0.0;
      if (Ef) {
        const double ax[9] = {cz * cy, sz * cy, -sy, -sz, cz, 0.0, 0.0, 0.0, 1.0};
        for (int k = 0; k < 9; ++k) Ef[k] = ax[k];
      }
    } else {
      const double* Rp = RGf + 9 * p;
      for (int r = 0; r < 3; ++r)
        for (int k = 0; k < 3; ++k) Rg[3 * r + k] = Rp[3 * r] * R[k] + Rp[3 * r + 1] * R[3 + k] + Rp[3 * r + 2] * R[6 + k];
      const double* o = c.k->offs + 3 * j;
      for (int r = 0; r < 3; ++r) Pf[3 * j + r] = Pf[3 * p + r] + Rp[3 * r] * o[0] + Rp[3 * r + 1] * o[1] + Rp[3 * r + 2] * o[2];
      if (Ef) {
        const double l0[3] = {cz * cy, sz * cy, -sy}, l1[3] = {-sz, cz, 0.0};
        double* e = Ef + 9 * j;
        for (int r = 0; r < 3; ++r) {
          e[r] = Rp[3 * r] * l0[0] + Rp[3 * r + 1] * l0[1] + Rp[3 * r + 2] * l0[2];
          e[3 + r] = Rp[3 * r] * l1[0] + Rp[3 * r + 1] * l1[1];
          e[6 + r] = Rp[3 * r + 2];
        }
      }
    }
  }
}


// ---- residual (:324-483) at n-vector `xsel` into m-vector `osel` ---------------------------------------------------------------------------
KO_DEV void kin_residual(KinCtx& c, int xsel, int osel) {
  const int F = c.k->F;
  kin_exchange_v(c, xsel);
  KC_EACH(c, w)
    KO_FOR(l, w.nh) fk_frame(c, nv_frame(w, w.nv[xsel], l), w.PN + 84 * l, w.RG + 252 * l, nullptr);
  KC_DONE
  KO_SYNC();
  const double pw = c.k->wt[0], sv = c.k->wt[1], sa = c.k->wt[2], dw = c.k->wt[3], vw = c.k->wt[4], fw = c.k->wt[5];
  KC_EACH(c, w)
    const double* x = w.nv[xsel];
    double* out = w.mv[osel];
    // y of the reference (:356-359) for data joint jd of local frame l: the root's entry is its translation, the others are root relative
    auto y_of = [&](int l, int jd, int k) { return jd == ROOT ? nv_frame(w, x, l)[k] : w.PN[84 * l + 3 * c.P->bwd[jd] + k]; };
    KO_FOR(idx, w.nf * NJ) {
      const int l = idx / NJ, jd = idx % NJ, f = w.a + l;
      const long long gi = (long long)f * NJ + jd;
      const bool has1 = f < F - 1, has2 = f < F - 2;
      double* r = out + NR * l;
      double y[3], yr[3];
      for (int k = 0; k < 3; ++k) { y[k] = y_of(l, jd, k); yr[k] = x[NV * l + k]; }
      {
        const double pwt = c.k->proj_w[gi];
        double r0 = 0, r1 = 0;
        if (pwt > 0) {
          const double ax = jd == ROOT ? yr[0] : y[0] + yr[0], ay = jd == ROOT ? yr[1] : y[1] + yr[1], az = jd == ROOT ? yr[2] : y[2] + yr[2];
          r0 = pw * pwt * (ax / az - c.k->pose2d[2 * gi]);
          r1 = pw * pwt * (ay / az - c.k->pose2d[2 * gi + 1]);
        }
        r[R_PROJ + 2 * jd] = r0; r[R_PROJ + 2 * jd + 1] = r1;
      }
      const bool ct = c.k->contact[gi] == 1;
      for (int k = 0; k < 3; ++k) {
        const double y1 = has1 ? y_of(l + 1, jd, k) : 0.0, y2 = has2 ? y_of(l + 2, jd, k) : 0.0;
        r[R_VEL + 3 * jd + k] = has1 ? sv * c.P->smooth_w[jd] * SMOOTH_VEL[k] * (y[k] - y1) : 0.0;
        r[R_ACC + 3 * jd + k] = has2 ? sa * ((y2 - y1) - (y1 - y[k])) : 0.0;
        const double tgt = jd == ROOT ? c.k->root_trans[3 * f + k] : c.k->pose3d[3 * gi + k];
        r[R_DATA + 3 * jd + k] = dw * (y[k] - tgt) * c.k->data_w[gi];
        r[R_CVEL + 3 * jd + k] = (has1 && ct) ? vw * ((yr[k] + y[k]) - (nv_frame(w, x, l + 1)[k] + y1)) : 0.0;
      }
      double d = 0;
      for (int k = 0; k < 3; ++k) d += c.k->fn[k] * (yr[k] + y[k] - c.k->fp[k]);
      r[R_FLOOR + jd] = ct ? fw * d : 0.0;
    }
    KO_FOR(idx, w.nf * NV) {
      const int l = idx / NV, i = idx % NV;
      out[NR * l + R_EUL + i] = (w.a + l < F - 1) ? sv * KO_SMOOTH_EULER * (x[idx] - nv_frame(w, x, l + 1)[i]) : 0.0;
    }
  KC_DONE
  KO_SYNC();
}

// ---- linearisation at n-vector `xsel`: positions and axes of the slice and its halo frames, projection coefficients of the slice ------------------
KO_DEV void kin_linearise(KinCtx& c, int xsel) {
  kin_exchange_v(c, xsel);
  KC_EACH(c, w)
    KO_FOR(l, w.nh) fk_frame(c, nv_frame(w, w.nv[xsel], l), w.P + 84 * l, w.RG + 252 * l, w.E + 252 * l);
  KC_DONE
  KO_SYNC();
  const double pw = c.k->wt[0];
  KC_EACH(c, w)
    const double* x = w.nv[xsel];
    KO_FOR(idx, w.nf * NJ) {
      const int l = idx / NJ, jd = idx % NJ;
      const double pwt = c.k->proj_w[(long long)(w.a + l) * NJ + jd];
      double cx = 0, czx = 0, czy = 0;
      if (pwt > 0) {
        double a[3];
        for (int k = 0; k < 3; ++k) a[k] = jd == ROOT ? x[NV * l + k] : w.P[84 * l + 3 * c.P->bwd[jd] + k] + x[NV * l + k];
        const double ww = pw * pwt;
        cx = ww / a[2]; czx = -ww * a[0] / (a[2] * a[2]); czy = -ww * a[1] / (a[2] * a[2]);
      }
      w.C[3 * idx] = cx; w.C[3 * idx + 1] = czx; w.C[3 * idx + 2] = czy;
    }
  KC_DONE
  KO_SYNC();
}

KO_DEV void kin_lsmr_tests(KinCtx& c);

// ---- the two products, matrix free, on the slice ----------------------------------------------------------------------------------------------
// A (frame, joint) item per lane, 32 lanes per frame (28 joints + 4 idle): a frame never straddles a wavefront, so the steps in which the joints of a frame
// exchange values (the walks up and down the tree, the two sums over a frame's joints) need no workgroup barrier, and what a lane needs to know about its
// joint (ancestors, descendants, the data / skeleton permutation) sits in registers for the whole launch.
// (Measured and dropped, profiles/r05_experiments.md section 8: recursions by tree level -- 28 cross products per frame instead of 101, but six dependent LDS
// round trips per product where the walks have one or two.)
//
// out = J v (:51-322 applied to a vector) for the rows of the slice; w.vh must hold v of the two frames after it (kin_exchange_v, or LSMR's second
// synchronisation).  With `fused`: the form LSMR's bidiagonalisation needs, out <- scale * (J v) + keep * out in place and acc[wi][0] += |out|^2 over the slice.
// L: the slice's arrays are in LDS; VL: so are v and out.
template <bool L, bool VL>
KO_DEV void kin_jv(KinCtx& c, int vsel, int osel, const double scale, const double keep, const bool fused, KoAcc (*acc)[KC_PARTS]) {
  typedef typename Sp<L>::p LP; typedef typename Sp<L>::cp LCP; typedef typename Sp<VL>::p VP; typedef typename Sp<VL>::cp VCP;
  const int F = c.k->F;
  const double sv = c.k->wt[1], sa = c.k->wt[2], vw = c.k->wt[4], fw = c.k->wt[5];
  const double se = sv * KO_SMOOTH_EULER;
  KC_EACH(c, w)
    (void)wi;
    VCP v = (VCP)w.nv[vsel]; LCP E = (LCP)w.E; LCP P = (LCP)w.P; LP LW = (LP)w.LW; LP LD = (LP)w.LD;
    KO_FOR(idx, w.nh * 32) {                  // omega_j = sum_a v_{j,a} axis_{j,a}
      const int l = idx >> 5, j = idx & 31;
      if (j >= NJ) continue;
      LCP e = E + 252 * l + 9 * j;
      const double v0 = nvf<VL>(w, v, l, 3 + 3 * j), v1 = nvf<VL>(w, v, l, 4 + 3 * j), v2 = nvf<VL>(w, v, l, 5 + 3 * j);
      for (int k = 0; k < 3; ++k) LW[84 * l + 3 * j + k] = v0 * e[k] + v1 * e[3 + k] + v2 * e[6 + k];
    }
    KO_WSYNC();
    KO_SEG(c, 0);
    KO_FOR(idx, w.nh * 32) {                  // dp_t = sum over the strict ancestors j of t of omega_j x (p_t - p_j), nearest first; stored in data order
      const int l = idx >> 5, t = idx & 31;
      if (t >= NJ) continue;
      double d[3] = {0, 0, 0};
      if (t == 0) { for (int k = 0; k < 3; ++k) d[k] = nvf<VL>(w, v, l, k); }
      else {
        LCP Pf = P + 84 * l; LCP Wf = LW + 84 * l;
        const double p0 = Pf[3 * t], p1 = Pf[3 * t + 1], p2 = Pf[3 * t + 2];
        const int na = kj_na(c, t);
        const unsigned long long al = kj_anc(c, t);
#pragma unroll 1
        for (int q0 = 0; q0 < na && q0 < 8; q0 += 4) {          // four ancestors at a time, all operands requested before the first is used
          double om[4][3], pj[4][3];
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const int j = (int)((al >> (8 * (q0 + q))) & 255);  // (beyond the walk: the table holds the joint itself)
            for (int k = 0; k < 3; ++k) { om[q][k] = Wf[3 * j + k]; pj[q][k] = Pf[3 * j + k]; }
          }
#pragma unroll
          for (int q = 0; q < 4; ++q)
            if (q0 + q < na) {
              const double r0 = p0 - pj[q][0], r1 = p1 - pj[q][1], r2 = p2 - pj[q][2];
              d[0] += om[q][1] * r2 - om[q][2] * r1; d[1] += om[q][2] * r0 - om[q][0] * r2; d[2] += om[q][0] * r1 - om[q][1] * r0;
            }
        }
        if (na > 8)
          for (int j = c.P->parents[(int)((al >> 56) & 255)]; j >= 0; j = c.P->parents[j]) {
            const double r0 = p0 - Pf[3 * j], r1 = p1 - Pf[3 * j + 1], r2 = p2 - Pf[3 * j + 2];
            d[0] += Wf[3 * j + 1] * r2 - Wf[3 * j + 2] * r1; d[1] += Wf[3 * j + 2] * r0 - Wf[3 * j] * r2; d[2] += Wf[3 * j] * r1 - Wf[3 * j + 1] * r0;
          }
      }
      const int ft = kj_f3(c, t);
      LD[84 * l + ft] = d[0]; LD[84 * l + ft + 1] = d[1]; LD[84 * l + ft + 2] = d[2];
    }
  KC_DONE
  KO_SYNC();
  KO_SEG(c, 1);
  KC_EACH(c, w)
    VCP v = (VCP)w.nv[vsel]; VP out = (VP)w.mv[osel]; LCP LD = (LCP)w.LD; LCP Cc = (LCP)w.C; LCP DW = (LCP)w.DW; typename Sp<L>::ci CT = (typename Sp<L>::ci)w.CT;
    KO_FOR(idx, w.nf * 32) {                  // the fifteen rows of (frame, joint)
      const int l = idx >> 5, jd = idx & 31, f = w.a + l;
      if (jd >= NJ) continue;
      const int it = NJ * l + jd;
      LCP dy = LD + 84 * l + 3 * jd;
      LCP d0 = LD + 84 * l;                                           // data joint 0: where the reference puts the root's projection derivative
      LCP dr = LD + 84 * l + 3 * ROOT;
      const double C0 = Cc[3 * it], C1 = Cc[3 * it + 1], C2 = Cc[3 * it + 2];
      const bool ct = (CT[it] & 1) != 0;
      const double dwj = DW[it], swj = sv * c.P->smooth_w[jd];
      const bool has1 = f < F - 1, has2 = f < F - 2;
      int rr[15]; double val[15];
      const double ex = jd == 0 ? dy[0] : d0[0] + dy[0], ey = jd == 0 ? dy[1] : d0[1] + dy[1], ez = jd == 0 ? dy[2] : d0[2] + dy[2];
      rr[0] = R_PROJ + 2 * jd; val[0] = C0 * ex + C1 * ez;
      rr[1] = R_PROJ + 2 * jd + 1; val[1] = C0 * ey + C2 * ez;
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        const int o = 2 + 4 * k;
        const double n1 = has1 ? dy[84 + k] : 0.0, n2 = has2 ? dy[168 + k] : 0.0, nr1 = has1 ? dr[84 + k] : 0.0;
        rr[o] = R_VEL + 3 * jd + k; val[o] = has1 ? swj * SMOOTH_VEL[k] * (dy[k] - n1) : 0.0;
        rr[o + 1] = R_CVEL + 3 * jd + k; val[o + 1] = (has1 && ct) ? vw * ((dr[k] + dy[k]) - (nr1 + n1)) : 0.0;
        rr[o + 2] = R_ACC + 3 * jd + k; val[o + 2] = has2 ? sa * (dy[k] - 2.0 * n1 + n2) : 0.0;
        rr[o + 3] = R_DATA + 3 * jd + k; val[o + 3] = dwj * dy[k];
      }
      double d = 0;
      for (int k = 0; k < 3; ++k) d += c.k->fn[k] * (dr[k] + dy[k]);
      rr[14] = R_FLOOR + jd; val[14] = ct ? fw * d : 0.0;
      VP r = out + NR * l;
      if (fused) {
        double old[15];
#pragma unroll
        for (int q = 0; q < 15; ++q) old[q] = r[rr[q]];
#pragma unroll
        for (int q = 0; q < 15; ++q) { val[q] = scale * val[q] + keep * old[q]; acc[wi][0].add(idx, val[q] * val[q]); }
      }
#pragma unroll
      for (int q = 0; q < 15; ++q) r[rr[q]] = val[q];
    }
    KO_FOR(idx, w.nf * NV) {                  // Euler-angle smoothness rows
      const int l = idx / NV, i = idx % NV;
      VP r = out + NR * l + R_EUL + i;
      double val = (w.a + l < F - 1) ? se * (v[idx] - nvf<VL>(w, v, l + 1, i)) : 0.0;
      if (fused) { val = scale * val + keep * *r; acc[wi][0].add(idx, val * val); }
      *r = val;
    }
  KC_DONE
  KO_SYNC();
  KO_SEG(c, 2);
}

// out = J^T u for the unknowns of the slice; w.uh must hold the boundary rows of u of the left neighbour (kin_exchange_u, or LSMR's first synchronisation).
// With `fused`: out <- scale * (J^T u) + keep * out in place and acc[wi][0] += |out|^2 over the slice.  No workgroup barrier inside: every step is within a frame.
template <bool L, bool VL>
KO_DEV void kin_jtu(KinCtx& c, int usel, int osel, const double scale_, const double keep_, const bool fused, KoAcc (*acc)[KC_PARTS], const bool late = false) {
  typedef typename Sp<L>::p LP; typedef typename Sp<L>::cp LCP; typedef typename Sp<VL>::p VP; typedef typename Sp<VL>::cp VCP;
  const int F = c.k->F;
  const double sv = c.k->wt[1], sa = c.k->wt[2], vw = c.k->wt[4], fw = c.k->wt[5];
  const double se = sv * KO_SMOOTH_EULER;
  KC_EACH(c, w)
    VCP u = (VCP)w.mv[usel]; VP out = (VP)w.nv[osel]; LP LQ = (LP)w.LW; LP LR = (LP)w.LD; LP LL = (LP)w.LL; LCP Cc = (LCP)w.C; LCP DW = (LCP)w.DW; LCP P = (LCP)w.P; LCP E = (LCP)w.E;
    typename Sp<L>::ci CT = (typename Sp<L>::ci)w.CT;
    KO_FOR(idx, w.nf * 32) {                  // each joint's own projection rows and contact rows acting on a position they reference
      const int l = idx >> 5, jd = idx & 31, f = w.a + l;
      if (jd >= NJ) continue;
      const int it = NJ * l + jd;
      const bool has1 = f < F - 1, hasm = f >= 1;
      VCP r = u + NR * l;
      const double C0 = Cc[3 * it], C1 = Cc[3 * it + 1], C2 = Cc[3 * it + 2];
      const double ux = r[R_PROJ + 2 * jd], uy = r[R_PROJ + 2 * jd + 1];
      const int cti = CT[it];
      const bool ct = (cti & 1) != 0, ctm = (cti & 2) != 0;
      const double uf = fw * r[R_FLOOR + jd];
      LQ[84 * l + 3 * jd] = C0 * ux; LQ[84 * l + 3 * jd + 1] = C0 * uy; LQ[84 * l + 3 * jd + 2] = C1 * ux + C2 * uy;
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        const double u5 = r[R_CVEL + 3 * jd + k], u5m = hasm ? mv_row<VL>(w, u, l - 1, R_CVEL + 3 * jd + k) : 0.0;
        double t = 0.0;
        t += (ct && has1) ? vw * u5 : 0.0;
        t += ct ? c.k->fn[k] * uf : 0.0;
        t -= ctm ? vw * u5m : 0.0;
        LR[84 * l + 3 * jd + k] = t;
      }
    }
    KO_WSYNC();
    KO_SEG(c, 5);
    KO_FOR(idx, w.nf * 32) {                  // lambda of data joint jd of frame f
      const int l = idx >> 5, jd = idx & 31, f = w.a + l;
      const bool act = jd < NJ;
      const int jq = act ? jd : 0;
      double q[3], r3[3], sq[3] = {0, 0, 0}, sr[3] = {0, 0, 0};
      for (int k = 0; k < 3; ++k) { q[k] = LQ[84 * l + 3 * jq + k]; r3[k] = LR[84 * l + 3 * jq + k]; }
#ifdef CHD_HOST_EMU
      if (jd == 0) ko_frame_sum3(LQ + 84 * l, q, jd, 0, sq);                // the misplaced root column: the other joints' projection terms
      if (jd == ROOT) ko_frame_sum3(LR + 84 * l, r3, jd, ROOT, sr);         // root + joint in the contact rows
#else
      ko_frame_sum3(nullptr, q, jd, 0, sq);
      ko_frame_sum3(nullptr, r3, jd, ROOT, sr);
#endif
      if (!act) continue;
      const int it = NJ * l + jd;
      double lam[3];
      for (int k = 0; k < 3; ++k) lam[k] = q[k] + r3[k];
      if (jd == 0) for (int k = 0; k < 3; ++k) lam[k] += sq[k];
      if (jd == ROOT) for (int k = 0; k < 3; ++k) lam[k] += sr[k];
      const double dwj = DW[it], swj = sv * c.P->smooth_w[jd];
      const bool b0 = f < F - 1, b1 = f >= 1, b2 = f < F - 2, b3 = f >= 1 && f - 1 < F - 2, b4 = f >= 2;
      VCP r = u + NR * l;
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        const double s = swj * SMOOTH_VEL[k];
        const double t0 = r[R_VEL + 3 * jd + k], t1 = b1 ? mv_row<VL>(w, u, l - 1, R_VEL + 3 * jd + k) : 0.0;
        const double t2 = r[R_ACC + 3 * jd + k], t3 = b3 ? mv_row<VL>(w, u, l - 1, R_ACC + 3 * jd + k) : 0.0, t4 = b4 ? mv_row<VL>(w, u, l - 2, R_ACC + 3 * jd + k) : 0.0;
        const double t5 = r[R_DATA + 3 * jd + k];
        lam[k] += b0 ? s * t0 : 0.0;
        lam[k] -= b1 ? s * t1 : 0.0;
        lam[k] += b2 ? sa * t2 : 0.0;
        lam[k] -= b3 ? 2.0 * sa * t3 : 0.0;
        lam[k] += b4 ? sa * t4 : 0.0;
        lam[k] += dwj * t5;
      }
      const int bj = kj_b3(c, jd);                // stored at the joint's skeleton index: the walk below then needs no table
      LL[84 * l + bj] = lam[0]; LL[84 * l + bj + 1] = lam[1]; LL[84 * l + bj + 2] = lam[2];
    }
    KO_WSYNC();
    KO_SEG(c, 6);
    LP MP = LR;                                 // (the contact terms are used up: the helpers' partial sums go here)
    KO_FOR(idx, w.nf * 32) {                  // M_j = sum over the strict descendants t of j of (p_t - p_j) x lambda_t: each lane its part of a walk
      const int l = idx >> 5, j = idx & 31;
      LCP Pf = P + 84 * l;
      const int jo = kj_wof(c, j), tbeg = kj_wt0(c, j), tend = kj_wt1(c, j);
      const double q0 = Pf[3 * jo], q1 = Pf[3 * jo + 1], q2 = Pf[3 * jo + 2];
      double M0 = 0, M1 = 0, M2 = 0;
      const unsigned mask = kj_desc(c, j);
#pragma unroll 1
      for (int t0 = tbeg; t0 <= tend; t0 += 4) {   // four candidates at a time, operands requested together
        double pt[4][3], lm[4][3];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int t = t0 + q <= tend ? t0 + q : jo;
          for (int k = 0; k < 3; ++k) { pt[q][k] = Pf[3 * t + k]; lm[q][k] = LL[84 * l + 3 * t + k]; }
        }
#pragma unroll
        for (int q = 0; q < 4; ++q)
          if (t0 + q <= tend && ((mask >> (t0 + q)) & 1u)) {
            const double r0 = pt[q][0] - q0, r1 = pt[q][1] - q1, r2 = pt[q][2] - q2;
            M0 += r1 * lm[q][2] - r2 * lm[q][1]; M1 += r2 * lm[q][0] - r0 * lm[q][2]; M2 += r0 * lm[q][1] - r1 * lm[q][0];
          }
      }
      LP o = j < NJ ? LQ + 84 * l + 3 * j : MP + 84 * l + 3 * (j - NJ);      // (the projection terms are used up too: a joint's own part waits here)
      o[0] = M0; o[1] = M1; o[2] = M2;
    }
    KO_WSYNC();
    double scale = scale_, keep = keep_;
    bool apply = true;
    if (late) {                                 // LSMR: su, beta, sv come from the scalar thread, which worked beside the steps above
      KO_SYNC();
      const double beta = c.S->beta;
      scale = c.S->su; keep = -beta * c.S->sv; apply = beta > 0;
    }
    if (apply)
    KO_FOR(idx, w.nf * 32) {                  // (J^T u)_{j,a} = axis_{j,a} . M_j, + the Euler-smoothness rows; the root's translation gets lambda of the root's data joint
      const int l = idx >> 5, j = idx & 31, f = w.a + l;
      if (j >= NJ) continue;
      double M0 = LQ[84 * l + 3 * j], M1 = LQ[84 * l + 3 * j + 1], M2 = LQ[84 * l + 3 * j + 2];
      const int hm = kj_whelp(c, j);
#pragma unroll
      for (int h = 0; h < 4; ++h)
        if ((hm >> h) & 1) { M0 += MP[84 * l + 3 * h]; M1 += MP[84 * l + 3 * h + 1]; M2 += MP[84 * l + 3 * h + 2]; }
      LCP e = E + 252 * l + 9 * j;
      LCP Lf = LL + 84 * l;
      VP o = out + NV * l;
      const bool has1 = f < F - 1, hasm = f >= 1, isroot = j == 0;
#pragma unroll
      for (int q = 0; q < 6; ++q) {             // three angles; the root translation for joint 0
        const bool live = q < 3 || isroot;
        if (!live) continue;
        const int kk = q < 3 ? 3 + 3 * j + q : q - 3;
        double eu = 0.0;
        eu += has1 ? se * u[NR * l + R_EUL + kk] : 0.0;
        eu -= hasm ? se * mv_row<VL>(w, u, l - 1, R_EUL + kk) : 0.0;
        double val = (q < 3 ? e[3 * q] * M0 + e[3 * q + 1] * M1 + e[3 * q + 2] * M2 : Lf[3 * c.P->bwd[ROOT] + (q - 3)]) + eu;
        if (fused) { val = scale * val + keep * o[kk]; acc[wi][0].add(idx, val * val); }
        o[kk] = val;
      }
    }
  KC_DONE
  KO_SYNC();
  KO_SEG(c, 7);
}

// ---- LSMR (Fong & Saunders 2011, as scipy.sparse.linalg.lsmr with x0 = None) -----------------------------------------------------
KO_DEV void sym_ortho(double a, double b, double& cc, double& s, double& r) {
  auto sgn = [](double v) { return v > 0 ? 1.0 : (v < 0 ? -1.0 : 0.0); };
  if (b == 0) { cc = sgn(a); s = 0; r = std::fabs(a); }
  else if (a == 0) { cc = 0; s = sgn(b); r = std::fabs(b); }
  else if (std::fabs(b) > std::fabs(a)) { const double tau = a / b; s = sgn(b) / std::sqrt(1 + tau * tau); cc = s * tau; r = b / s; }
  else { const double tau = b / a; cc = sgn(a) / std::sqrt(1 + tau * tau); s = cc * tau; r = a / cc; }
}


// The second half's estimates and stopping tests of iteration S.itn (scipy lsmr, after the update of x), run by the scalar thread beside the next iteration's J^T U.
// Must come BEFORE that iteration's first rotations (they overwrite beta, the rotation and thetabar ...).
KO_DEV void kin_lsmr_tests(KinCtx& c) {
  KO_LDSQ KinLsmr& S = *c.S;
  const int itn = S.itn;
  if (itn <= S.tested) return;
  S.tested = itn;
  const double beta = S.beta, alpha = S.alpha, cc = S.cc, s = S.s, chat = S.chat, shat = S.shat, thetabar = S.thetabar, rhobarold = S.rhobarold, zetaold = S.zetaold, rhotemp = S.rhotemp;
  const double rhobar = S.rhobar, zeta = S.zeta;
  double ctildeold, stildeold, rhotildeold;
  const double normx = S.nx2 > 0 ? std::sqrt(S.nx2) : 0.0;
  const double betaacute = chat * S.betadd, betacheck = -shat * S.betadd;
  const double betahat = cc * betaacute;
  S.betadd = -s * betaacute;
  const double thetatildeold = S.thetatilde;
  sym_ortho(S.rhodold, thetabar, ctildeold, stildeold, rhotildeold);
  S.thetatilde = stildeold * rhobar;
  S.rhodold = ctildeold * rhobar;
  S.betad = -stildeold * S.betad + ctildeold * betahat;
  S.tautildeold = (zetaold - thetatildeold * S.tautildeold) / rhotildeold;
  const double taud = (zeta - S.thetatilde * S.tautildeold) / S.rhodold;
  S.d += betacheck * betacheck;
  const double normr = std::sqrt(S.d + (S.betad - taud) * (S.betad - taud) + S.betadd * S.betadd);
  S.normA2 += beta * beta;
  const double normA = std::sqrt(S.normA2);
  S.normA2 += alpha * alpha;
  S.maxrbar = S.maxrbar > rhobarold ? S.maxrbar : rhobarold;
  if (itn > 1) S.minrbar = S.minrbar < rhobarold ? S.minrbar : rhobarold;
  const double condA = (S.maxrbar > rhotemp ? S.maxrbar : rhotemp) / (S.minrbar < rhotemp ? S.minrbar : rhotemp);
  const double normar = std::fabs(S.zetabar);
  const double normb = S.normb;
  const double test1 = normr / normb;
  const double test2 = normA * normr != 0 ? normar / (normA * normr) : INFINITY;
  const double test3 = 1 / condA;
  const double t1 = test1 / (1 + normA * normx / normb);
  const double rtol = c.P->btol + c.P->atol * normA * normx / normb;
  int st = 0;
  if (itn >= S.maxiter) st = 7;
  if (1 + test3 <= 1) st = 6;
  if (1 + test2 <= 1) st = 5;
  if (1 + t1 <= 1) st = 4;
  if (test3 <= S.ctol) st = 3;
  if (test2 <= c.P->atol) st = 2;
  if (test1 <= rtol) st = 1;
  S.istop = st;
}

// min |J x - b|^2 + damp^2 |x|^2 into the GN slices, b = the FV slices; uses U (m), V, H, HB (n).  Returns the iteration count.
// The Golub-Kahan vectors are kept UNNORMALISED (u = su * U, v = sv * V): each half step is one
// fused product, beta u = A v - alpha u  ->  U <- sv * (J V) - (alpha su) * U,  beta = |U|,  su = 1 / beta, and the same for V.
// Two synchronisations of the cluster per iteration:
//   1. after the rows of U: |U|^2 and the boundary rows of U for the right neighbour.  beta fixes rho and with it k1: hbar <- k1 hbar + h can follow at once;
//   2. after V: |V|^2, the first two frames of V for the left neighbour, and x.x, x.hbar, hbar.hbar -- alpha then fixes k2, k3: x <- x + k2 hbar, h <- k3 h + sv v,
//      |x|^2 = x.x + 2 k2 x.hbar + k2^2 hbar.hbar.
// The scalar recurrences (four plane rotations, the estimates of |r|, |A|, cond A, the stopping tests: ~30 divisions and square roots in a dependent chain) run
// in ONE thread (KO_IS_SCALAR), on a state kept in LDS (KinLsmr), and as far as the data allows BESIDE the products instead of between them:
//   * the first half's rotations (beta -> rho, k1) while the other wavefronts compute J^T U -- only its last step (V <- su J^T U - beta sv V) needs them;
//   * of the second half only alpha, the rotation and k2, k3 (what the vector updates need) stay between the synchronisation and the updates; the estimates and
//     the stopping tests of iteration k run in the same window of iteration k + 1, and are looked at after that iteration's second synchronisation, BEFORE its
//     update of x: a stop costs one wasted iteration (which touched U, V and h-bar: nothing reads them after the solve), about one in four hundred.
template <bool L>
KO_DEV int kin_lsmr_on(KinCtx& c, double damp, int* istop_out) {
  typedef typename Sp<L>::p LP; typedef typename Sp<L>::cp LCP;
  KO_LDSQ KinLsmr& S = *c.S;
  // LSMR's x, h-bar and h: three entries per thread at most when the slice is in LDS (13 frames x 87 <= 3 x 512): in registers for the whole solve -- no
  // device-memory traffic inside the loop; x goes to its slice once at the end.  (Slices in device memory, and the emulation: the arrays.)
  double xr[3] = {0.0, 0.0, 0.0}, hbr[3] = {0.0, 0.0, 0.0}, hr[3] = {0.0, 0.0, 0.0};
  KoAcc acc[KC_NW][KC_PARTS];
  KC_EACH(c, w)
    const double* b = w.mv[M_FV]; LP u = (LP)w.mv[M_U]; LP v = (LP)w.nv[N_V];
    KO_FOR(i, w.nf * NR) { const double t = b[i]; u[i] = t; acc[wi][0].add(i, t * t); }
    KO_FOR(i, w.nf * NV) { if (!L) { w.nv[N_GN][i] = 0; w.nv[N_HB][i] = 0; } v[i] = 0; }
  KC_DONE
  KO_SYNC();
  kin_pub_u<L>(c, M_U);
  kc_sync(c, acc, 1, -1, HALO_U);
  if (KO_IS_SCALAR()) {
    const long long n = c.k->q->n, m = c.k->q->m;
    S.maxiter = c.P->lsmr_maxiter > 0 ? c.P->lsmr_maxiter : (int)(m < n ? m : n);
    S.damp = damp; S.ctol = c.P->conlim > 0 ? 1 / c.P->conlim : 0;
    S.normb = std::sqrt(kc_sum(c, 0));
    S.beta = S.normb; S.alpha = 0; S.su = S.beta > 0 ? 1 / S.beta : 0.0; S.sv = 0;
  }
  KO_SYNC();
  if (S.beta > 0) {
    KoAcc a2[KC_NW][KC_PARTS];
    const long long t0_ = KO_CLOCK();
    kin_jtu<L, L>(c, M_U, N_V, S.su, 0.0, true, a2);
    kin_pub_v<L>(c, N_V);
    kc_sync(c, a2, 1, +1, HALO_V);
    c.t_jtu += KO_CLOCK() - t0_;
    if (KO_IS_SCALAR()) S.alpha = std::sqrt(kc_sum(c, 0));
  }
  if (KO_IS_SCALAR()) {
    if (S.alpha > 0) S.sv = 1 / S.alpha;
    S.itn = 0; S.istop = 0; S.tested = 0;
    S.zetabar = S.alpha * S.beta; S.alphabar = S.alpha; S.rho = 1; S.rhobar = 1; S.cbar = 1; S.sbar = 0;
    S.betadd = S.beta; S.betad = 0; S.rhodold = 1; S.tautildeold = 0; S.thetatilde = 0; S.zeta = 0; S.d = 0;
    S.normA2 = S.alpha * S.alpha; S.maxrbar = 0; S.minrbar = 1e100;
  }
  KO_SYNC();
  {
    const double sv0 = S.sv;
    KC_EACH(c, w)
      (void)wi;
      LP h = (LP)w.nv[N_H]; LCP v = (LCP)w.nv[N_V];
      if (L) { for (int q = 0; q < 3; ++q) { const int i = KO_TID + q * KO_NT; if (i < w.nf * NV) hr[q] = sv0 * v[i]; } }
      else KO_FOR(i, w.nf * NV) h[i] = sv0 * v[i];
    KC_DONE
  }
  KO_SYNC();
  if (S.alpha * S.beta == 0 || S.normb == 0) {
    if (L) {
      KC_EACH(c, w)
        (void)wi;
        for (int q = 0; q < 3; ++q) { const int i = KO_TID + q * KO_NT; if (i < w.nf * NV) w.nv[N_GN][i] = 0.0; }
      KC_DONE
      KO_SYNC();
    }
    *istop_out = 0; return 0;
  }
#if defined(KIN_PROFILE) && !defined(CHD_HOST_EMU)
  c.tlast = (long long)clock64();
#endif
  const int maxiter = S.maxiter;
  int itn = 0, istop = 0;
  while (true) {
    if (kc_dead(c)) break;
    KO_SEG(c, 15);
    {
      KoAcc b2[KC_NW][KC_PARTS];
      const long long t0_ = KO_CLOCK();
      kin_jv<L, L>(c, N_V, M_U, S.sv, -S.alpha * S.su, true, b2);
      kin_pub_u<L>(c, M_U);
      kc_sync(c, b2, 1, -1, HALO_U);
      KO_SEG(c, 3);
      c.t_jv += KO_CLOCK() - t0_;
    }
    ++itn;
    if (KO_IS_SCALAR()) {                       // beside J^T U: the stopping tests of the iteration before, then this iteration's first rotations
      const double beta = std::sqrt(kc_sum(c, 0));
      kin_lsmr_tests(c);
      double chat, shat, alphahat, cc, s, rho;
      sym_ortho(S.alphabar, S.damp, chat, shat, alphahat);
      const double rhoold = S.rho;
      sym_ortho(alphahat, beta, cc, s, rho);
      S.beta = beta; S.chat = chat; S.shat = shat; S.cc = cc; S.s = s; S.rhoold = rhoold; S.rho = rho;
      S.rhobarold = S.rhobar; S.zetaold = S.zeta; S.thetabar = S.sbar * rho; S.rhotemp = S.cbar * rho;
      S.k1 = -(S.thetabar * rho / (rhoold * S.rhobarold));
      if (beta > 0) S.su = 1 / beta;
    }
    KO_SEG(c, 4);
    {
      KoAcc a2[KC_NW][KC_PARTS];
      const long long t0_ = KO_CLOCK();
      kin_jtu<L, L>(c, M_U, N_V, 0.0, 0.0, true, a2, true);                 // (its last step waits for the scalars above and takes su, beta, sv from the state)
      kin_pub_v<L>(c, N_V);
      const double k1 = S.k1;
      KC_EACH(c, w)
        double* __restrict__ hb = w.nv[N_HB]; LCP h = (LCP)w.nv[N_H]; const double* __restrict__ x = w.nv[N_GN];
        if (L) {
#pragma unroll
          for (int q = 0; q < 3; ++q) {
            const int i = KO_TID + q * KO_NT;
            if (i < w.nf * NV) { const double t = hbr[q] * k1 + hr[q], xi = xr[q]; hbr[q] = t; a2[wi][1].add(i, xi * xi); a2[wi][2].add(i, xi * t); a2[wi][3].add(i, t * t); }
          }
        } else
        KO_FOR(i, w.nf * NV) {
          const double t = hb[i] * k1 + h[i], xi = x[i];
          hb[i] = t;
          a2[wi][1].add(i, xi * xi); a2[wi][2].add(i, xi * t); a2[wi][3].add(i, t * t);
        }
      KC_DONE
      KO_SEG(c, 8);
      kc_sync(c, a2, 4, +1, HALO_V);
      KO_SEG(c, 9);
      c.t_jtu += KO_CLOCK() - t0_;
    }
    istop = S.istop;                            // (of the iteration before: x has not moved since)
    if (istop > 0) { --itn; break; }
    if (KO_IS_SCALAR()) {                       // what the vector updates need
      const double beta = S.beta, cc = S.cc, s = S.s, rho = S.rho;
      double alpha = S.alpha;
      if (beta > 0) {
        alpha = std::sqrt(kc_sum(c, 0));
        if (alpha > 0) S.sv = 1 / alpha;
      }
      S.alpha = alpha;
      double cbar, sbar, rhobar;
      const double thetanew = s * alpha;
      S.alphabar = cc * alpha;
      sym_ortho(S.cbar * rho, thetanew, cbar, sbar, rhobar);
      S.cbar = cbar; S.sbar = sbar; S.rhobar = rhobar;
      const double zeta = cbar * S.zetabar;
      S.zeta = zeta;
      S.zetabar = -sbar * S.zetabar;
      const double k2 = zeta / (rho * rhobar);
      S.k2 = k2; S.k3 = -(thetanew / rho);
      S.nx2 = kc_sum(c, 1) + 2.0 * k2 * kc_sum(c, 2) + k2 * k2 * kc_sum(c, 3);
      S.itn = itn;
    }
    KO_SYNC();
    {
      const double k2 = S.k2, k3 = S.k3, sv = S.sv;
      KC_EACH(c, w)
        (void)wi;
        double* __restrict__ x = w.nv[N_GN]; LP h = (LP)w.nv[N_H]; const double* __restrict__ hb = w.nv[N_HB]; LCP v = (LCP)w.nv[N_V];
        if (L) {
#pragma unroll
          for (int q = 0; q < 3; ++q) { const int i = KO_TID + q * KO_NT; if (i < w.nf * NV) { xr[q] = xr[q] + k2 * hbr[q]; hr[q] = hr[q] * k3 + sv * v[i]; } }
        } else
        KO_FOR(i, w.nf * NV) { x[i] = x[i] + k2 * hb[i]; h[i] = h[i] * k3 + sv * v[i]; }
      KC_DONE
    }
    KO_SEG(c, 10);
  }
  KO_SYNC();
  if (L) {
    KC_EACH(c, w)
      (void)wi;
      for (int q = 0; q < 3; ++q) { const int i = KO_TID + q * KO_NT; if (i < w.nf * NV) w.nv[N_GN][i] = xr[q]; }
    KC_DONE
  }
  KO_SYNC();
  *istop_out = istop;
  return itn;
}
KO_DEV int kin_lsmr(KinCtx& c, double damp, int* istop_out) {
#ifndef CHD_HOST_EMU
  if (c.wg[0].in_lds) return kin_lsmr_on<true>(c, damp, istop_out);
#endif
  return kin_lsmr_on<false>(c, damp, istop_out);
}

// ---- 2-D trust-region problem (scipy solve_trust_region_2d) -------------------------------------------------------------------------
KO_DEV void tr2d(double B00, double B01, double B11, double g0, double g1, double Delta, double& p0, double& p1) {
  if (B00 > 0) {                              // Cholesky succeeds <=> both leading minors positive
    const double l00 = std::sqrt(B00), l10 = B01 / l00, d11 = B11 - l10 * l10;
    if (d11 > 0) {
      const double l11 = std::sqrt(d11);
      const double z0 = -g0 / l00, z1 = (-g1 - l10 * z0) / l11;
      const double q1 = z1 / l11, q0 = (z0 - l10 * q1) / l00;
      if (q0 * q0 + q1 * q1 <= Delta * Delta) { p0 = q0; p1 = q1; return; }
    }
  }
  // boundary p = Delta (sin phi, cos phi): the global minimiser of a degree-2 trigonometric polynomial
  const double a = B00 * Delta * Delta, b = B01 * Delta * Delta, cq = B11 * Delta * Delta, dl = g0 * Delta, fl = g1 * Delta;
  auto val = [&](double ph) { const double s = std::sin(ph), co = std::cos(ph); return 0.5 * (a * s * s + 2 * b * s * co + cq * co * co) + dl * s + fl * co; };
  auto der = [&](double ph) { const double s = std::sin(ph), co = std::cos(ph); return (a - cq) * s * co + b * (co * co - s * s) + dl * co - fl * s; };
  const int K = 360;
  const double step = 6.283185307179586476925 / K;
  double best = 0, bestv = val(0.0);
  for (int i = 0; i < K; ++i) {
    const double lo = i * step, hi = (i + 1) * step;
    double dlo = der(lo), dhi = der(hi);
    if (dlo <= 0 && dhi >= 0) {               // a minimum inside [lo, hi]
      double x0 = lo, x1 = hi;
      for (int it = 0; it < 60; ++it) { const double mid = 0.5 * (x0 + x1); if (der(mid) <= 0) x0 = mid; else x1 = mid; }
      const double ph = 0.5 * (x0 + x1), v = val(ph);
      if (v < bestv) { bestv = v; best = ph; }
    }
    const double v = val(lo);
    if (v < bestv) { bestv = v; best = lo; }
  }
  p0 = Delta * std::sin(best); p1 = Delta * std::cos(best);
}


// sums over the slices: up to three dot products of vectors picked by (kind 'n' / 'm', index) in one synchronisation
struct KinDot { char kind; int a, b; };
KO_DEV void kin_dots(KinCtx& c, const KinDot* d, int nd, double* out) {
  KoAcc acc[KC_NW][KC_PARTS];
  KC_EACH(c, w)
    for (int k = 0; k < nd; ++k) {
      const bool isn = d[k].kind == 'n';
      const double* a = isn ? w.nv[d[k].a] : w.mv[d[k].a];
      const double* b = isn ? w.nv[d[k].b] : w.mv[d[k].b];
      KO_FOR(i, w.nf * (isn ? (int)NV : (int)NR)) acc[wi][k].add(i, a[i] * b[i]);
    }
  KC_DONE
  kc_sync(c, acc, nd);
  for (int k = 0; k < nd; ++k) out[k] = kc_sum(c, k);
}
KO_DEV double kin_dot(KinCtx& c, char kind, int a, int b) { const KinDot d = {kind, a, b}; double r; kin_dots(c, &d, 1, &r); return r; }

// ---- the solve: scipy trf_no_bounds (x_scale = 1, linear loss, tr_solver = 'lsmr', regularize = True) ---------------------------------
// stats: cost, nfev, njev, status, total LSMR iterations, last gradient infinity norm, shares of the time in LSMR's two halves
KO_DEV void kin_solve(KinCtx& c, double* xio, double* stats) {
  const KO_LDSQ KinParams& P = *c.P;
  c.t_jv = c.t_jtu = 0;
  for (int k = 0; k < 16; ++k) c.seg[k] = 0;
  const long long t_begin = KO_CLOCK();
  KC_EACH(c, w)
    KO_FOR(i, w.nf * NV) w.nv[N_X][i] = xio[(long long)NV * w.a + i];
    KO_FOR(i, HALO_V) w.vh[i] = 0.0;
    KO_FOR(i, HALO_U) w.uh[i] = 0.0;
    KO_FOR(idx, w.nf * NJ) {                  // the items' constants
      const long long gi = (long long)w.a * NJ + idx;
      w.DW[idx] = c.k->wt[3] * c.k->data_w[gi];
      w.CT[idx] = (c.k->contact[gi] == 1 ? 1 : 0) | ((gi >= NJ && c.k->contact[gi - NJ] == 1) ? 2 : 0);
    }
  KC_DONE
  KO_SYNC();
  kin_residual(c, N_X, M_FV);
  int nfev = 1, njev = 1, status = -1;
  long long lsmr_total = 0;
  kin_linearise(c, N_X);
  double cost = 0.5 * kin_dot(c, 'm', M_FV, M_FV);
  { kin_exchange_u(c, M_FV); KoAcc none[KC_NW][KC_PARTS]; KIN_DISPATCH(kin_jtu, false, c, M_FV, N_G, 1.0, 0.0, false, none); }
  double Delta = std::sqrt(kin_dot(c, 'n', N_X, N_X));
  if (Delta == 0) Delta = 1.0;
  double g_norm = 0;
  while (true) {
    if (kc_dead(c)) break;
    {
      KoAcc gm[KC_NW][KC_PARTS];
      KC_EACH(c, w)
        KO_FOR(i, w.nf * NV) gm[wi][0].hi(i, std::fabs(w.nv[N_G][i]));
      KC_DONE
      kc_sync(c, gm, 1, 0, 0, true);
      g_norm = kc_sum(c, 0, true);
    }
    if (g_norm < P.gtol) status = 1;
    if (status != -1 || nfev == P.max_nfev) break;
    // regularisation from the Cauchy step (build_quadratic_1d / minimize_quadratic_1d)
    KC_EACH(c, w)
      KO_FOR(i, w.nf * NV) w.nv[N_S0][i] = -w.nv[N_G][i];
    KC_DONE
    KO_SYNC();
    KoAcc none[KC_NW][KC_PARTS];
    kin_exchange_v(c, N_S0);
    KIN_DISPATCH(kin_jv, false, c, N_S0, M_T1, 1.0, 0.0, false, none);
    double gg, a;
    { const KinDot dd[2] = {{'m', M_T1, M_T1}, {'n', N_G, N_G}}; double r[2]; kin_dots(c, dd, 2, r); a = 0.5 * r[0]; gg = r[1]; }
    const double b = -gg, to_tr = Delta / std::sqrt(gg);
    double ag = 0.0;                            // t = 0
    { const double y1 = to_tr * (a * to_tr + b); if (y1 < ag) ag = y1; }
    if (a != 0) { const double ext = -0.5 * b / a; if (0 < ext && ext < to_tr) { const double y2 = ext * (a * ext + b); if (y2 < ag) ag = y2; } }
    const double reg_term = -ag / (Delta * Delta);
    int istop = 0;
    { const int it_ = kin_lsmr(c, std::sqrt(reg_term), &istop); lsmr_total += it_; KO_TRACE("iter cost %.10g Delta %.10g gnorm %.6g damp %.10g lsmr %d istop %d\n", cost, Delta, g_norm, std::sqrt(reg_term), it_, istop); }
    // orthonormal basis of span{g, gn}
    const double ng = std::sqrt(gg);
    KC_EACH(c, w)
      KO_FOR(i, w.nf * NV) w.nv[N_S0][i] = w.nv[N_G][i] / ng;
    KC_DONE
    KO_SYNC();
    double pr = kin_dot(c, 'n', N_S0, N_GN);
    KC_EACH(c, w)
      KO_FOR(i, w.nf * NV) w.nv[N_S1][i] = w.nv[N_GN][i] - pr * w.nv[N_S0][i];
    KC_DONE
    KO_SYNC();
    pr = kin_dot(c, 'n', N_S0, N_S1);           // second pass
    double s1;
    {
      KoAcc sa[KC_NW][KC_PARTS];
      KC_EACH(c, w)
        KO_FOR(i, w.nf * NV) { const double t = w.nv[N_S1][i] - pr * w.nv[N_S0][i]; w.nv[N_S1][i] = t; sa[wi][0].add(i, t * t); }
      KC_DONE
      kc_sync(c, sa, 1);
      s1 = std::sqrt(kc_sum(c, 0));
    }
    const double is1 = s1 > 0 ? 1 / s1 : 0.0;
    KC_EACH(c, w)
      KO_FOR(i, w.nf * NV) w.nv[N_S1][i] *= is1;
    KC_DONE
    KO_SYNC();
    kin_exchange_v(c, N_S0);
    KIN_DISPATCH(kin_jv, false, c, N_S0, M_T1, 1.0, 0.0, false, none);
    kin_exchange_v(c, N_S1);
    KIN_DISPATCH(kin_jv, false, c, N_S1, M_T2, 1.0, 0.0, false, none);
    double B00, B01, B11, gS0, gS1;
    { const KinDot dd[3] = {{'m', M_T1, M_T1}, {'m', M_T1, M_T2}, {'m', M_T2, M_T2}}; double r[3]; kin_dots(c, dd, 3, r); B00 = r[0]; B01 = r[1]; B11 = r[2]; }
    { const KinDot dd[2] = {{'n', N_S0, N_G}, {'n', N_S1, N_G}}; double r[2]; kin_dots(c, dd, 2, r); gS0 = r[0]; gS1 = r[1]; }
    double actual = -1, cost_new = cost;
    while (actual <= 0 && nfev < P.max_nfev && !kc_dead(c)) {
      double p0, p1;
      tr2d(B00, B01, B11, gS0, gS1, Delta, p0, p1);
      const double predicted = -(0.5 * (B00 * p0 * p0 + 2 * B01 * p0 * p1 + B11 * p1 * p1) + gS0 * p0 + gS1 * p1);
      const double step_norm = std::sqrt(p0 * p0 + p1 * p1);          // |S p| with orthonormal S
      KC_EACH(c, w)
        KO_FOR(i, w.nf * NV) w.nv[N_XN][i] = w.nv[N_X][i] + (p0 * w.nv[N_S0][i] + p1 * w.nv[N_S1][i]);
      KC_DONE
      KO_SYNC();
      kin_residual(c, N_XN, M_FN);
      ++nfev;
      double cn, bad, xx;
      {
        KoAcc sa[KC_NW][KC_PARTS];
        KC_EACH(c, w)
          KO_FOR(i, w.nf * NR) { const double t = w.mv[M_FN][i]; sa[wi][0].add(i, t * t); if (!(std::fabs(t) <= 1.79e308)) sa[wi][1].add(i, 1.0); }
          KO_FOR(i, w.nf * NV) sa[wi][2].add(i, w.nv[N_X][i] * w.nv[N_X][i]);
        KC_DONE
        kc_sync(c, sa, 3);
        cn = kc_sum(c, 0); bad = kc_sum(c, 1); xx = kc_sum(c, 2);
      }
      if (bad > 0) { Delta = 0.25 * step_norm; continue; }
      cost_new = 0.5 * cn;
      actual = cost - cost_new;
      KO_TRACE("   try p (%.10g, %.10g) pred %.10g actual %.10g B (%.8g %.8g %.8g) gS (%.8g %.8g)\n", p0, p1, predicted, actual, B00, B01, B11, gS0, gS1);
      double ratio;
      if (predicted > 0) ratio = actual / predicted;
      else if (predicted == 0 && actual == 0) ratio = 1;
      else ratio = 0;
      double Delta_new = Delta;
      if (ratio < 0.25) Delta_new = 0.25 * step_norm;
      else if (ratio > 0.75 && step_norm > 0.95 * Delta) Delta_new = 2.0 * Delta;
      const bool ftol_ok = actual < P.ftol * cost && ratio > 0.25;
      const bool xtol_ok = step_norm < P.xtol * (P.xtol + std::sqrt(xx));
      if (ftol_ok && xtol_ok) status = 4; else if (ftol_ok) status = 2; else if (xtol_ok) status = 3;
      if (status != -1) break;
      Delta = Delta_new;
    }
    if (actual > 0) {
      KC_EACH(c, w)
        KO_FOR(i, w.nf * NV) w.nv[N_X][i] = w.nv[N_XN][i];
        KO_FOR(i, w.nf * NR) w.mv[M_FV][i] = w.mv[M_FN][i];
      KC_DONE
      cost = cost_new;
      KO_SYNC();
      kin_linearise(c, N_X);
      ++njev;
      kin_exchange_u(c, M_FV);
      KIN_DISPATCH(kin_jtu, false, c, M_FV, N_G, 1.0, 0.0, false, none);
    }
  }
  KC_EACH(c, w)
    KO_FOR(i, w.nf * NV) xio[(long long)NV * w.a + i] = w.nv[N_X][i];
    if (w.g == 0 && KO_TID == 0) {
      stats[0] = cost; stats[1] = nfev; stats[2] = njev; stats[3] = status == -1 ? 0 : status; stats[4] = (double)lsmr_total; stats[5] = g_norm;
      const double tall = (double)(KO_CLOCK() - t_begin);
      stats[6] = tall > 0 ? (double)c.t_jv / tall : 0.0; stats[7] = tall > 0 ? (double)c.t_jtu / tall : 0.0;
      for (int k = 0; k < 16; ++k) stats[8 + k] = (double)c.seg[k];
    }
#if defined(KIN_PROFILE) && !defined(CHD_HOST_EMU)
    if (KO_TID == 0 && w.g < 16) { stats[24 + 2 * w.g] = (double)c.seg[12]; stats[25 + 2 * w.g] = (double)c.seg[13]; }
#endif
  KC_DONE
  KO_SYNC();
}

// ---- binding a clip to the cluster ---------------------------------------------------------------------------------------------------------------------
// (the LDS structs are filled by ONE thread; the caller puts a barrier behind kin_bind_clip + kin_bind_wg)
KO_DEV void kin_bind_clip(KO_LDSQ KinClip& k, const KinSeq* q, const double* dpool, const int* ipool) {
  k.q = q;
  const int F = q->F;
  k.F = F;
  for (int i = 0; i < 6; ++i) k.wt[i] = q->w[i];
  for (int i = 0; i < 3; ++i) { k.fn[i] = q->floor_n[i]; k.fp[i] = q->floor_p[i]; }
  const double* d = dpool + q->o_const;
  k.offs = d; d += 84; k.pose3d = d; d += 84LL * F; k.root_trans = d; d += 3LL * F; k.pose2d = d; d += 56LL * F; k.proj_w = d; d += 28LL * F; k.data_w = d;
  k.contact = ipool + q->o_contact;
}
// workgroup g of G: its frames, its slices of the clip's device-memory workspace, and -- when the slice fits -- the LDS block for everything LSMR's products touch
KO_DEV void kin_bind_wg(const KinCtx& c, KO_LDSQ KinWg& w, int g, double* work, double* lds, int lds_doubles) {
  const int F = c.k->F, G = c.G;
  w.g = g; w.a = (int)((long long)g * F / G); w.nf = (int)((long long)(g + 1) * F / G) - w.a;
  w.nh = F - w.a < w.nf + 2 ? F - w.a : w.nf + 2;
  double* b = work + c.k->q->o_work;
  for (int i = 0; i < N_COUNT; ++i) { w.nv[i] = b + (long long)NV * w.a; b += (long long)NV * F; }
  for (int i = 0; i < M_COUNT; ++i) { w.mv[i] = b + (long long)NR * w.a; b += (long long)NR * F; }
  const long long FH = F + 2LL * G, ah = w.a + 2LL * g;
  w.P = b + 84 * ah; b += 84 * FH; w.E = b + 252 * ah; b += 252 * FH; w.RG = b + 252 * ah; b += 252 * FH; w.PN = b + 84 * ah; b += 84 * FH;
  w.LW = b + 84 * ah; b += 84 * FH; w.LD = b + 84 * ah; b += 84 * FH;
  w.C = b + 84LL * w.a; b += 84LL * F; w.LL = b + 84LL * w.a; b += 84LL * F;
  w.DW = b + 28LL * w.a; b += 28LL * F; w.CT = (int*)b + 28LL * w.a;
  // LDS: the received halos always; the slice when it fits (h-bar and LSMR's x stay in device memory: one streaming pass per iteration)
  double* l = lds;
  w.vh = l; l += HALO_V; w.uh = l; l += KC_HALO;
  const int nf = w.nf;
  w.in_lds = (long long)KIN_LDS_FIXED + (long long)KIN_LDS_PER_FRAME * nf <= lds_doubles;
  if (w.in_lds) {
    w.mv[M_U] = l; l += NR * nf; w.nv[N_V] = l; l += NV * nf; w.nv[N_H] = l; l += NV * nf;
    w.P = l; l += 84 * (nf + 2); w.E = l; l += 252 * (nf + 2); w.C = l; l += 84 * nf;
    w.LW = l; l += 84 * (nf + 2); w.LD = l; l += 84 * (nf + 2); w.LL = l; l += 84 * nf;
    w.DW = l; l += 28 * nf; w.CT = (int*)l;
  }
}

}  // namespace chd_kin

/* chd_kinopt.h -- C ABI of the kinematic optimisation's least-squares solves.
 *
 * Replaces, for a whole batch of videos at once, each of the two calls
 *     cur_sol = least_squares(fun_anim_for_projection, init_sol, max_nfev=50, jac=jac_anim_for_projection_sparse,
 *                             gtol=1e-12, bounds=[-inf, inf], tr_solver='lsmr', args=(skeleton, poses3D, root_pos, ...))
 * of the reference's `optimize_trajectory` (src/optimize/optimize_trajectory.py:660-670 and :779-789; residual :324-483,
 * Jacobian :51-322, skeleton tables src/optimize/SkeletonDefinitions.py:64-137; SciPy: trust-region-reflective without
 * bounds, 2-D subspace + LSMR).  The steps around the two solves (bone lengths, weights, IK initialisation through
 * chd_ik.h, Huber floor fit, contact relabelling, outputs) are host code: contact-human-dynamics_amd/kinematic_optimizer.py.
 * Plain pointers and sizes only; all arrays are caller-owned host buffers of IEEE doubles / 32-bit ints.
 * The skeleton is the reference's combined 28-joint skeleton (body-25 + three spine joints): "skeleton order" is the joint
 * order of skeleton_fitting/combined_body_25.bvh, "data order" the order of the 2D / 3D estimates (body-25, then the spine).
 */
#ifndef CHD_KINOPT_H
#define CHD_KINOPT_H

#ifdef __cplusplus
extern "C" {
#endif

#define CHD_KIN_JOINTS 28
#define CHD_KIN_UNKNOWNS_PER_FRAME 87     /* root translation + 28 Euler triples (xyz, world order) */

typedef struct chd_kin_config {
  int max_nfev;            /* 50     (optimize_trajectory.py:661) */
  double ftol, xtol, gtol; /* 1e-8, 1e-8 (SciPy defaults), 1e-12 (:664) */
  double lsmr_atol, lsmr_btol, lsmr_conlim;   /* 1e-6, 1e-6, 1e8 (SciPy's lsmr defaults; least_squares passes no tr_options) */
  int lsmr_maxiter;        /* 0 = min(rows, unknowns), SciPy's default */
  int parents[CHD_KIN_JOINTS];   /* skeleton.parents (BVH order); parents[0] = -1, parents[j] < j */
  int reserved[4];         /* tuning knobs, 0 = default: [1] doubles of LDS per workgroup (default 19 760 = 154 KB: slices of up to 13 frames stay in LDS),
                              [2] frames per workgroup (default: what the LDS block holds; smaller = more workgroups per clip, larger = slices that live in
                              device memory), [3] = 0x7e57: test hook (one workgroup of the first cluster never shows up: the call must come back
                              with an error after the waiting limit, not hang), [0] = 1: no retry with one workgroup per clip when a launch turns out not to be fully resident (below).  A clip of F frames is solved by a cluster of ceil(F / frames-per-workgroup) workgroups (at most
                              16, never slices of fewer than two frames).  Results are bitwise reproducible for fixed values and independent of the batch a clip is in. */
} chd_kin_config;

/* One video, one solve: the `args` tuple of :667-670 after `optimize_trajectory`'s own preparation (:544-572). */
typedef struct chd_kin_seq {
  int n_frames;                /* F >= 3 */
  const double* offsets;       /* 28 x 3: bone offsets of the fitted skeleton (update_skeleton, :485-520), skeleton order, root row zero */
  const double* pose3d;        /* F x 28 x 3: poses3D, data order, root relative */
  const double* root_trans;    /* F x 3: root_pos */
  const double* pose2d_n;      /* F x 28 x 2: joints_2d_normalized (:568-569) */
  const double* proj_w;        /* F x 28: proj_weights (:564, :571) */
  const double* data_w;        /* F x 28: data_weights (:566, :572) */
  const int* contact;          /* F x 28: velConstraints == 1, data order */
  double floor_n[3], floor_p[3];   /* plane_normal, plane_point */
  double w_proj, w_smooth_vel, w_smooth_acc, w_data, w_vel, w_floor;   /* :630-635 / :773-778 */
  double* x;                   /* F x 87: init_sol in, cur_sol.x out */
  double cost;                 /* out: cur_sol.cost */
  int nfev, njev, status;      /* out: as scipy.optimize.OptimizeResult (status 0 max_nfev, 1 gtol, 2 ftol, 3 xtol, 4 both) */
  int lsmr_iterations;         /* out: LSMR iterations summed over the solve */
  double optimality;           /* out: infinity norm of the gradient at the solution */
  double jv_fraction, jtu_fraction;   /* out (monitoring): share of the solve's device time spent in the products J v and J^T u inside LSMR */
} chd_kin_seq;

const char* chd_kin_version(void);
void chd_kin_config_default(chd_kin_config* cfg);      /* the reference's values; parents of combined_body_25.bvh */

/* Solves B (video, stage) problems on HIP device `device`, each on a cluster of workgroups whose members own runs of consecutive frames (LSMR's state
 * in their compute units' LDS; two small exchanges between neighbours per LSMR iteration).  The launch is persistent -- as many clusters as the device
 * holds resident, taking clips from a queue -- and its workgroups wait on each other, so calls from several host threads take turns on the device.
 * Returns 0 on success; non-zero with a message in chd_kin_last_error() (no device, bad sizes, allocation failure).  There is no CPU path. */
int chd_kin_solve_batch(const chd_kin_config* cfg, int device, int B, chd_kin_seq* seqs);
const char* chd_kin_last_error(void);

/* Device time (HIP events around the launch) of the last successful call on this thread, in milliseconds. */
double chd_kin_last_kernel_ms(void);

/* The workgroups of a cluster wait on each other: the launch assumes they are all resident, which holds on a device that is exclusive to the process.  When a
 * cluster's bounded wait (5 s) expires -- another process / rank on the same GPU, a compute-unit mask, a long co-tenant kernel -- chd_kin_solve_batch solves the
 * batch once more with ONE workgroup per clip (slices in device memory, no wait across compute units; slower, results equal to rounding) instead of failing,
 * unless cfg->reserved[0] == 1.  1 if the last call on this thread took that path. */
int chd_kin_last_call_retried(void);

#ifdef __cplusplus
}
#endif
#endif

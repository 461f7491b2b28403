/* chd_ik.h -- C ABI of the IK back-projection step (first "next" row after the physics hot path).
 *
 * Replaces, for a whole batch of videos at once, the call
 *     ik = JacobianInverseKinematicsCK(anim, targetmap, translate=True, iterations=30, smoothness=0.001, damping=7.0); ik()
 * of the reference's `apply_results` (src/utils/towr_utils.py:841-843; solver: src/skeleton_fitting/ik/
 * InverseKinematics.py:326-561; forward kinematics: Animation.py:294-323, 379-414; Euler / quaternion conventions:
 * Quaternions.py:215-227, 401-420).  Plain pointers and sizes only; all arrays are caller-owned host buffers of IEEE
 * doubles / 32-bit ints.  Quaternions are (w, x, y, z).
 *
 * Checked against vectors produced by the reference's own solver, on the MI355X (tests/test_ik_gpu.py) and through the host
 * emulation of the kernel source (tests/test_ik_emu.py).  The same entry point serves the IK initialisation of the
 * kinematic optimisation (translate = 0, iterations = 200, smoothness = 0; optimize_trajectory.py:611-617).
 */
#ifndef CHD_IK_H
#define CHD_IK_H

#ifdef __cplusplus
extern "C" {
#endif

#define CHD_IK_MAX_JOINTS 64      /* joints of a skeleton (the reference's characters have 25-31 + 2 added heels) */
#define CHD_IK_MAX_TARGETS 26     /* targeted joints (apply_results: upper-body joints + 2 toes + 2 heels, towr_utils.py:826-840;
                                     the kinematic optimisation's initialisation: 25 of the 28 combined-skeleton joints, optimize_trajectory.py:605-617) */

typedef struct chd_ik_config {
  int iterations;        /* 30   (towr_utils.py:843) */
  int translate;         /* 1    (every joint's translation is an unknown as well) */
  double damping;        /* 7.0 */
  double smoothness;     /* 0.001 */
  double gamma;          /* 1.0  (InverseKinematics.py:451) */
} chd_ik_config;

/* One video.  rot / pos are the local joint rotations / translations of the animation handed to the solver
 * (apply_results has already replaced the root's by the optimised COM trajectory, towr_utils.py:820-823). */
typedef struct chd_ik_seq {
  int n_frames;              /* F */
  int n_joints;              /* J <= CHD_IK_MAX_JOINTS; joint 0 is the root; parents[j] < j */
  const int* parents;        /* J; parents[0] = -1 (Animation.parents) */
  int n_targets;             /* T <= CHD_IK_MAX_TARGETS */
  const int* target_joints;  /* T: keys of `targetmap`, in its iteration order */
  const double* targets;     /* T x F x 3: values of `targetmap` (global positions, centimetres in the reference) */
  const double* rot_in;      /* F x J x 4 */
  const double* pos_in;      /* F x J x 3 */
  double* rot_out;           /* F x J x 4: anim.rotations after ik() */
  double* pos_out;           /* F x J x 3: anim.positions after ik() */
} chd_ik_seq;

const char* chd_ik_version(void);
void chd_ik_config_default(chd_ik_config* cfg);

/* Solves B videos on HIP device `device`.  Returns 0 on success; non-zero with a message in chd_ik_last_error()
 * (no device, bad sizes, allocation failure).  There is no CPU path. */
int chd_ik_solve_batch(const chd_ik_config* cfg, int device, int B, const chd_ik_seq* seqs);
const char* chd_ik_last_error(void);

/* Device time of the last successful chd_ik_solve_batch on the calling thread: milliseconds between HIP events placed
 * around its `iterations` kernel launches (uploads, downloads and allocation excluded), and the number of (video, frame)
 * workgroups per launch.  For the roofline accounting of this row ("Next row"). */
double chd_ik_last_kernel_ms(void);
long long chd_ik_last_frames(void);

#ifdef __cplusplus
}
#endif
#endif

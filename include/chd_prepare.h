/*
 * chd_prepare.h — C ABI of libchd_prepare.so: the producer side of the physics stage's inputs.
 *
 * Reference being replaced: `prepare_input` (src/utils/towr_utils.py:451-777), which the driver starts as one child process per video
 * (scripts/run_phys_mocap.py:137-150: `python utils/towr_utils.py --anim <bvh> --floor ... --contacts ... --out phys_optim_in_<char>`), and the BVH reader
 * it begins with (src/skeleton_fitting/ik/BVH.py:25-168).  Two pieces:
 *
 *   chd_bvh_load_batch   reads a LIST of BVH files on the host's cores (token-stream parser of the HIERARCHY / MOTION grammar with the semantics of BVH.load:
 *                        End Sites are not joints, rotation order from the first CHANNELS line, Euler angles in degrees composed in local order, every joint
 *                        taken to have as many channels as the LAST joint declares) -> local rotations as quaternions, local translations, offsets, parents;
 *   chd_prep_frames      the per-frame numerics of prepare_input for ANY number of frames of ONE skeleton (the frames of all clips of a run, concatenated) in
 *                        one HIP kernel launch on a gfx950 device, one thread per frame: the two forward-kinematics passes (towr_utils.py:483-535 with root
 *                        rotation / translation zeroed, :542-655 as animated, heels appended), centre of mass from the segment tables (:803-810), hip offsets,
 *                        inertia about the centre of mass, toe / heel trajectories, toe-heel distance -- in the solver's frame (z up, metres, x and y negated).
 *
 * What stays on the host (sequential per clip, a few hundred operations): unwrapping the root's Euler angles (:620-629), contact run lengths (:695-725), the
 * scalars of skel_info.txt, writing the four files.  Conventions as chd_phys.h: plain C, 0 on success, no CPU fallback for chd_prep_frames.
 */
#ifndef CHD_PREPARE_H
#define CHD_PREPARE_H

#ifdef __cplusplus
extern "C" {
#endif

#define CHD_PREP_ABI_VERSION 2
#define CHD_PREP_MAX_JOINTS 64
#define CHD_PREP_MAX_SEGMENTS 32
#define CHD_PREP_MAX_SEGMENT_JOINTS 256
#define CHD_PREP_OUT_STRIDE 28      /* per frame: hip_l[3] hip_r[3] inertia[6: xx yy zz xy xz yz] com[3] ltoe[3] lheel[3] rtoe[3] rheel[3] toe_heel_distance */

/* The character's tables (character_info_utils.py getters used by towr_utils.py:466-481) for one skeleton hierarchy. */
typedef struct chd_prep_skeleton {
  int n_joints;                 /* joints of the animation as pass 2 sees it (heels included: the file's joints + 2 when they had to be appended, towr_utils.py:401-423) */
  int n_joints_body;            /* joints of the file itself: what pass 1 (root zeroed) and the segment tables refer to */
  int parents[CHD_PREP_MAX_JOINTS];
  int n_segments;
  int seg_first[CHD_PREP_MAX_SEGMENTS + 1];           /* segment s owns seg_joint[seg_first[s] .. seg_first[s + 1]) */
  int seg_joint[CHD_PREP_MAX_SEGMENT_JOINTS];
  double seg_mass_fraction[CHD_PREP_MAX_SEGMENTS];    /* get_character_seg_to_mass_perc_map x 0.01 */
  double mass;                                        /* get_character_mass (kg) */
  int hip_inds[2], toe_inds[2], heel_inds[2];         /* [left, right]; heel_inds = the appended joints (n_joints - 2, n_joints - 1) when the character has none */
} chd_prep_skeleton;

int chd_prep_version(void);
/* rot: n_frames x n_joints x 4 quaternions (w, x, y, z), pos: n_frames x n_joints x 3 local translations (centimetres, y up), both host pointers;
 * out: n_frames x CHD_PREP_OUT_STRIDE doubles.  Fails (< 0) without a HIP device. */
int chd_prep_frames(const chd_prep_skeleton* skel, int device, long long n_frames, const double* rot, const double* pos, double* out);
double chd_prep_last_kernel_ms(void);
const char* chd_prep_last_error(void);

/* One parsed BVH file.  All arrays are owned by the library until chd_bvh_free. */
typedef struct chd_bvh_clip {
  int n_frames, n_joints, channels;     /* channels per joint as the LAST joint declares them: 3, 6 or 9 */
  double frame_time;
  char order[4];                        /* rotation order of the first CHANNELS line, e.g. "zyx" */
  char* names;                          /* joint names joined by '\n' */
  int* parents;                         /* n_joints, -1 for the root */
  double* offsets;                      /* n_joints x 3 */
  double* positions;                    /* n_frames x n_joints x 3 */
  double* rotations;                    /* n_frames x n_joints x 4 (w, x, y, z) */
  char* error;                          /* NULL, or why this file could not be read (the other files of the batch are unaffected) */
} chd_bvh_clip;
/* Parses n files on up to n_threads host threads (0 = all cores).  Returns the number of files that failed (their `error` is set); < 0 on bad arguments. */
int chd_bvh_load_batch(int n, const char* const* paths, int n_threads, chd_bvh_clip* out /* n entries */);
void chd_bvh_free(int n, chd_bvh_clip* clips);

/* ---- the two JSON inputs in front of the kinematic optimisation and the contact network (ABI version 2; host code, csrc/chd_json.hpp) -----------------------------
 * Reference readers: openpose_utils.py:48-76 (one result file per frame, people[0].pose_keypoints_2d; the contact network's data set and
 * src/optimize/kinematic_optimizer.py:60-75 read the same directory) and totalcap_utils.py:33-79 (monocular total capture's tracked_results.json).  Both are
 * read for ALL videos of a run on the host's cores.  A file that is not the JSON these formats use fails ITS clip with a message naming it. */
typedef struct chd_keypoint_clip {
  int n_frames;                         /* result files of the directory whose name ends in .json, in name order */
  double* data;                         /* n_frames x num_joints x 3 (x, y, confidence); zeros for a frame without people */
  char* error;
} chd_keypoint_clip;
int chd_openpose_load_dirs(int n, const char* const* dirs, int num_joints, int n_threads, chd_keypoint_clip* out /* n entries */);
void chd_openpose_free(int n, chd_keypoint_clip* clips);

typedef struct chd_totalcap_clip {
  int n_frames, n_joints, n_smpl_joints, n_body_coeffs, n_face_coeffs;      /* 25, 22, 30, 200 in the reference's data */
  double* root_trans;                   /* n_frames x 3 */
  double* joint3d;                      /* n_frames x n_joints x 3 */
  double* smpl_joint3d;                 /* n_frames x n_smpl_joints x 3 */
  double* smpl_joint_angles;            /* n_frames x n_smpl_joints x 3 (angle-axis) */
  double* body_coeffs;                  /* n_frames x n_body_coeffs */
  double* face_coeffs;                  /* n_frames x n_face_coeffs */
  char* error;
} chd_totalcap_clip;
int chd_totalcap_load_batch(int n, const char* const* paths, int n_threads, chd_totalcap_clip* out /* n entries */);
void chd_totalcap_free(int n, chd_totalcap_clip* clips);

#ifdef __cplusplus
}
#endif
#endif /* CHD_PREPARE_H */

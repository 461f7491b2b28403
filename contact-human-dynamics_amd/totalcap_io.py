"""Ingest of the monocular-total-capture output the kinematic optimisation starts from (``tracked_results.json``): host mirror of
the functions ``optimize_2d_3d`` calls before ``optimize_trajectory`` (src/optimize/kinematic_optimizer.py:64-74, :153) --
``load_totalcap_results``, ``normalize_root_pos``, ``create_combined_model``, ``combined_angles_from_smpl``
(src/utils/totalcap_utils.py:33-79, :134-186).  Pinned to what the reference's own functions return
(tests/golden/totalcap_golden.npz, tests/test_kinopt_driver.py)."""
import json
from dataclasses import dataclass

import numpy as np

BODY_25_ROOT_IDX = 8          # totalcap_utils.py:13
SMPL_ROOT_IDX = 0             # :16
SMPL_SPINE_JOINTS = (3, 6, 9)     # :18: the three SMPL joints appended to body-25 in the combined model
# character_info_utils.py:224-253: combined-skeleton joint -> SMPL joint whose angles initialise it (-1: none, zero rotation)
COMBINED_SKEL_TO_SMPL = np.array([0, 1, 4, 7, -1, -1, 10, 2, 5, 8, -1, -1, 11, 3, 6, 9, 12, 15, -1, -1, -1, -1, 16, 18, 20, 17, 19, 21])


@dataclass
class TotalCapResults:
    """totalcap_utils.TotalCapResults (:23-31)."""
    root_trans: np.ndarray            # F x 3
    joint3d: np.ndarray               # F x 25 x 3 (OpenPose body-25 joints)
    smpl_joint3d: np.ndarray          # F x 22 x 3
    smpl_joint_angles: np.ndarray     # F x 22 x 3, angle-axis, radians
    body_coeffs: np.ndarray           # F x 30
    face_coeffs: np.ndarray           # F x 200


def _xyz(d):
    return [d['x'], d['y'], d['z']]


def load_totalcap_results(file_path):
    """:33-79.  One JSON object: totalcapResults = [ {trans, joints[{pos}], SMPLJoints[{pos, rot}], bodyCoeffs, faceCoeffs}, ... ]."""
    with open(file_path, 'r') as fh:
        frames = json.load(fh)['totalcapResults']
    return TotalCapResults(
        root_trans=np.array([_xyz(fr['trans']) for fr in frames], dtype=np.float64),
        joint3d=np.array([[_xyz(j['pos']) for j in fr['joints']] for fr in frames], dtype=np.float64),
        smpl_joint3d=np.array([[_xyz(j['pos']) for j in fr['SMPLJoints']] for fr in frames], dtype=np.float64),
        smpl_joint_angles=np.array([[_xyz(j['rot']) for j in fr['SMPLJoints']] for fr in frames], dtype=np.float64),
        body_coeffs=np.array([fr['bodyCoeffs'] for fr in frames], dtype=np.float64),
        face_coeffs=np.array([fr['faceCoeffs'] for fr in frames], dtype=np.float64))


def normalize_root_pos(root_trans, joint3d, root_idx=BODY_25_ROOT_IDX):
    """:177-186: the regressed root joint is not at the origin; move its offset into the global translation."""
    return root_trans + joint3d[:, root_idx], joint3d - joint3d[:, root_idx][:, None]


def create_combined_model(body25_joint3d, smpl_joint3d):
    """:165-174: body-25 plus the three SMPL spine joints (both root relative)."""
    return np.concatenate([body25_joint3d, smpl_joint3d[:, list(SMPL_SPINE_JOINTS)]], axis=1)


def combined_angles_from_smpl(smpl_angles):
    """:134-147: F x 22 x 3 SMPL angles -> F x 28 x 3 in combined-skeleton order (zero where no SMPL joint corresponds)."""
    out = np.zeros((smpl_angles.shape[0], len(COMBINED_SKEL_TO_SMPL), 3))
    has = COMBINED_SKEL_TO_SMPL >= 0
    out[:, has] = smpl_angles[:, COMBINED_SKEL_TO_SMPL[has]]
    return out

"""ctypes layer over ``libchd_prepare.so`` (include/chd_prepare.h): the batched per-frame numerics of ``prepare_input`` as one HIP kernel launch
(``prep_frames``) and the native BVH reader (``load_bvh_batch``).  No CPU path for the kernel: without the library / a HIP device the call raises."""
import ctypes as C
import os
import subprocess

import numpy as np

from . import skeleton_io as sk

_HERE = os.path.dirname(os.path.abspath(__file__))
_CSRC = os.path.join(_HERE, 'csrc')
LIB_PATH = os.path.join(_CSRC, 'libchd_prepare.so')
SOURCES = ['chd_prepare.hip', 'chd_prepare_kernels.hpp', 'chd_bvh.hpp', 'chd_json.hpp']
EXPORTS = ['chd_prep_version', 'chd_prep_frames', 'chd_prep_last_kernel_ms', 'chd_prep_last_error', 'chd_bvh_load_batch', 'chd_bvh_free',
           'chd_openpose_load_dirs', 'chd_openpose_free', 'chd_totalcap_load_batch', 'chd_totalcap_free']
ABI_VERSION = 2
MAX_JOINTS, MAX_SEGMENTS, MAX_SEGMENT_JOINTS, OUT_STRIDE = 64, 32, 256, 28
PD = C.POINTER(C.c_double)


class ChdPrepSkeleton(C.Structure):
    _fields_ = [('n_joints', C.c_int), ('n_joints_body', C.c_int), ('parents', C.c_int * MAX_JOINTS), ('n_segments', C.c_int),
                ('seg_first', C.c_int * (MAX_SEGMENTS + 1)), ('seg_joint', C.c_int * MAX_SEGMENT_JOINTS), ('seg_mass_fraction', C.c_double * MAX_SEGMENTS),
                ('mass', C.c_double), ('hip_inds', C.c_int * 2), ('toe_inds', C.c_int * 2), ('heel_inds', C.c_int * 2)]


class ChdBvhClip(C.Structure):
    _fields_ = [('n_frames', C.c_int), ('n_joints', C.c_int), ('channels', C.c_int), ('frame_time', C.c_double), ('order', C.c_char * 4),
                ('names', C.c_void_p), ('parents', C.POINTER(C.c_int)), ('offsets', PD), ('positions', PD), ('rotations', PD), ('error', C.c_void_p)]


def build_library(force=False, verbose=False):
    """hipcc --offload-arch=gfx950 (cross-compiles without a GPU); in-tree so that the .so travels with the repo."""
    srcs = [os.path.join(_CSRC, s) for s in SOURCES] + [os.path.join(_HERE, '..', 'include', 'chd_prepare.h')]
    if not force and os.path.exists(LIB_PATH) and all(os.path.getmtime(s) <= os.path.getmtime(LIB_PATH) for s in srcs):
        return LIB_PATH
    cmd = ['hipcc', '--offload-arch=gfx950', '-O3', '-ffp-contract=off', '-std=c++17', '-fPIC', '-shared', os.path.join(_CSRC, 'chd_prepare.hip'), '-o', LIB_PATH]
    if verbose:
        print(' '.join(cmd))
    subprocess.check_call(cmd)
    return LIB_PATH


_LIB = None


def load_library():
    global _LIB
    if _LIB is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError('libchd_prepare.so is not built (run __graft_entry__.build())')
        L = C.CDLL(LIB_PATH)
        L.chd_prep_version.restype = C.c_int
        if L.chd_prep_version() != ABI_VERSION:
            raise RuntimeError('libchd_prepare.so has ABI version %d, expected %d' % (L.chd_prep_version(), ABI_VERSION))
        L.chd_prep_frames.argtypes = [C.POINTER(ChdPrepSkeleton), C.c_int, C.c_longlong, PD, PD, PD]
        L.chd_prep_last_error.restype = C.c_char_p
        L.chd_prep_last_kernel_ms.restype = C.c_double
        L.chd_bvh_load_batch.argtypes = [C.c_int, C.POINTER(C.c_char_p), C.c_int, C.POINTER(ChdBvhClip)]
        L.chd_bvh_free.argtypes = [C.c_int, C.POINTER(ChdBvhClip)]
        L.chd_openpose_load_dirs.argtypes = [C.c_int, C.POINTER(C.c_char_p), C.c_int, C.c_int, C.POINTER(ChdKeypointClip)]
        L.chd_openpose_free.argtypes = [C.c_int, C.POINTER(ChdKeypointClip)]
        L.chd_totalcap_load_batch.argtypes = [C.c_int, C.POINTER(C.c_char_p), C.c_int, C.POINTER(ChdTotalcapClip)]
        L.chd_totalcap_free.argtypes = [C.c_int, C.POINTER(ChdTotalcapClip)]
        _LIB = L
    return _LIB


def skeleton_of(character, parents_with_heels, n_joints_body):
    """The character's tables as the kernel reads them.  `parents_with_heels`: hierarchy of the animation pass 2 sees (heels appended when the character has none)."""
    J1 = len(parents_with_heels)
    if J1 > MAX_JOINTS:
        raise ValueError('skeletons of more than %d joints are not supported' % MAX_JOINTS)
    s = ChdPrepSkeleton()
    s.n_joints = J1; s.n_joints_body = int(n_joints_body)
    for j, a in enumerate(parents_with_heels):
        s.parents[j] = int(a)
    segs = list(character.seg_to_joints.items())
    if len(segs) > MAX_SEGMENTS or sum(len(j) for _, j in segs) > MAX_SEGMENT_JOINTS:
        raise ValueError('segment tables too large')
    s.n_segments = len(segs)
    k = 0
    for i, (key, joints) in enumerate(segs):
        s.seg_first[i] = k
        for j in joints:
            s.seg_joint[k] = int(j); k += 1
        s.seg_mass_fraction[i] = character.seg_to_mass_perc[key] * 0.01
    s.seg_first[len(segs)] = k
    s.mass = float(character.mass)
    lh, rh = (J1 - 2, J1 - 1) if character.heel_inds is None else character.heel_inds
    for i in range(2):
        s.hip_inds[i] = int(character.hip_inds[i]); s.toe_inds[i] = int(character.toe_inds[i])
    s.heel_inds[0], s.heel_inds[1] = int(lh), int(rh)
    return s


def prep_frames(skel, rot, pos, device=0):
    """rot (N, J, 4), pos (N, J, 3) -> (N, 28) on the GPU (one launch); raises without the library or a HIP device."""
    L = load_library()
    rot = np.ascontiguousarray(rot, dtype=np.float64); pos = np.ascontiguousarray(pos, dtype=np.float64)
    N = rot.shape[0]
    assert rot.shape == (N, skel.n_joints, 4) and pos.shape == (N, skel.n_joints, 3)
    out = np.empty((N, OUT_STRIDE))
    rc = L.chd_prep_frames(C.byref(skel), int(device), N, rot.ctypes.data_as(PD), pos.ctypes.data_as(PD), out.ctypes.data_as(PD))
    if rc != 0:
        raise RuntimeError('chd_prep_frames: ' + (L.chd_prep_last_error() or b'').decode())
    return out


def last_kernel_ms():
    return float(load_library().chd_prep_last_kernel_ms())


class ChdKeypointClip(C.Structure):
    _fields_ = [('n_frames', C.c_int), ('data', PD), ('error', C.c_void_p)]


class ChdTotalcapClip(C.Structure):
    _fields_ = [('n_frames', C.c_int), ('n_joints', C.c_int), ('n_smpl_joints', C.c_int), ('n_body_coeffs', C.c_int), ('n_face_coeffs', C.c_int),
                ('root_trans', PD), ('joint3d', PD), ('smpl_joint3d', PD), ('smpl_joint_angles', PD), ('body_coeffs', PD), ('face_coeffs', PD), ('error', C.c_void_p)]


def _arr(ptr, shape):
    n = int(np.prod(shape))
    return np.ctypeslib.as_array(ptr, shape=(max(n, 1),))[:n].reshape(shape).copy()


def load_keypoint_dirs(dirs, num_joints=25, n_threads=0):
    """`contact_net.load_keypoint_dir` for a list of OpenPose result directories, read natively on the host's cores -> [F x num_joints x 3].  A directory with a file that
    is not what OpenPose writes raises ValueError naming the file."""
    L = load_library()
    n = len(dirs)
    arr = (C.c_char_p * n)(*[os.fsencode(p) for p in dirs])
    clips = (ChdKeypointClip * n)()
    rc = L.chd_openpose_load_dirs(n, arr, int(num_joints), int(n_threads), clips)
    try:
        if rc < 0:
            raise ValueError('chd_openpose_load_dirs: bad arguments')
        out = []
        for c in clips:
            if c.error:
                raise ValueError(C.string_at(c.error).decode())
            out.append(_arr(c.data, (c.n_frames, num_joints, 3)))
        return out
    finally:
        L.chd_openpose_free(n, clips)


def load_totalcap_batch(paths, n_threads=0):
    """`totalcap_io.load_totalcap_results` for a list of tracked_results.json files, read natively on the host's cores -> [TotalCapResults]."""
    from .totalcap_io import TotalCapResults
    L = load_library()
    n = len(paths)
    arr = (C.c_char_p * n)(*[os.fsencode(p) for p in paths])
    clips = (ChdTotalcapClip * n)()
    rc = L.chd_totalcap_load_batch(n, arr, int(n_threads), clips)
    try:
        if rc < 0:
            raise ValueError('chd_totalcap_load_batch: bad arguments')
        out = []
        for c in clips:
            if c.error:
                raise ValueError(C.string_at(c.error).decode())
            F = c.n_frames
            out.append(TotalCapResults(root_trans=_arr(c.root_trans, (F, 3)), joint3d=_arr(c.joint3d, (F, c.n_joints, 3)), smpl_joint3d=_arr(c.smpl_joint3d, (F, c.n_smpl_joints, 3)),
                                       smpl_joint_angles=_arr(c.smpl_joint_angles, (F, c.n_smpl_joints, 3)), body_coeffs=_arr(c.body_coeffs, (F, c.n_body_coeffs)),
                                       face_coeffs=_arr(c.face_coeffs, (F, c.n_face_coeffs))))
        return out
    finally:
        L.chd_totalcap_free(n, clips)


def load_bvh_batch(paths, n_threads=0):
    """`skeleton_io.load_bvh` for a list of files, parsed natively on the host's cores -> [(Motion, names, frame time)].  A file that cannot be read raises
    ValueError naming it (as the Python reader does)."""
    L = load_library()
    n = len(paths)
    arr = (C.c_char_p * n)(*[os.fsencode(p) for p in paths])
    clips = (ChdBvhClip * n)()
    rc = L.chd_bvh_load_batch(n, arr, int(n_threads), clips)
    try:
        if rc < 0:
            raise ValueError('chd_bvh_load_batch: bad arguments')
        out = []
        for c in clips:
            if c.error:
                raise ValueError(C.string_at(c.error).decode())
            F, J = c.n_frames, c.n_joints
            names = C.string_at(c.names).decode().split('\n')
            parents = np.ctypeslib.as_array(c.parents, shape=(J,)).astype(int)
            offsets = np.ctypeslib.as_array(c.offsets, shape=(J, 3)).copy()
            positions = np.ctypeslib.as_array(c.positions, shape=(max(F, 1), J, 3))[:F].copy()
            rotations = np.ctypeslib.as_array(c.rotations, shape=(max(F, 1), J, 4))[:F].copy()
            out.append((sk.Motion(rotations, positions, np.tile(np.array([1.0, 0.0, 0.0, 0.0]), (J, 1)), offsets, parents), names, float(c.frame_time)))
        return out
    finally:
        L.chd_bvh_free(n, clips)

"""ctypes mirror of include/chd_ik.h (IK back-projection step)."""
import ctypes as C

import numpy as np

PD = C.POINTER(C.c_double)
PI = C.POINTER(C.c_int)


class ChdIkConfig(C.Structure):
    _fields_ = [('iterations', C.c_int), ('translate', C.c_int), ('damping', C.c_double), ('smoothness', C.c_double), ('gamma', C.c_double)]

    @classmethod
    def default(cls, iterations=30, translate=1, damping=7.0, smoothness=0.001, gamma=1.0):
        """The arguments of the reference's call (towr_utils.py:843)."""
        return cls(iterations, translate, damping, smoothness, gamma)


class ChdIkSeq(C.Structure):
    _fields_ = [('n_frames', C.c_int), ('n_joints', C.c_int), ('parents', PI), ('n_targets', C.c_int), ('target_joints', PI),
                ('targets', PD), ('rot_in', PD), ('pos_in', PD), ('rot_out', PD), ('pos_out', PD)]


def seqs_to_c(seqs):
    """list of dicts -> (array of ChdIkSeq, objects to keep alive, list of (rot_out, pos_out) arrays)."""
    arr = (ChdIkSeq * len(seqs))()
    keep, outs = [], []
    for i, s in enumerate(seqs):
        parents = np.ascontiguousarray(s['parents'], dtype=np.int32); tj = np.ascontiguousarray(s['target_joints'], dtype=np.int32)
        targets = np.ascontiguousarray(s['targets'], dtype=np.float64)
        rot = np.ascontiguousarray(s['rot'], dtype=np.float64); pos = np.ascontiguousarray(s['pos'], dtype=np.float64)
        F, J = rot.shape[:2]
        assert rot.shape == (F, J, 4) and pos.shape == (F, J, 3) and targets.shape == (len(tj), F, 3) and parents.shape == (J,)
        ro = np.zeros_like(rot); po = np.zeros_like(pos)
        keep += [parents, tj, targets, rot, pos, ro, po]
        arr[i] = ChdIkSeq(F, J, parents.ctypes.data_as(PI), len(tj), tj.ctypes.data_as(PI), targets.ctypes.data_as(PD),
                          rot.ctypes.data_as(PD), pos.ctypes.data_as(PD), ro.ctypes.data_as(PD), po.ctypes.data_as(PD))
        outs.append((ro, po))
    return arr, keep, outs

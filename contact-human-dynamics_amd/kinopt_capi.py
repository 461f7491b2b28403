"""ctypes mirror of include/chd_kinopt.h (least-squares solves of the kinematic optimisation)."""
import ctypes as C

import numpy as np

PD = C.POINTER(C.c_double)
PI = C.POINTER(C.c_int)
NJ = 28
NV = 87


class ChdKinConfig(C.Structure):
    _fields_ = [('max_nfev', C.c_int), ('ftol', C.c_double), ('xtol', C.c_double), ('gtol', C.c_double),
                ('lsmr_atol', C.c_double), ('lsmr_btol', C.c_double), ('lsmr_conlim', C.c_double), ('lsmr_maxiter', C.c_int),
                ('parents', C.c_int * NJ), ('reserved', C.c_int * 4)]


class ChdKinSeq(C.Structure):
    _fields_ = [('n_frames', C.c_int), ('offsets', PD), ('pose3d', PD), ('root_trans', PD), ('pose2d_n', PD), ('proj_w', PD), ('data_w', PD),
                ('contact', PI), ('floor_n', C.c_double * 3), ('floor_p', C.c_double * 3),
                ('w_proj', C.c_double), ('w_smooth_vel', C.c_double), ('w_smooth_acc', C.c_double), ('w_data', C.c_double), ('w_vel', C.c_double), ('w_floor', C.c_double),
                ('x', PD), ('cost', C.c_double), ('nfev', C.c_int), ('njev', C.c_int), ('status', C.c_int), ('lsmr_iterations', C.c_int), ('optimality', C.c_double), ('jv_fraction', C.c_double), ('jtu_fraction', C.c_double)]


def problems_to_c(problems):
    """list of dicts (offsets (28,3), pose3d (F,28,3), root_trans (F,3), pose2d_n (F,28,2), proj_w (F,28), data_w (F,28), contact (F,28),
    floor_n, floor_p, weights (6,), x0 (F*87,)) -> (array of ChdKinSeq, objects to keep alive, list of x arrays written by the solve)."""
    arr = (ChdKinSeq * len(problems))()
    keep, xs = [], []
    for i, p in enumerate(problems):
        F = int(np.asarray(p['pose3d']).shape[0])
        d = {k: np.ascontiguousarray(p[k], dtype=np.float64) for k in ('offsets', 'pose3d', 'root_trans', 'pose2d_n', 'proj_w', 'data_w')}
        assert d['offsets'].shape == (NJ, 3) and d['pose3d'].shape == (F, NJ, 3) and d['root_trans'].shape == (F, 3)
        assert d['pose2d_n'].shape[:2] == (F, NJ) and d['proj_w'].shape == (F, NJ) and d['data_w'].shape == (F, NJ)
        d['pose2d_n'] = np.ascontiguousarray(d['pose2d_n'][:, :, :2])
        ct = np.ascontiguousarray(np.asarray(p['contact']) == 1, dtype=np.int32)
        assert ct.shape == (F, NJ)
        x = np.array(p['x0'], dtype=np.float64).reshape(F * NV).copy()
        w = [float(v) for v in p['weights']]
        s = ChdKinSeq()
        s.n_frames = F
        for k, v in d.items():
            setattr(s, k, v.ctypes.data_as(PD))
        s.contact = ct.ctypes.data_as(PI)
        s.floor_n = (C.c_double * 3)(*[float(v) for v in p['floor_n']]); s.floor_p = (C.c_double * 3)(*[float(v) for v in p['floor_p']])
        s.w_proj, s.w_smooth_vel, s.w_smooth_acc, s.w_data, s.w_vel, s.w_floor = w
        s.x = x.ctypes.data_as(PD)
        arr[i] = s
        keep += list(d.values()) + [ct, x]
        xs.append(x)
    return arr, keep, xs


def results_of(arr, xs):
    return [dict(x=xs[i], cost=arr[i].cost, nfev=arr[i].nfev, njev=arr[i].njev, status=arr[i].status, lsmr_iterations=arr[i].lsmr_iterations,
                 optimality=arr[i].optimality, jv_fraction=arr[i].jv_fraction, jtu_fraction=arr[i].jtu_fraction) for i in range(len(xs))]

"""ctypes driver for the CPU oracle (oracle/liboracle.so) — TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

PD = C.POINTER(C.c_double)


class OrcSeqIn(C.Structure):
    _fields_ = [('F', C.c_int), ('dt', C.c_double), ('hip_l', PD), ('hip_r', PD),
                ('leg_len', C.c_double), ('heel_len', C.c_double), ('heel_dist', C.c_double), ('mass', C.c_double),
                ('inertia', PD), ('com', PD), ('euler', PD), ('ltoe', PD), ('lheel', PD), ('rtoe', PD), ('rheel', PD),
                ('normal', C.c_double * 3), ('point', C.c_double * 3), ('start_contact', C.c_int * 4),
                ('n_phases', C.c_int * 4), ('durations', PD * 4)]


class OrcConfig(C.Structure):
    _fields_ = [('w_com_lin', C.c_double), ('w_com_ang', C.c_double), ('w_ee', C.c_double), ('w_smooth', C.c_double),
                ('w_dur', C.c_double), ('max_iter', C.c_int * 6), ('tol', C.c_double)]


def build(force=False):
    so = os.path.join(_HERE, 'liboracle.so')
    srcs = [os.path.join(_HERE, f) for f in ('oracle_capi.cpp', 'nlp_model.hpp', 'ipm_solver.hpp')]
    if force or not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs):
        subprocess.check_call(['make', '-C', _HERE, '-s'])
    return so


def lib():
    global _LIB
    if _LIB is None:
        so = os.path.join(_HERE, 'liboracle.so')
        if not os.path.exists(so):
            build()
        L = C.CDLL(so)
        L.orc_create.restype = C.c_void_p
        L.orc_create.argtypes = [C.POINTER(OrcSeqIn), C.POINTER(OrcConfig)]
        L.orc_destroy.argtypes = [C.c_void_p]
        L.orc_set_stage.argtypes = [C.c_void_p, C.c_int]
        L.orc_n.argtypes = [C.c_void_p]; L.orc_m.argtypes = [C.c_void_p]
        L.orc_total_time.argtypes = [C.c_void_p]; L.orc_total_time.restype = C.c_double
        L.orc_get_x.argtypes = [C.c_void_p, PD]; L.orc_set_x.argtypes = [C.c_void_p, PD]
        L.orc_eval.argtypes = [C.c_void_p, PD, PD, PD, PD, PD, PD]
        L.orc_eval_lam.argtypes = [C.c_void_p, PD, PD, PD, PD, PD, PD, PD]
        L.orc_bounds.argtypes = [C.c_void_p, PD, PD]
        L.orc_row_family.argtypes = [C.c_void_p, C.POINTER(C.c_int)]
        L.orc_var_offsets.argtypes = [C.c_void_p, C.POINTER(C.c_int)]
        L.orc_sample_solution.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_int), PD, PD, PD, PD, C.POINTER(C.c_int)]
        L.orc_sample_solution.restype = C.c_int
        L.orc_solve_stage.argtypes = [C.c_void_p, C.c_int, C.c_int, PD]
        L.orc_solve_stage.restype = C.c_int
        L.orc_set_durations.argtypes = [C.c_void_p, PD]
        L.orc_n_phases.argtypes = [C.c_void_p, C.c_int]
        L.orc_set_study_mask.argtypes = [C.c_int, C.c_double]
        L.orc_set_ratio_low.argtypes = [C.c_double]
        if os.environ.get('ORC_STUDY_MASK'):          # study runs only (tests/tools): 0 = the shipped algorithm
            L.orc_set_study_mask(int(os.environ['ORC_STUDY_MASK']), float(os.environ.get('ORC_CLIP_CAP', '0')))
        _LIB = L
    return _LIB


def _p(a):
    return a.ctypes.data_as(PD)


class OracleProblem:
    """One sequence's NLP on the CPU oracle."""

    def __init__(self, seq, w_com_lin=0.4, w_com_ang=1.7, w_ee=0.3, w_smooth=0.1, w_dur=0.1,
                 max_iter=(7000, 7000, 7000, 2500, 2000, 7000), tol=1e-3):
        L = lib()
        self._keep = []

        def arr(a):
            a = np.ascontiguousarray(np.asarray(a, dtype=np.float64))
            self._keep.append(a)
            return _p(a)

        s = OrcSeqIn()
        s.F = seq.F; s.dt = seq.dt
        s.hip_l = arr(seq.hip_l); s.hip_r = arr(seq.hip_r)
        s.leg_len = seq.leg_len; s.heel_len = seq.heel_len; s.heel_dist = seq.heel_dist; s.mass = seq.mass
        s.inertia = arr(seq.inertia); s.com = arr(seq.com); s.euler = arr(seq.euler)
        s.ltoe = arr(seq.ltoe); s.lheel = arr(seq.lheel); s.rtoe = arr(seq.rtoe); s.rheel = arr(seq.rheel)
        for d in range(3):
            s.normal[d] = float(seq.normal[d]); s.point[d] = float(seq.point[d])
        for e in range(4):
            s.start_contact[e] = int(seq.start_contact[e])
            s.n_phases[e] = len(seq.durations[e])
            s.durations[e] = arr(seq.durations[e])
        c = OrcConfig()
        c.w_com_lin, c.w_com_ang, c.w_ee, c.w_smooth, c.w_dur = w_com_lin, w_com_ang, w_ee, w_smooth, w_dur
        for i in range(6):
            c.max_iter[i] = int(max_iter[i])
        c.tol = tol
        self.h = L.orc_create(C.byref(s), C.byref(c))
        if not self.h:
            raise RuntimeError('orc_create failed')
        self.F = seq.F
        self.stage = 0

    def __del__(self):
        if getattr(self, 'h', None):
            lib().orc_destroy(self.h)
            self.h = None

    def set_stage(self, stage):
        lib().orc_set_stage(self.h, stage)
        self.stage = stage

    @property
    def n(self):
        return lib().orc_n(self.h)

    @property
    def m(self):
        return lib().orc_m(self.h)

    @property
    def T(self):
        return lib().orc_total_time(self.h)

    def get_x(self):
        x = np.zeros(self.n)
        lib().orc_get_x(self.h, _p(x))
        return x

    def set_x(self, x):
        x = np.ascontiguousarray(x, dtype=np.float64)
        lib().orc_set_x(self.h, _p(x))

    def eval(self, x, jac=True, hess=False, lam=None):
        n, m = self.n, self.m
        x = np.ascontiguousarray(x, dtype=np.float64)
        f = C.c_double(0)
        g = np.zeros(n); c = np.zeros(m)
        J = np.zeros((m, n)) if jac else None
        H = np.zeros((n, n)) if hess else None
        if lam is not None:
            lam = np.ascontiguousarray(lam, dtype=np.float64)
            lib().orc_eval_lam(self.h, _p(x), _p(lam), C.byref(f), _p(g), _p(c), _p(J) if jac else None, _p(H) if hess else None)
        else:
            lib().orc_eval(self.h, _p(x), C.byref(f), _p(g), _p(c), _p(J) if jac else None, _p(H) if hess else None)
        return f.value, g, c, J, H

    def eval_state(self, stage, node_vars, durations):
        """The model of `stage` at a point given from outside (another solver's result): node variables in variable-set order and ALL phase durations of the
        four end-effectors (NLP order; a list of four arrays).  Returns dict(objective, violation, dynamics_violation, c, cl, cu)."""
        L = lib()
        d = np.ascontiguousarray(np.concatenate([np.asarray(v, dtype=np.float64) for v in durations]))
        assert [len(v) for v in durations] == [L.orc_n_phases(self.h, e) for e in range(4)]
        L.orc_set_durations(self.h, _p(d))
        self.set_stage(stage)
        x = self.get_x()
        nn = len(node_vars)
        assert nn <= x.size
        x[:nn] = node_vars
        if x.size > nn:                       # the stage optimises durations: they follow the node variables, all but each end-effector's last phase
            x[nn:] = np.concatenate([np.asarray(v, dtype=np.float64)[:-1] for v in durations])
        f, _, c, _, _ = self.eval(x, jac=False)
        cl, cu = self.bounds_at(x)
        viol = np.maximum(np.maximum(cl - c, c - cu), 0.0)
        dyn = viol[self.row_family() == 16]                      # FAM_DYNAMIC (nlp_model.hpp)
        return dict(objective=f, violation=float(viol.max()) if viol.size else 0.0, dynamics_violation=float(dyn.max()) if dyn.size else 0.0, c=c, cl=cl, cu=cu, x=x)

    def bounds_at(self, x):
        """row bounds without moving the point (orc_bounds evaluates at the problem's current x)"""
        self.set_x(x)
        return self.bounds()

    def bounds(self):
        cl = np.zeros(self.m); cu = np.zeros(self.m)
        lib().orc_bounds(self.h, _p(cl), _p(cu))
        return cl, cu

    def row_family(self):
        fam = np.zeros(self.m, dtype=np.int32)
        lib().orc_row_family(self.h, fam.ctypes.data_as(C.POINTER(C.c_int)))
        return fam

    def var_offsets(self):
        off = np.zeros(11, dtype=np.int32)
        lib().orc_var_offsets(self.h, off.ctypes.data_as(C.POINTER(C.c_int)))
        return off

    def sample_solution(self):
        cap = self.F + 8
        hdr = C.c_int(0)
        bl = np.zeros((cap, 3)); ba = np.zeros((cap, 3))
        ep = np.zeros((4, cap, 3)); ef = np.zeros((4, cap, 3)); ct = np.zeros((4, cap), dtype=np.int32)
        ns = lib().orc_sample_solution(self.h, cap, C.byref(hdr), _p(bl), _p(ba), _p(ep), _p(ef),
                                       ct.ctypes.data_as(C.POINTER(C.c_int)))
        return dict(num_frames=hdr.value, n_samples=ns, base_lin=bl[:ns], base_ang_deg=ba[:ns],
                    ee_pos=ep[:, :ns], ee_force=ef[:, :ns], contact=ct[:, :ns])

    def solve_stage(self, stage, max_iter=0):
        st = np.zeros(8)
        status = lib().orc_solve_stage(self.h, stage, max_iter, _p(st))
        self.stage = stage
        return status, dict(iters=int(st[0]), kkt_error=st[1], constr_viol=st[2], objective=st[3], mu=st[4],
                            n_factor=int(st[5]), N=int(st[6]), bandwidth=int(st[7]))

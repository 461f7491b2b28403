"""ctypes driver of the host emulation build (tests/host_emu/libchd_emu.so) — test infrastructure."""
import ctypes as C
import os
import subprocess
import sys

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_ROOT = os.path.dirname(os.path.dirname(_HERE))
if _ROOT not in sys.path:
    sys.path.insert(0, _ROOT)
import chd_amd  # noqa: E402
from chd_amd.phys_capi import ChdConfig, ChdSeqIn, PD, default_config, seq_to_c  # noqa: E402

_LIB = None


# CHD_EMU_VARIANT=noinertia builds / loads the emulation with -DCHD_INERTIA_RETRY=0 (the round-1 GPU build's behaviour, chd_kernels.hpp); one variant per process
_VARIANT = os.environ.get('CHD_EMU_VARIANT', '')


def build(force=False):
    so = os.path.join(_HERE, 'libchd_emu%s.so' % ('_' + _VARIANT if _VARIANT else ''))
    csrc = os.path.join(_ROOT, 'contact-human-dynamics_amd', 'csrc')
    srcs = [os.path.join(_HERE, 'emu.cpp')] + [os.path.join(csrc, f) for f in ('chd_kernels.hpp', 'chd_model.hpp', 'chd_device.hpp')]
    if force or not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs):
        flags = ['-DCHD_INERTIA_RETRY=0'] if _VARIANT == 'noinertia' else []
        subprocess.check_call(['g++', '-O2', '-std=c++17', '-fPIC', '-shared', '-Wno-unused-variable'] + flags + ['-o', so, srcs[0]])
    return so


def lib():
    global _LIB
    if _LIB is None:
        L = C.CDLL(build())
        L.emu_create.restype = C.c_void_p
        L.emu_create.argtypes = [C.POINTER(ChdSeqIn), C.POINTER(ChdConfig)]
        L.emu_destroy.argtypes = [C.c_void_p]
        L.emu_sizes.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_int)]
        L.emu_eval.argtypes = [C.c_void_p, C.c_int, PD, PD, PD, PD, PD, PD, PD]
        L.emu_eval_lam.argtypes = [C.c_void_p, C.c_int, PD, PD, PD, PD, PD, PD, PD, PD]
        L.emu_linsolve.argtypes = [C.c_void_p, C.c_int, C.c_double, C.c_double, PD, PD, C.c_int]
        L.emu_solve.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int]
        L.emu_rebuild_fallback.argtypes = [C.c_void_p]
        L.emu_get_state.argtypes = [C.c_void_p, C.c_int, PD, PD]
        L.emu_get_out.argtypes = [C.c_void_p, PD, C.POINTER(C.c_int)]
        _LIB = L
    return _LIB


def _p(a):
    return a.ctypes.data_as(PD) if a is not None else None


class EmuProblem:
    def __init__(self, seq, cfg=None):
        self.keep = []
        self.cfg = cfg or default_config()
        self.cin = seq_to_c(seq, self.keep)
        self.h = lib().emu_create(C.byref(self.cin), C.byref(self.cfg))
        if not self.h:
            raise RuntimeError('emu_create failed')
        self.F = seq.F

    def __del__(self):
        if getattr(self, 'h', None):
            lib().emu_destroy(self.h); self.h = None

    def sizes(self, stage):
        o = (C.c_int * 8)()
        lib().emu_sizes(self.h, stage, o)
        return dict(n=o[0], m=o[1], Nb=o[2], bc=o[3], w=o[4], nnz_jac=o[5], valid=o[6], cap=o[7])

    def eval(self, stage, x=None, jac=True, hess=True, lam=None):
        sz = self.sizes(stage); n, m = sz['n'], sz['m']
        xo = np.zeros(n); g = np.zeros(n); c = np.zeros(m); f = C.c_double(0)
        J = np.zeros((m, n)) if jac else None
        H = np.zeros((n, n)) if hess else None
        xx = np.ascontiguousarray(x, dtype=np.float64) if x is not None else None
        ll = np.ascontiguousarray(lam, dtype=np.float64) if lam is not None else None
        err = lib().emu_eval_lam(self.h, stage, _p(xx), _p(ll), _p(xo), C.byref(f), _p(g), _p(c), _p(J), _p(H))
        return dict(x=xo, f=f.value, g=g, c=c, J=J, H=H, err=err)

    def linsolve(self, stage, b, dw=1e-4, dval=1e-3, refine=2):
        x = np.zeros_like(b)
        bad = lib().emu_linsolve(self.h, stage, dw, dval, _p(np.ascontiguousarray(b)), _p(x), refine)
        return x, bad

    def solve(self, first, last, lds=0):
        lib().emu_solve(self.h, first, last, lds)

    def rebuild_fallback(self):
        return lib().emu_rebuild_fallback(self.h)

    def state(self, snap):
        """(node variables, phase durations [4 lists]) behind snapshot `snap`: what chd_debug_get_state returns on the device"""
        n = self.sizes(0)['n']
        xv = np.zeros(n + 8); ph = np.zeros(4 * 64)
        nn = lib().emu_get_state(self.h, snap, _p(xv), _p(ph))
        return xv[:nn].copy(), ph

    def results(self):
        cap = self.sizes(0)['cap']
        od = np.zeros(6 * 8 + 3 * 10 * cap * 3 + 24); oi = np.zeros(8 + 3 * 4 * cap, dtype=np.int32)
        lib().emu_get_out(self.h, _p(od), oi.ctypes.data_as(C.POINTER(C.c_int)))
        stats = od[:48].reshape(6, 8)
        snaps = []
        blocks = od[48:48 + 3 * 10 * cap * 3].reshape(3, 10, cap, 3)
        for s in range(3):
            ns = oi[2 * s]
            snaps.append(dict(n_samples=int(ns), num_frames=int(oi[2 * s + 1]), base_lin=blocks[s, 0, :ns], base_ang_deg=blocks[s, 1, :ns],
                              ee_pos=blocks[s, 2:6, :ns], ee_force=blocks[s, 6:10, :ns],
                              contact=oi[8:].reshape(3, 4, cap)[s, :, :ns]))
        return stats, snaps

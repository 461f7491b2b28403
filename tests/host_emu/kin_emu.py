"""ctypes wrapper of the CPU emulation of the kinematic-optimisation kernel source (tests/host_emu/kin_emu.cpp)."""
import ctypes as C
import os
import subprocess

import numpy as np

import chd_amd  # noqa: F401
from chd_amd.kinopt_capi import ChdKinConfig, PD, problems_to_c, results_of

HERE = os.path.dirname(os.path.abspath(__file__))
SO = os.path.join(HERE, 'libkin_emu.so')
_lib = None


def build(force=False):
    src = os.path.join(HERE, 'kin_emu.cpp')
    deps = [src] + [os.path.join(HERE, '..', '..', 'contact-human-dynamics_amd', 'csrc', f) for f in ('chd_kinopt_kernels.hpp', 'chd_kinopt_host.hpp')]
    deps.append(os.path.join(HERE, '..', '..', 'include', 'chd_kinopt.h'))
    if force or not os.path.exists(SO) or any(os.path.getmtime(d) > os.path.getmtime(SO) for d in deps):
        subprocess.check_call(['g++', '-O2', '-std=c++17', '-fPIC', '-shared', '-DCHD_HOST_EMU', src, '-o', SO])


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(SO)
        _lib.kin_emu_last_error.restype = C.c_char_p
    return _lib


def default_config(**kw):
    cfg = ChdKinConfig()
    lib().kin_emu_config_default(C.byref(cfg))
    for k, v in kw.items():
        setattr(cfg, k, v)
    return cfg


def cluster_size(cfg, n_frames):
    """workgroups the library would give a clip of `n_frames` frames under `cfg`"""
    return int(lib().kin_emu_cluster_size(C.byref(cfg), int(n_frames)))


def solve(problems, cfg=None):
    cfg = cfg or default_config()
    arr, keep, xs = problems_to_c(problems)
    if lib().kin_emu_solve_batch(C.byref(cfg), len(problems), arr) != 0:
        raise RuntimeError(lib().kin_emu_last_error().decode())
    return results_of(arr, xs)


def probe(problem, mode, vec=None, cfg=None, aux=0.0):
    """mode 0 residual, 1 J v, 2 J^T u, 3 LSMR (aux = damp; cfg.lsmr_maxiter bounds it) at the problem's start point."""
    cfg = cfg or default_config()
    arr, keep, xs = problems_to_c([problem])
    F = arr[0].n_frames
    n, m = 87 * F, 507 * F - 423
    out = np.zeros(m if mode in (0, 1) else n)
    a = C.c_double(aux)
    v = np.ascontiguousarray(vec, dtype=np.float64) if vec is not None else np.zeros(1)
    if lib().kin_emu_probe(C.byref(cfg), arr, mode, v.ctypes.data_as(PD), out.ctypes.data_as(PD), C.byref(a)) != 0:
        raise RuntimeError(lib().kin_emu_last_error().decode())
    return out, a.value

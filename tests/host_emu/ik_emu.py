"""ctypes wrapper of the CPU emulation of the IK kernel source (tests/host_emu/ik_emu.cpp)."""
import ctypes as C
import os
import subprocess

import chd_amd  # noqa: F401
from chd_amd.ik_capi import ChdIkConfig, seqs_to_c

HERE = os.path.dirname(os.path.abspath(__file__))
SO = os.path.join(HERE, 'libik_emu.so')
_lib = None


def build(force=False):
    src = os.path.join(HERE, 'ik_emu.cpp')
    deps = [src] + [os.path.join(HERE, '..', '..', 'contact-human-dynamics_amd', 'csrc', f) for f in ('chd_ik_kernels.hpp', 'chd_ik_host.hpp')]
    if force or not os.path.exists(SO) or any(os.path.getmtime(d) > os.path.getmtime(SO) for d in deps):
        subprocess.check_call(['g++', '-O2', '-std=c++17', '-fPIC', '-shared', '-DCHD_HOST_EMU', src, '-o', SO])


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(SO)
        _lib.ik_emu_last_error.restype = C.c_char_p
    return _lib


def solve(seqs, cfg=None):
    """seqs: list of dicts (parents, target_joints, targets (T,F,3), rot (F,J,4), pos (F,J,3)) -> list of (rot, pos)."""
    cfg = cfg or ChdIkConfig.default()
    arr, keep, outs = seqs_to_c(seqs)
    if lib().ik_emu_solve_batch(C.byref(cfg), len(seqs), arr) != 0:
        raise RuntimeError(lib().ik_emu_last_error().decode())
    return outs

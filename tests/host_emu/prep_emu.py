"""ctypes driver of the host emulation of the prepare_input kernel (tests/host_emu/libprep_emu.so) -- test infrastructure."""
import ctypes as C
import os
import subprocess
import sys

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_ROOT = os.path.dirname(os.path.dirname(_HERE))
if _ROOT not in sys.path:
    sys.path.insert(0, _ROOT)
import chd_amd  # noqa: E402,F401
from chd_amd import prepare_capi as pc  # noqa: E402


def build(force=False):
    so = os.path.join(_HERE, 'libprep_emu.so')
    srcs = [os.path.join(_HERE, 'prep_emu.cpp'), os.path.join(_ROOT, 'contact-human-dynamics_amd', 'csrc', 'chd_prepare_kernels.hpp'), os.path.join(_ROOT, 'include', 'chd_prepare.h')]
    if force or not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs):
        subprocess.check_call(['g++', '-O2', '-std=c++17', '-fPIC', '-shared', '-o', so, srcs[0]])
    return so


def frames(skel, rot, pos):
    L = C.CDLL(build())
    rot = np.ascontiguousarray(rot, dtype=np.float64); pos = np.ascontiguousarray(pos, dtype=np.float64)
    out = np.empty((rot.shape[0], pc.OUT_STRIDE))
    L.prep_emu_frames.argtypes = [C.POINTER(pc.ChdPrepSkeleton), C.c_longlong, pc.PD, pc.PD, pc.PD]
    L.prep_emu_frames(C.byref(skel), rot.shape[0], rot.ctypes.data_as(pc.PD), pos.ctypes.data_as(pc.PD), out.ctypes.data_as(pc.PD))
    return out

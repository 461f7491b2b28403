"""Python host side of the IK back-projection step: a thin ``ctypes`` layer over ``libchd_ik.so``.

It replaces the solver call inside the reference's ``apply_results`` (``src/utils/towr_utils.py:841-843``)::

    ik = JacobianInverseKinematicsCK(anim, targetmap, translate=True, iterations=30, smoothness=0.001, damping=7.0, silent=False)
    ik()

for a whole batch of videos at once: ``solve(seqs)`` takes, per video, what that call reads -- ``anim.parents``,
``anim.rotations.qs`` (F x J x 4, w x y z), ``anim.positions`` (F x J x 3) and the ``targetmap`` (joint index ->
F x 3 global positions) -- and returns the rotations / positions ``ik()`` leaves in ``anim``.  All arithmetic happens in
the HIP library; there is no CPU fallback (the constructor raises without the library or a GPU).

Checked against reference-generated vectors on the MI355X (tests/test_ik_gpu.py) and through the host emulation of the kernel source
(tests/test_ik_emu.py).  The kinematic optimisation's initialisation uses the same entry point (translate = 0, 200 iterations, 25 targets).
"""
import ctypes as C
import os
import subprocess

import numpy as np

from .ik_capi import ChdIkConfig, seqs_to_c

_HERE = os.path.dirname(os.path.abspath(__file__))
_CSRC = os.path.join(_HERE, 'csrc')
LIB_PATH = os.path.join(_CSRC, 'libchd_ik.so')
SOURCES = ['chd_ik.hip', 'chd_ik_kernels.hpp', 'chd_ik_host.hpp']
EXPORTS = ['chd_ik_version', 'chd_ik_config_default', 'chd_ik_solve_batch', 'chd_ik_last_error', 'chd_ik_last_kernel_ms', 'chd_ik_last_frames']


def build_library(force=False, verbose=False):
    """hipcc --offload-arch=gfx950 (cross-compiles without a GPU); in-tree so that the .so travels with the repo."""
    srcs = [os.path.join(_CSRC, s) for s in SOURCES] + [os.path.join(_HERE, '..', 'include', 'chd_ik.h')]
    if not force and os.path.exists(LIB_PATH) and all(os.path.getmtime(s) <= os.path.getmtime(LIB_PATH) for s in srcs):
        return LIB_PATH
    cmd = ['hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-shared', os.path.join(_CSRC, 'chd_ik.hip'), '-o', LIB_PATH]
    if verbose:
        print(' '.join(cmd))
    subprocess.check_call(cmd)
    return LIB_PATH


def load_library():
    if not os.path.exists(LIB_PATH):
        raise RuntimeError('libchd_ik.so is not built (run __graft_entry__.build()); the IK step has no CPU path')
    lib = C.CDLL(LIB_PATH)
    lib.chd_ik_version.restype = C.c_char_p
    lib.chd_ik_last_error.restype = C.c_char_p
    lib.chd_ik_last_kernel_ms.restype = C.c_double
    lib.chd_ik_last_frames.restype = C.c_longlong
    return lib


class IkBackProject:
    def __init__(self, device=0, config=None):
        self.lib = load_library()
        self.device = device
        self.cfg = config or ChdIkConfig.default()

    def solve(self, seqs):
        """seqs: list of dicts with keys parents (J,), target_joints (T,), targets (T, F, 3), rot (F, J, 4), pos (F, J, 3).
        Returns a list of (rot, pos) arrays."""
        arr, keep, outs = seqs_to_c(seqs)
        if self.lib.chd_ik_solve_batch(C.byref(self.cfg), self.device, len(seqs), arr) != 0:
            raise RuntimeError('chd_ik_solve_batch: ' + self.lib.chd_ik_last_error().decode())
        return outs

    def last_kernel_ms(self):
        """Device time of the last `solve` (HIP events around its launches) and the frames per launch."""
        return float(self.lib.chd_ik_last_kernel_ms()), int(self.lib.chd_ik_last_frames())

    @staticmethod
    def targetmap_to_arrays(targetmap):
        """dict joint -> (F, 3), in its iteration order (the order the reference's Jacobian uses)."""
        joints = np.array(list(targetmap.keys()), dtype=np.int32)
        return joints, np.stack([np.asarray(v, dtype=np.float64) for v in targetmap.values()], axis=0)

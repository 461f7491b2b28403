"""Python host side of the physics stage: a thin ``ctypes`` layer over ``libchd_phys.so``.

This is the in-process replacement for the child process the reference starts per video
(``subprocess.run(['./phys_optim', '--in_dir', ..., '--nframes', ..., '--out_dir', ...,
'--w_com_lin', ...])``, ``scripts/run_phys_mocap.py:159-174``).  All arithmetic happens in the
HIP library; nothing here falls back to a CPU solve — if the extension or a GPU is missing
the constructor raises.
"""
import ctypes as C
import os
import subprocess

import numpy as np

from .io_formats import Solution
from .phys_capi import (ChdBatchStats, ChdCallStats, ChdConfig, ChdSeqIn, ChdSeqOut, N_SNAPSHOTS, N_STAGES, PD, default_config,
                        seq_to_c)

_HERE = os.path.dirname(os.path.abspath(__file__))
_CSRC = os.path.join(_HERE, 'csrc')
LIB_PATH = os.environ.get('CHD_PHYS_LIB') or os.path.join(_CSRC, 'libchd_phys.so')      # (override: kernel experiments with variant builds)
SOURCES = ['chd_phys.hip', 'chd_kernels.hpp', 'chd_model.hpp', 'chd_device.hpp', 'chd_io.hpp']
ABI_VERSION = 2      # CHD_PHYS_ABI_VERSION of include/chd_phys.h
SNAPSHOT_FILES = ('sol_out_no_dynamics.txt', 'sol_out_dynamics.txt', 'sol_out_durations.txt')

EXPORTS = ['chd_phys_version', 'chd_config_default', 'chd_phys_create', 'chd_phys_destroy', 'chd_phys_last_error',
           'chd_batch_upload', 'chd_batch_solve', 'chd_batch_fetch', 'chd_batch_free', 'chd_batch_get_stats',
           'chd_phys_solve_batch', 'chd_phys_get_call_stats', 'chd_phys_solve_dirs', 'chd_debug_sizes', 'chd_debug_eval', 'chd_debug_linsolve', 'chd_debug_get_state']


def build_library(force=False, verbose=False):
    """Compile the HIP library for gfx950 in-tree (hipcc cross-compiles without a GPU)."""
    srcs = [os.path.join(_CSRC, s) for s in SOURCES] + [os.path.join(os.path.dirname(_HERE), 'include', 'chd_phys.h')]
    if not force and os.path.exists(LIB_PATH) and all(os.path.getmtime(s) <= os.path.getmtime(LIB_PATH) for s in srcs):
        return LIB_PATH
    hipcc = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
    cmd = [hipcc, '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-shared', '-Wno-unused-value',
           os.path.join(_CSRC, 'chd_phys.hip'), '-o', LIB_PATH]
    if verbose:
        print(' '.join(cmd))
    subprocess.check_call(cmd)
    return LIB_PATH


CLI_PATH = os.path.join(_HERE, 'cli', 'phys_optim')


def build_cli(force=False, verbose=False):
    """The `phys_optim` executable (cli/phys_optim_main.cpp): the reference's gflags, one sequence per process, for an
    unmodified scripts/run_phys_mocap.py --towr_phys_optim_path <this cli directory>.  Links libchd_phys.so by rpath."""
    build_library()
    src = os.path.join(_HERE, 'cli', 'phys_optim_main.cpp')
    if not force and os.path.exists(CLI_PATH) and all(os.path.getmtime(s) <= os.path.getmtime(CLI_PATH) for s in (src, LIB_PATH)):
        return CLI_PATH
    cmd = ['g++', '-O2', '-std=c++17', src, '-o', CLI_PATH, '-L' + _CSRC, '-lchd_phys', '-Wl,-rpath,$ORIGIN/../csrc']
    if verbose:
        print(' '.join(cmd))
    subprocess.check_call(cmd)
    return CLI_PATH


_LIB = None


def load_library():
    """dlopen ``libchd_phys.so`` and declare the prototypes of ``include/chd_phys.h``."""
    global _LIB
    if _LIB is not None:
        return _LIB
    if not os.path.exists(LIB_PATH):
        raise RuntimeError('libchd_phys.so is not built (%s); run __graft_entry__.build() — there is no CPU fallback' % LIB_PATH)
    L = C.CDLL(LIB_PATH)
    vp = C.c_void_p
    L.chd_phys_version.restype = C.c_int
    if L.chd_phys_version() != ABI_VERSION:
        raise RuntimeError('libchd_phys.so has ABI version %d, this host layer was written against %d (include/chd_phys.h): rebuild with __graft_entry__.build()'
                           % (L.chd_phys_version(), ABI_VERSION))
    L.chd_config_default.argtypes = [C.POINTER(ChdConfig)]
    L.chd_phys_create.argtypes = [C.POINTER(ChdConfig), C.c_int, C.POINTER(vp)]
    L.chd_phys_destroy.argtypes = [vp]
    L.chd_phys_last_error.argtypes = [vp]; L.chd_phys_last_error.restype = C.c_char_p
    L.chd_batch_upload.argtypes = [vp, C.c_int, C.POINTER(ChdSeqIn), C.POINTER(vp)]
    L.chd_batch_solve.argtypes = [vp, vp]
    L.chd_batch_fetch.argtypes = [vp, vp, C.POINTER(ChdSeqOut)]
    L.chd_batch_free.argtypes = [vp, vp]
    L.chd_batch_get_stats.argtypes = [vp, vp, C.POINTER(ChdBatchStats)]
    L.chd_phys_solve_batch.argtypes = [vp, C.c_int, C.POINTER(ChdSeqIn), C.POINTER(ChdSeqOut)]
    L.chd_phys_get_call_stats.argtypes = [vp, C.POINTER(ChdCallStats)]
    L.chd_phys_solve_dirs.argtypes = [vp, C.c_int, C.POINTER(C.c_char_p), C.POINTER(C.c_char_p), C.POINTER(C.c_int), C.POINTER(C.c_int)]
    L.chd_debug_sizes.argtypes = [vp, vp, C.c_int, C.c_int] + [C.POINTER(C.c_int)] * 5
    L.chd_debug_eval.argtypes = [vp, vp, C.c_int, C.c_int, PD, PD, PD, PD, PD, PD, PD]
    L.chd_debug_get_state.argtypes = [vp, vp, C.c_int, C.c_int, PD, C.POINTER(C.c_int), PD, C.POINTER(C.c_int)]
    L.chd_debug_linsolve.argtypes = [vp, vp, C.c_int, C.c_int, C.c_double, C.c_double, C.c_int, C.c_int, PD, PD, PD]
    _LIB = L
    return L


class PhysError(RuntimeError):
    pass


class SeqResult:
    """Per-sequence output: three snapshots (:class:`io_formats.Solution`) + solver statistics."""

    def __init__(self, dt, out, bufs):
        self.snapshots = []
        for s in range(N_SNAPSHOTS):
            sn = out.snap[s]
            ns = min(sn.n_samples, sn.capacity)
            bl, ba, ep, ef, ct = bufs[s]
            self.snapshots.append(Solution(dt=dt, num_frames=sn.num_frames_header, base_lin=bl[:ns].copy(), base_ang_deg=ba[:ns].copy(),
                                           ee_pos=ep[:, :ns].copy(), ee_force=ef[:, :ns].copy(), contact=ct[:, :ns].astype(np.int64)))
        self.stage_status = [out.stage_status[i] for i in range(N_STAGES)]
        self.stage_iters = [out.stage_iters[i] for i in range(N_STAGES)]
        self.stage_stalled = [out.stage_stalled[i] for i in range(N_STAGES)]
        self.stage_factorizations = [out.stage_factorizations[i] for i in range(N_STAGES)]
        self.rejected = self.stage_status[0] == -4       # refused at set-up (inconsistent inputs): nothing was solved
        self.stage_kkt_error = [out.stage_kkt_error[i] for i in range(N_STAGES)]
        self.stage_constr_viol = [out.stage_constr_viol[i] for i in range(N_STAGES)]
        self.stage_objective = [out.stage_objective[i] for i in range(N_STAGES)]
        self.dynamics_succeed = bool(out.dynamics_succeed)
        self.durations_succeed = bool(out.durations_succeed)
        self.sizes = dict(n=out.n_vars, m=out.n_rows, kkt_dim=out.kkt_dim, halfband=out.kkt_halfband, border=out.kkt_border,
                          nnz_jac=out.nnz_jac)

    @property
    def total_iters(self):
        return sum(self.stage_iters)


def alloc_outputs(seqs):
    """Caller-owned result arrays of ``chd_seq_out`` for a list of sequences: (ctypes array, per-sequence NumPy buffers)."""
    B = len(seqs)
    outs = (ChdSeqOut * B)()
    bufs = []
    for i, s in enumerate(seqs):
        cap = s.F + 4
        per = []
        for k in range(N_SNAPSHOTS):
            bl = np.zeros((cap, 3)); ba = np.zeros((cap, 3)); ep = np.zeros((4, cap, 3)); ef = np.zeros((4, cap, 3))
            ct = np.zeros((4, cap), dtype=np.uint8)
            sn = outs[i].snap[k]
            sn.capacity = cap
            sn.base_lin = bl.ctypes.data_as(PD); sn.base_ang_deg = ba.ctypes.data_as(PD)
            sn.ee_pos = ep.ctypes.data_as(PD); sn.ee_force = ef.ctypes.data_as(PD)
            sn.contact = ct.ctypes.data_as(C.POINTER(C.c_ubyte))
            per.append((bl, ba, ep, ef, ct))
        bufs.append(per)
    return outs, bufs


class Batch:
    """Device-resident batch (``chd_batch``): upload once, solve (the timed part), fetch."""

    def __init__(self, solver, seqs):
        self.solver = solver
        self.seqs = list(seqs)
        self._keep = []
        B = len(self.seqs)
        self._in = (ChdSeqIn * B)(*[seq_to_c(s, self._keep) for s in self.seqs])
        self.h = C.c_void_p()
        solver._check(solver.L.chd_batch_upload(solver.h, B, self._in, C.byref(self.h)), 'chd_batch_upload')

    def solve(self):
        self.solver._check(self.solver.L.chd_batch_solve(self.solver.h, self.h), 'chd_batch_solve')
        st = ChdBatchStats()
        self.solver._check(self.solver.L.chd_batch_get_stats(self.solver.h, self.h, C.byref(st)), 'chd_batch_get_stats')
        return dict(kernel_ms=[st.kernel_ms[0], st.kernel_ms[1]], host_ms=st.host_ms, total_iters=st.total_iters,
                    total_factorizations=st.total_factorizations, alg_bytes=st.alg_bytes, n_fallback=st.n_fallback,
                    phase_ms=[st.phase_ms[i] for i in range(24)], max_seq_ms=st.max_seq_ms, n_stalled=st.n_stalled,
                    n_rejected=st.n_rejected, n_workgroups=st.n_workgroups)

    def fetch(self):
        B = len(self.seqs)
        outs, bufs = alloc_outputs(self.seqs)
        self.solver._check(self.solver.L.chd_batch_fetch(self.solver.h, self.h, outs), 'chd_batch_fetch')
        return [SeqResult(self.seqs[i].dt, outs[i], bufs[i]) for i in range(B)]

    def sizes(self, seq, stage):
        v = [C.c_int() for _ in range(5)]
        self.solver._check(self.solver.L.chd_debug_sizes(self.solver.h, self.h, seq, stage, *[C.byref(x) for x in v]), 'chd_debug_sizes')
        return dict(n=v[0].value, m=v[1].value, kkt_dim=v[2].value, halfband=v[3].value, border=v[4].value)

    def debug_eval(self, seq, stage, x=None, jac=True, hess=True):
        sz = self.sizes(seq, stage); n, m = sz['n'], sz['m']
        xo = np.zeros(n); g = np.zeros(n); c = np.zeros(m); f = C.c_double(0)
        J = np.zeros((m, n)) if jac else None
        H = np.zeros((n, n)) if hess else None
        xx = np.ascontiguousarray(x, dtype=np.float64) if x is not None else None
        p = lambda a: a.ctypes.data_as(PD) if a is not None else None   # noqa: E731
        rc = self.solver.L.chd_debug_eval(self.solver.h, self.h, seq, stage, p(xx), p(xo), C.byref(f), p(g), p(c), p(J), p(H))
        if rc < 0:
            raise PhysError('chd_debug_eval: ' + self.solver.last_error())
        return dict(x=xo, f=f.value, g=g, c=c, J=J, H=H, err=rc)

    def get_state(self, seq, snapshot):
        """The point behind output snapshot `snapshot` of a solved batch in the NLP's variables: (node variables, [phase durations of the 4 end-effectors])."""
        n = self.sizes(seq, 4)['n']
        xv = np.zeros(n + 8); ph = np.zeros(4 * 64)
        nn = C.c_int(0); nph = (C.c_int * 4)()
        self.solver._check(self.solver.L.chd_debug_get_state(self.solver.h, self.h, seq, snapshot, xv.ctypes.data_as(PD), C.byref(nn), ph.ctypes.data_as(PD), nph), 'chd_debug_get_state')
        durs, off = [], 0
        for e in range(4):
            durs.append(ph[off:off + nph[e]].copy()); off += nph[e]
        return xv[:nn.value].copy(), durs

    def debug_linsolve(self, seq, stage, rhs, dw=1e-4, dval=1e-3, which=0, reps=1):
        """Factor / solve self test of the KKT matrix of (seq, stage) at the initial state (`which` is ignored since round 4: one factorisation is left).
        Returns (x, info) with info = dict(bad_pivots, factor_us, solve_us, ran, phase_us)."""
        sz = self.sizes(seq, stage)
        rhs = np.ascontiguousarray(rhs, dtype=np.float64)
        assert rhs.size == sz['kkt_dim']
        x = np.zeros(sz['kkt_dim']); info = np.zeros(14)
        self.solver._check(self.solver.L.chd_debug_linsolve(self.solver.h, self.h, seq, stage, dw, dval, which, reps, rhs.ctypes.data_as(PD), x.ctypes.data_as(PD),
                                                            info.ctypes.data_as(PD)), 'chd_debug_linsolve')
        return x, dict(bad_pivots=int(info[0]), factor_us=info[1] / 100.0 / max(1, reps), solve_us=info[2] / 100.0, ran=int(info[3]),
                       phase_us={k: info[4 + k - 6] / 100.0 / max(1, reps) for k in range(6, 16)})

    def free(self):
        if self.h:
            self.solver.L.chd_batch_free(self.solver.h, self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class PhysOptim:
    """One solver handle bound to one GPU (``chd_handle``)."""

    def __init__(self, device=0, config=None, **cfg_kw):
        self.L = load_library()
        self.cfg = config if config is not None else default_config(**cfg_kw)
        self.h = C.c_void_p()
        rc = self.L.chd_phys_create(C.byref(self.cfg), int(device), C.byref(self.h))
        if rc != 0:
            raise PhysError('chd_phys_create failed (rc=%d): no usable HIP device %d — the physics stage has no CPU path' % (rc, device))

    def last_error(self):
        return (self.L.chd_phys_last_error(self.h) or b'').decode()

    def _check(self, rc, what):
        if rc != 0:
            raise PhysError('%s failed: %s' % (what, self.last_error()))

    def upload(self, seqs):
        return Batch(self, seqs)

    def solve(self, seqs):
        """Upload + solve + fetch.  Returns (results, stats)."""
        b = self.upload(seqs)
        try:
            st = b.solve()
            return b.fetch(), st
        finally:
            b.free()

    def solve_batch(self, seqs):
        """The whole call (``chd_phys_solve_batch``): set-up, upload, solve and fetch, pipelined over chunks of the batch so that the host work hides
        behind the device's.  Returns (results, call statistics)."""
        seqs = list(seqs)
        keep = []
        cin = (ChdSeqIn * len(seqs))(*[seq_to_c(s, keep) for s in seqs])
        outs, bufs = alloc_outputs(seqs)
        self._check(self.L.chd_phys_solve_batch(self.h, len(seqs), cin, outs), 'chd_phys_solve_batch')
        return [SeqResult(seqs[i].dt, outs[i], bufs[i]) for i in range(len(seqs))], self.call_stats()

    def call_stats(self):
        st = ChdCallStats()
        self._check(self.L.chd_phys_get_call_stats(self.h, C.byref(st)), 'chd_phys_get_call_stats')
        return {k: getattr(st, k) for k, _ in ChdCallStats._fields_}

    def solve_dirs(self, in_dirs, out_dirs, nframes):
        """Batched drop-in for ``./phys_optim --in_dir D_in --nframes F --out_dir D_out`` run once per directory."""
        B = len(in_dirs)
        a = (C.c_char_p * B)(*[os.fsencode(d) for d in in_dirs])
        o = (C.c_char_p * B)(*[os.fsencode(d) for d in out_dirs])
        nf = (C.c_int * B)(*[int(x) for x in nframes])
        st = (C.c_int * B)()
        self._check(self.L.chd_phys_solve_dirs(self.h, B, a, o, nf, st), 'chd_phys_solve_dirs')
        return [st[i] for i in range(B)]

    def close(self):
        if self.h:
            self.L.chd_phys_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

"""ctypes mirror of ``include/chd_phys.h`` (the C ABI of ``libchd_phys.so``)."""
import ctypes as C

import numpy as np

PD = C.POINTER(C.c_double)
N_STAGES = 6
N_SNAPSHOTS = 3


class ChdConfig(C.Structure):
    _fields_ = [('w_com_lin', C.c_double), ('w_com_ang', C.c_double), ('w_ee', C.c_double),
                ('w_smooth', C.c_double), ('w_dur', C.c_double), ('max_iter', C.c_int * N_STAGES),
                ('tol', C.c_double), ('threads_per_sequence', C.c_int), ('stall_window', C.c_int), ('max_workgroups', C.c_int),
                ('lds_kilobytes', C.c_int), ('factorisation', C.c_int), ('pipeline_chunk', C.c_int), ('damping_rule', C.c_int), ('reserved', C.c_int * 1)]


class ChdSeqIn(C.Structure):
    _fields_ = [('F', C.c_int), ('dt', C.c_double), ('hip_l', PD), ('hip_r', PD),
                ('leg_len', C.c_double), ('heel_len', C.c_double), ('heel_dist', C.c_double), ('mass', C.c_double),
                ('inertia', PD), ('com', PD), ('euler', PD), ('ltoe', PD), ('lheel', PD), ('rtoe', PD), ('rheel', PD),
                ('normal', C.c_double * 3), ('point', C.c_double * 3), ('start_contact', C.c_int * 4),
                ('n_phases', C.c_int * 4), ('durations', PD * 4)]


class ChdSnapshot(C.Structure):
    _fields_ = [('capacity', C.c_int), ('n_samples', C.c_int), ('num_frames_header', C.c_int),
                ('base_lin', PD), ('base_ang_deg', PD), ('ee_pos', PD), ('ee_force', PD),
                ('contact', C.POINTER(C.c_ubyte))]


class ChdSeqOut(C.Structure):
    _fields_ = [('snap', ChdSnapshot * N_SNAPSHOTS), ('stage_status', C.c_int * N_STAGES),
                ('stage_iters', C.c_int * N_STAGES), ('stage_stalled', C.c_int * N_STAGES), ('stage_factorizations', C.c_int * N_STAGES),
                ('stage_kkt_error', C.c_double * N_STAGES),
                ('stage_constr_viol', C.c_double * N_STAGES), ('stage_objective', C.c_double * N_STAGES),
                ('dynamics_succeed', C.c_int), ('durations_succeed', C.c_int),
                ('n_vars', C.c_int), ('n_rows', C.c_int), ('kkt_dim', C.c_int), ('kkt_halfband', C.c_int),
                ('kkt_border', C.c_int), ('nnz_jac', C.c_longlong)]


class ChdBatchStats(C.Structure):
    _fields_ = [('kernel_ms', C.c_double * 2), ('host_ms', C.c_double), ('total_iters', C.c_longlong),
                ('total_factorizations', C.c_longlong), ('alg_bytes', C.c_double), ('n_fallback', C.c_int),
                ('phase_ms', C.c_double * 24), ('max_seq_ms', C.c_double), ('n_stalled', C.c_int), ('n_rejected', C.c_int),
                ('n_workgroups', C.c_int)]


class ChdCallStats(C.Structure):
    _fields_ = [('wall_ms', C.c_double), ('prep_ms', C.c_double), ('setup_cpu_ms', C.c_double), ('setup_wall_ms', C.c_double), ('upload_ms', C.c_double),
                ('wait_for_pool_ms', C.c_double), ('finish_ms', C.c_double), ('kernel_ms', C.c_double), ('sequence_ms', C.c_double), ('max_seq_ms', C.c_double),
                ('alg_bytes', C.c_double), ('total_iters', C.c_longlong), ('total_factorizations', C.c_longlong),
                ('n_sequences', C.c_int), ('n_chunks', C.c_int), ('chunk', C.c_int), ('host_threads', C.c_int), ('n_fallback', C.c_int), ('n_stalled', C.c_int),
                ('n_rejected', C.c_int)]


def default_config(**kw):
    """Defaults of the reference's gflags / IPOPT options (phys_optim.cpp:27-31, :567-578, :640-743)."""
    c = ChdConfig()
    c.w_com_lin, c.w_com_ang, c.w_ee, c.w_smooth, c.w_dur = 0.4, 1.7, 0.3, 0.1, 0.1
    for i, v in enumerate((7000, 7000, 7000, 2500, 2000, 7000)):
        c.max_iter[i] = v
    c.tol = 1e-3
    c.threads_per_sequence = 0
    c.stall_window = 0          # no stall guard: a stage runs to its iteration cap like IPOPT would
    c.max_workgroups = 0
    c.lds_kilobytes = 0
    for k, v in kw.items():
        if k == 'max_iter':
            for i, it in enumerate(v):
                c.max_iter[i] = int(it)
        else:
            setattr(c, k, v)
    return c


def seq_to_c(seq, keep):
    """Fill a ChdSeqIn from an io_formats.SeqInput; arrays are appended to ``keep`` to stay alive."""
    def arr(a):
        a = np.ascontiguousarray(np.asarray(a, dtype=np.float64))
        keep.append(a)
        return a.ctypes.data_as(PD)

    s = ChdSeqIn()
    s.F = int(seq.F); s.dt = float(seq.dt)
    s.hip_l = arr(seq.hip_l); s.hip_r = arr(seq.hip_r)
    s.leg_len, s.heel_len, s.heel_dist, s.mass = float(seq.leg_len), float(seq.heel_len), float(seq.heel_dist), float(seq.mass)
    s.inertia = arr(seq.inertia); s.com = arr(seq.com); s.euler = arr(seq.euler)
    s.ltoe = arr(seq.ltoe); s.lheel = arr(seq.lheel); s.rtoe = arr(seq.rtoe); s.rheel = arr(seq.rheel)
    for d in range(3):
        s.normal[d] = float(seq.normal[d]); s.point[d] = float(seq.point[d])
    for e in range(4):
        s.start_contact[e] = int(seq.start_contact[e])
        s.n_phases[e] = len(seq.durations[e])
        s.durations[e] = arr(seq.durations[e])
    return s

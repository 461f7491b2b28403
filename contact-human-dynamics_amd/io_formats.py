"""File formats at the drop-in boundary of the physics stage.

Inputs  ``phys_optim_in_<char>/{skel,motion,terrain,contact}_info.txt``
        writer in the reference: ``src/utils/towr_utils.py:585-777``
        reader in the reference: ``towr_phys_optim/phys_optim.cpp:155-267``
Outputs ``phys_optim_out_<char>/{sol_out_*.txt,success_log.txt}``
        writer in the reference: ``towr_phys_optim/phys_optim.cpp:63-153``
        reader in the reference: ``src/utils/towr_utils.py:51-99``

Everything here is host-side plumbing; no arithmetic of the NLP lives here.
"""
import os
from dataclasses import dataclass, field
from typing import List

import numpy as np


@dataclass
class SeqInput:
    """Contents of the four input text files for one sequence (fp64, z-up, metres)."""
    F: int
    dt: float
    hip_l: np.ndarray        # F x 3  left hip offset in the base frame
    hip_r: np.ndarray        # F x 3
    leg_len: float           # hip -> toe maximum
    heel_len: float          # hip -> heel maximum
    heel_dist: float         # toe <-> heel distance
    mass: float
    inertia: np.ndarray      # F x 6  Ixx Iyy Izz Ixy Ixz Iyz
    com: np.ndarray          # F x 3
    euler: np.ndarray        # F x 3  extrinsic xyz Euler, radians
    ltoe: np.ndarray         # F x 3  (motion file order: L-toe, L-heel, R-toe, R-heel)
    lheel: np.ndarray
    rtoe: np.ndarray
    rheel: np.ndarray
    normal: np.ndarray       # 3
    point: np.ndarray        # 3
    start_contact: List[int]           # 4, file order L-toe, L-heel, R-toe, R-heel
    durations: List[List[float]]       # 4 lists
    contacts: np.ndarray = field(default=None, repr=False)   # optional F x 4 flags (foot_contacts.npy order)


def contact_durations(contacts, dt):
    """Phase durations from per-frame binary contact flags.

    Same semantics as ``towr_utils.find_contact_durations`` (``towr_utils.py:435-449``):
    the loop runs over frames ``0..F-2`` so the durations sum to ``(F-1)*dt`` and are
    accumulated by repeated ``+= dt``.
    """
    prev = contacts[0]
    cur = 0.0
    out = []
    for i in range(0, len(contacts) - 1):
        state = contacts[i]
        if state != prev:
            out.append(cur)
            cur = dt
        else:
            cur += dt
        prev = state
    out.append(cur)
    return out


# ----------------------------------------------------------------------------
# input files
# ----------------------------------------------------------------------------
def _row(v):
    return ' '.join(str(float(x)) for x in v)


def write_inputs(seq: SeqInput, out_dir: str):
    """Write the four ``*_info.txt`` files exactly as ``prepare_input`` lays them out
    (``towr_utils.py:585-777``): Python ``str(float)`` tokens, whitespace separated."""
    os.makedirs(out_dir, exist_ok=True)
    with open(os.path.join(out_dir, 'skel_info.txt'), 'w') as f:
        for i in range(seq.F):
            f.write(_row(seq.hip_l[i]) + '\n')
        for i in range(seq.F):
            f.write(_row(seq.hip_r[i]) + '\n')
        f.write(str(float(seq.leg_len)) + '\n')
        f.write(str(float(seq.heel_len)) + '\n')
        f.write(str(float(seq.heel_dist)) + '\n')
        f.write(str(float(seq.mass)) + '\n')
        for i in range(seq.F):
            f.write(_row(seq.inertia[i]) + '\n')
    with open(os.path.join(out_dir, 'motion_info.txt'), 'w') as f:
        f.write(str(float(seq.dt)) + '\n')
        for arr in (seq.com, seq.euler, seq.ltoe, seq.lheel, seq.rtoe, seq.rheel):
            f.write(' '.join(_row(arr[i]) for i in range(seq.F)) + '\n')
    with open(os.path.join(out_dir, 'terrain_info.txt'), 'w') as f:
        f.write(_row(seq.normal) + '\n')
        f.write(_row(seq.point))
    with open(os.path.join(out_dir, 'contact_info.txt'), 'w') as f:
        for e in range(4):
            f.write(str(int(seq.start_contact[e])) + '\n')
            f.write(str(len(seq.durations[e])) + '\n')
            f.write(' '.join(str(float(d)) for d in seq.durations[e]))
            if e < 3:
                f.write('\n')


def read_inputs(in_dir: str, nframes: int) -> SeqInput:
    """Token-stream reader with the semantics of ``phys_optim.cpp:155-267``
    (``operator>>`` on whitespace-separated tokens; ``nframes`` comes from the CLI)."""
    def toks(name):
        with open(os.path.join(in_dir, name)) as f:
            return f.read().split()

    F = nframes
    s = toks('skel_info.txt')
    need = 6 * F + 4 + 6 * F
    if len(s) < need:
        raise ValueError('skel_info.txt: expected %d tokens, found %d' % (need, len(s)))
    a = np.array(s[:need], dtype=np.float64)
    hip_l = a[:3 * F].reshape(F, 3)
    hip_r = a[3 * F:6 * F].reshape(F, 3)
    leg_len, heel_len, heel_dist, mass = a[6 * F:6 * F + 4]
    inertia = a[6 * F + 4:].reshape(F, 6)

    mt = toks('motion_info.txt')
    need = 1 + 18 * F
    if len(mt) < need:
        raise ValueError('motion_info.txt: expected %d tokens, found %d' % (need, len(mt)))
    a = np.array(mt[:need], dtype=np.float64)
    dt = float(a[0])
    blocks = a[1:].reshape(6, F, 3)

    tt = np.array(toks('terrain_info.txt')[:6], dtype=np.float64)

    ct = toks('contact_info.txt')
    pos = 0
    start, durs = [], []
    for e in range(4):
        flag = ct[pos]; pos += 1
        if flag not in ('0', '1'):
            raise ValueError('contact_info.txt: start flag must be 0/1 (operator>> into bool)')
        start.append(int(flag))
        n = int(ct[pos]); pos += 1
        durs.append([float(x) for x in ct[pos:pos + n]]); pos += n
    return SeqInput(F=F, dt=dt, hip_l=hip_l, hip_r=hip_r, leg_len=float(leg_len), heel_len=float(heel_len),
                    heel_dist=float(heel_dist), mass=float(mass), inertia=inertia,
                    com=blocks[0], euler=blocks[1], ltoe=blocks[2], lheel=blocks[3], rtoe=blocks[4], rheel=blocks[5],
                    normal=tt[:3].copy(), point=tt[3:6].copy(), start_contact=start, durations=durs)


# ----------------------------------------------------------------------------
# output files
# ----------------------------------------------------------------------------
@dataclass
class Solution:
    """One ``sol_out_*.txt`` snapshot. NLP end-effector order: L-toe, R-toe, L-heel, R-heel."""
    dt: float
    num_frames: int          # header value int((T+1e-5)/dt)+1  (phys_optim.cpp:71)
    base_lin: np.ndarray     # S x 3
    base_ang_deg: np.ndarray  # S x 3 (degrees, phys_optim.cpp:97)
    ee_pos: np.ndarray       # 4 x S x 3
    ee_force: np.ndarray     # 4 x S x 3
    contact: np.ndarray      # 4 x S  (0/1)


def _g10(x):
    # std::ofstream with precision(10), default float field == printf("%.10g")
    return '%.10g' % x


def write_solution(sol: Solution, path: str):
    """Same line layout as ``SaveSolution`` (``phys_optim.cpp:63-143``): label line,
    value line; triples separated by single spaces, no trailing space; 10 significant digits."""
    with open(path, 'w') as f:
        f.write('dt\n' + _g10(sol.dt) + '\n')
        f.write('num_frames\n%d\n' % sol.num_frames)
        f.write('num_feet\n%d\n' % sol.ee_pos.shape[0])
        f.write('base_lin\n' + ' '.join(_g10(v) for v in sol.base_lin.reshape(-1)) + '\n')
        f.write('base_ang\n' + ' '.join(_g10(v) for v in sol.base_ang_deg.reshape(-1)) + '\n')
        for i in range(sol.ee_pos.shape[0]):
            f.write('foot%d_pos\n' % i + ' '.join(_g10(v) for v in sol.ee_pos[i].reshape(-1)) + '\n')
        for i in range(sol.ee_force.shape[0]):
            f.write('foot%d_force\n' % i + ' '.join(_g10(v) for v in sol.ee_force[i].reshape(-1)) + '\n')
        for i in range(sol.contact.shape[0]):
            f.write('foot%d_contact\n' % i + ' '.join('%d' % int(v) for v in sol.contact[i]) + '\n')


def write_success_log(path: str, dynamics_ok: bool, durations_ok: bool):
    """``SaveSuccessLog`` (``phys_optim.cpp:145-153``)."""
    with open(path, 'w') as f:
        f.write('dynamics %d\n' % int(bool(dynamics_ok)))
        f.write('durations %d\n' % int(bool(durations_ok)))


def load_results(file_path: str) -> Solution:
    """Line-number-indexed parser with the semantics of ``towr_utils.load_results``
    lines 57-99 (the coordinate flip / Euler re-wrap that follow in the reference are
    post-processing outside this stage)."""
    with open(file_path) as f:
        lines = [ln.replace('\n', '') for ln in f.readlines()]
    idx = 1
    dt = float(lines[idx]); idx += 2
    num_frames = int(lines[idx]); idx += 2
    num_feet = int(lines[idx]); idx += 2
    base_lin = np.reshape(np.array([float(x) for x in lines[idx].split(' ')]), (num_frames, 3)); idx += 2
    base_ang = np.reshape(np.array([float(x) for x in lines[idx].split(' ')]), (num_frames, 3)); idx += 2
    pos = []
    for _ in range(num_feet):
        pos.append(np.reshape(np.array([float(x) for x in lines[idx].split(' ')]), (num_frames, 3))); idx += 2
    frc = []
    for _ in range(num_feet):
        frc.append(np.reshape(np.array([float(x) for x in lines[idx].split(' ')]), (num_frames, 3))); idx += 2
    con = []
    for _ in range(num_feet):
        con.append(np.reshape(np.array([int(x) for x in lines[idx].split(' ')]), (num_frames,))); idx += 2
    return Solution(dt=dt, num_frames=num_frames, base_lin=base_lin, base_ang_deg=base_ang,
                    ee_pos=np.stack(pos), ee_force=np.stack(frc), contact=np.stack(con))

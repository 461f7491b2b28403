"""Synthetic "Mixamo-like" walking sequences generated directly in NLP-input space.

The reference ships no example inputs (no ``phys_optim_in_*`` directories, no
``foot_contacts.npy`` — SURVEY.md §8c/§8d), so every benchmark / test input is a
seeded synthetic sequence that has exactly the content of the four text files
``towr_utils.prepare_input`` writes (``src/utils/towr_utils.py:585-777``):
z-up frame, metres, 30 fps, a rigid two-contact-point (toe, heel) foot, hip
offsets in the base frame, per-frame point-mass inertia, a floor plane and the
per-end-effector contact durations.

Recipe: gait cycle, speed and yaw rate are randomised per seed,
mass is one of the three character masses of ``character_info_utils.py``
(36.5 / 73 / 146 kg) and 1 cm Gaussian jitter on the COM / feet targets mimics
kinematic noise.
"""
import numpy as np

from .io_formats import SeqInput, contact_durations

HEEL_DIST = 0.18


def _smoothstep(u):
    u = np.clip(u, 0.0, 1.0)
    return u * u * (3.0 - 2.0 * u)


def make_walk(seed=0, F=90, fps=30.0, randomize=True, tilt_deg=0.0, jitter=0.01,
              cycle=None, speed=None, yaw_rate=None, mass=None):
    """Return a :class:`SeqInput` for one synthetic walk.

    seed        RNG seed (``numpy.random.default_rng``)
    F           number of frames
    randomize   False -> the fixed "plumbing" walk of BASELINE config 1
    tilt_deg    floor tilt about the x axis (BASELINE config 5 uses 10 degrees)
    """
    rng = np.random.default_rng(seed)
    dt = 1.0 / fps
    if randomize:
        cycle = 2.0 * rng.uniform(0.4, 0.7) if cycle is None else cycle
        speed = rng.uniform(0.3, 1.5) if speed is None else speed
        yaw_rate = rng.uniform(-0.3, 0.3) if yaw_rate is None else yaw_rate
        mass = float(rng.choice([36.5, 73.0, 146.0])) if mass is None else mass
    else:
        cycle = 1.0 if cycle is None else cycle
        speed = 0.8 if speed is None else speed
        yaw_rate = 0.05 if yaw_rate is None else yaw_rate
        mass = 73.0 if mass is None else mass
    t = np.arange(F) * dt
    T = t[-1]

    # ---- base path ------------------------------------------------------
    yaw = yaw_rate * t
    heading = np.stack([np.cos(yaw), np.sin(yaw), np.zeros(F)], axis=1)
    left = np.stack([-np.sin(yaw), np.cos(yaw), np.zeros(F)], axis=1)
    path = np.zeros((F, 3))
    path[1:, :] = np.cumsum(0.5 * (heading[1:] + heading[:-1]) * speed * dt, axis=0)
    phase0 = rng.uniform(0.0, 1.0) if randomize else 0.0
    com = path.copy()
    com[:, 2] = 0.9 + 0.02 * np.sin(2 * np.pi * (2.0 * t / cycle + phase0))
    com += left * (0.02 * np.sin(2 * np.pi * (t / cycle + phase0)))[:, None]
    euler = np.stack([0.03 * np.sin(2 * np.pi * (t / cycle + phase0)),
                      0.05 + 0.02 * np.sin(2 * np.pi * (2 * t / cycle + phase0)),
                      yaw], axis=1)

    # ---- rigid feet: gait events as fractions of the cycle ---------------
    # heel strike 0.0, toe down 0.1, heel off 0.4, toe off 0.6, swing until 1.0
    ev_td, ev_ho, ev_to = 0.1, 0.4, 0.6
    pitch_strike, pitch_off = 0.35, 0.5      # rad, toe-up at strike / heel-up at toe-off
    step_len = speed * cycle
    half_width = 0.09

    def path_at(time):
        """Position/heading of the base path at arbitrary time (extrapolating)."""
        time = np.asarray(time, dtype=float)
        yw = yaw_rate * time
        if abs(yaw_rate) < 1e-9:
            pos = np.stack([speed * time, np.zeros_like(time), np.zeros_like(time)], axis=-1)
        else:
            r = speed / yaw_rate
            pos = np.stack([r * np.sin(yw), r * (1 - np.cos(yw)), np.zeros_like(time)], axis=-1)
        hd = np.stack([np.cos(yw), np.sin(yw), np.zeros_like(time)], axis=-1)
        lf = np.stack([-np.sin(yw), np.cos(yw), np.zeros_like(time)], axis=-1)
        return pos, hd, lf

    def foot(side, offset):
        """side=+1 left/-1 right, offset = cycle phase offset in [0,1)."""
        toe = np.zeros((F, 3)); heel = np.zeros((F, 3))
        toe_c = np.zeros(F, dtype=np.int64); heel_c = np.zeros(F, dtype=np.int64)
        up = np.array([0.0, 0.0, 1.0])
        for i in range(F):
            ph = t[i] / cycle + offset + phase0
            k = np.floor(ph)
            u = ph - k
            # footprint k: heel lands where the path is at mid-stance of this cycle
            t_mid = (k - offset - phase0 + 0.3) * cycle

            def footprint(tm):
                p, hd, lf = path_at(tm)
                heel_p = p + lf * (side * half_width) - hd * (0.5 * HEEL_DIST)
                return heel_p, hd

            heel_k, hd_k = footprint(t_mid)
            toe_k = heel_k + hd_k * HEEL_DIST
            if u < ev_td:                       # heel down, toe rotating down about the heel
                phi = -pitch_strike * (1.0 - _smoothstep(u / ev_td))
                heel[i] = heel_k
                toe[i] = heel_k + HEEL_DIST * (np.cos(phi) * hd_k - np.sin(phi) * up)
                heel_c[i] = 1
            elif u < ev_ho:                     # flat foot
                heel[i] = heel_k; toe[i] = toe_k
                heel_c[i] = 1; toe_c[i] = 1
            elif u < ev_to:                     # toe down, heel rising about the toe
                phi = pitch_off * _smoothstep((u - ev_ho) / (ev_to - ev_ho))
                toe[i] = toe_k
                heel[i] = toe_k + HEEL_DIST * (-np.cos(phi) * hd_k + np.sin(phi) * up)
                toe_c[i] = 1
            else:                               # swing to the next footprint's heel-strike pose
                s = _smoothstep((u - ev_to) / (1.0 - ev_to))
                heel_n, hd_n = footprint(t_mid + cycle)
                toe_start = toe_k
                toe_end = heel_n + HEEL_DIST * (np.cos(-pitch_strike) * hd_n + np.sin(pitch_strike) * up)
                phi = pitch_off + (-pitch_strike - pitch_off) * s
                hd_s = hd_k + (hd_n - hd_k) * s
                hd_s = hd_s / np.linalg.norm(hd_s)
                lift = 0.08 * np.sin(np.pi * (u - ev_to) / (1.0 - ev_to))
                toe[i] = toe_start + (toe_end - toe_start) * s + up * lift
                heel[i] = toe[i] + HEEL_DIST * (-np.cos(phi) * hd_s + np.sin(phi) * up)
        return toe, heel, toe_c, heel_c

    ltoe, lheel, ltoe_c, lheel_c = foot(+1.0, 0.15)
    rtoe, rheel, rtoe_c, rheel_c = foot(-1.0, 0.65)

    # ---- skeleton quantities ----------------------------------------------
    hip_l = np.tile(np.array([0.0, half_width, -0.05]), (F, 1))
    hip_r = np.tile(np.array([0.0, -half_width, -0.05]), (F, 1))
    hip_l += rng.normal(0, 0.002, hip_l.shape)
    hip_r += rng.normal(0, 0.002, hip_r.shape)

    def rot(e):
        x, y, z = e
        cx, sx, cy, sy, cz, sz = np.cos(x), np.sin(x), np.cos(y), np.sin(y), np.cos(z), np.sin(z)
        return np.array([[cy * cz, cz * sx * sy - cx * sz, sx * sz + cx * cz * sy],
                         [cy * sz, cx * cz + sx * sy * sz, cx * sy * sz - cz * sx],
                         [-sy, cy * sx, cx * cy]])

    hipw_l = np.stack([rot(euler[i]) @ hip_l[i] + com[i] for i in range(F)])
    hipw_r = np.stack([rot(euler[i]) @ hip_r[i] + com[i] for i in range(F)])
    leg_len = 1.04 * max(np.linalg.norm(ltoe - hipw_l, axis=1).max(), np.linalg.norm(rtoe - hipw_r, axis=1).max())
    heel_len = 1.04 * max(np.linalg.norm(lheel - hipw_l, axis=1).max(), np.linalg.norm(rheel - hipw_r, axis=1).max())

    scale = mass / 73.0
    inertia = np.zeros((F, 6))
    base_I = np.array([9.0, 9.0, 1.2]) * scale
    inertia[:, :3] = base_I * (1.0 + 0.05 * rng.normal(size=(F, 3)))
    inertia[:, 3:] = 0.05 * scale * rng.normal(size=(F, 3))

    # ---- target jitter (kinematic noise) ----------------------------------
    if jitter > 0:
        com_t = com + rng.normal(0, jitter, com.shape)
        ltoe_t = ltoe + rng.normal(0, jitter, ltoe.shape)
        lheel_t = lheel + rng.normal(0, jitter, lheel.shape)
        rtoe_t = rtoe + rng.normal(0, jitter, rtoe.shape)
        rheel_t = rheel + rng.normal(0, jitter, rheel.shape)
        euler_t = euler + rng.normal(0, 0.01, euler.shape)
    else:
        com_t, ltoe_t, lheel_t, rtoe_t, rheel_t, euler_t = com, ltoe, lheel, rtoe, rheel, euler

    # ---- floor (optionally tilted about x: rotate the whole scene) ---------
    normal = np.array([0.0, 0.0, 1.0]); point = np.array([0.0, 0.0, 0.0])
    if tilt_deg != 0.0:
        a = np.radians(tilt_deg)
        Rx = np.array([[1, 0, 0], [0, np.cos(a), -np.sin(a)], [0, np.sin(a), np.cos(a)]])
        normal = Rx @ normal
        com_t, ltoe_t, lheel_t, rtoe_t, rheel_t = [v @ Rx.T for v in (com_t, ltoe_t, lheel_t, rtoe_t, rheel_t)]
        euler_t = euler_t.copy(); euler_t[:, 0] += a

    seq = SeqInput(
        F=F, dt=dt, hip_l=hip_l, hip_r=hip_r, leg_len=float(leg_len), heel_len=float(heel_len),
        heel_dist=HEEL_DIST, mass=float(mass), inertia=inertia, com=com_t, euler=euler_t,
        ltoe=ltoe_t, lheel=lheel_t, rtoe=rtoe_t, rheel=rheel_t, normal=normal, point=point,
        start_contact=[int(ltoe_c[0]), int(lheel_c[0]), int(rtoe_c[0]), int(rheel_c[0])],
        durations=[contact_durations(ltoe_c, dt), contact_durations(lheel_c, dt),
                   contact_durations(rtoe_c, dt), contact_durations(rheel_c, dt)])
    seq.contacts = np.stack([lheel_c, ltoe_c, rheel_c, rtoe_c], axis=1)   # foot_contacts.npy column order (README.md:87)
    return seq


def make_batch(B, F=90, seed0=0, **kw):
    """BASELINE config 2/3: B independent sequences, seeds seed0..seed0+B-1."""
    return [make_walk(seed=seed0 + i, F=F, **kw) for i in range(B)]


def make_kin_clip(seed, F, offsets_template, parents, upright=False):
    """A synthetic input of the kinematic optimisation (`optimize_trajectory`'s arguments) on the combined 28-joint skeleton: smooth
    random joint angles with quiet legs, a swing-knee bend, a root drifting in front of the camera (y down, z forward, centimetres:
    the monocular-total-capture frame), noisy 3D joints / 2D projections / confidences, alternating foot contacts with one spurious
    label.  Same recipe as tests/golden/make_kinopt_golden.py.  `offsets_template` / `parents`: the skeleton's rest pose."""
    from . import kinematic_optimizer as kopt
    from . import skeleton_io as sio
    OFFSETS, PARENTS = np.asarray(offsets_template, dtype=np.float64), np.asarray(parents)
    rng = np.random.default_rng(seed)
    nj = 28
    t = np.arange(F)[:, None, None] / 30.0
    amp = rng.uniform(0.05, 0.35, size=(1, nj, 3)); ph = rng.uniform(0, 2 * np.pi, size=(1, nj, 3)); fr = rng.uniform(0.5, 2.0, size=(1, nj, 3))
    amp[:, 1:13] = rng.uniform(0.01, 0.04, size=(1, 12, 3))
    eul = amp * np.sin(2 * np.pi * fr * t + ph)
    eul[:, 0] += np.array([0.1 + (np.pi if upright else 0.0), 0.4, 0.05])      # upright: the template's legs (along -y) point down in the camera frame (y down), as a person stands
    half = F // 2
    swing = np.sin(np.pi * np.clip((np.arange(F) - half) / max(F - half - 1, 1), 0, 1)) ** 2
    eul[:, 2, 0] += 0.9 * swing; eul[:, 8, 0] += 0.9 * swing[::-1]
    rot = sio.quat_from_euler(eul, order='xyz', world=True)
    offsets = OFFSETS * rng.uniform(0.9, 1.15)
    root = np.array([20.0, 40.0, 320.0]) + np.arange(F)[:, None] * np.array([1.5, 0.05, -0.8]) * (10.0 / F) + rng.normal(size=(F, 3)) * 0.3
    pos = np.repeat(offsets[None], F, axis=0); pos[:, 0] = root
    gp = sio.positions_global(sio.Motion(rot, pos, np.tile([1.0, 0, 0, 0], (nj, 1)), offsets, PARENTS))
    gabs = gp[:, kopt.BACKWARD_MAPPING]
    p3 = gabs - root[:, None] + rng.normal(size=gabs.shape) * 1.5
    p3[:, kopt.ROOT_IDX] = 0.0
    focal = np.array([2000.0, 2000.0])
    p2 = np.stack([focal[0] * gabs[..., 0] / gabs[..., 2] + 960.0, focal[1] * gabs[..., 1] / gabs[..., 2] + 540.0], axis=2) + rng.normal(size=(F, nj, 2)) * 3.0
    conf = rng.uniform(0.3, 1.0, size=(F, nj)); conf[rng.uniform(size=(F, nj)) < 0.05] = 0.0
    p2[:, 25:] = 0.0; conf[:, 25:] = 0.0
    ang = 2.0 * np.arccos(np.clip(rot[..., 0], -1, 1))
    ax = rot[..., 1:] / np.maximum(np.linalg.norm(rot[..., 1:], axis=-1, keepdims=True), 1e-12)
    vel = np.zeros((F, nj))
    vel[:half + 1, 19] = 1; vel[:half + 1, 20] = 1; vel[:half, 21] = 1
    vel[half:, 22] = 1; vel[half:, 23] = 1; vel[half + 1:, 24] = 1
    vel[half + (F - half) // 2, 21] = 1
    return dict(poses2D=p2, joint_conf_2d=conf, poses3D=p3, root_pos=root + rng.normal(size=root.shape), joint_angles=-(ax * ang[..., None]) + rng.normal(size=(F, nj, 3)) * 0.03,
                offsets=OFFSETS, parents=PARENTS, ppx=960.0, ppy=540.0, camFocal=focal, velConstraints=vel)

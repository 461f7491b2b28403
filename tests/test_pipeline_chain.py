"""The three rows in sequence on one small clip, CPU only: `prepare_input` -> the physics NLP (host emulation of the
kernel source instead of the HIP launch) -> solution file -> `apply_results` (host emulation of the IK kernel source) ->
BVH.  Checks that each stage accepts what the previous one produces; parity of each stage is covered by its own tests."""
import os
import sys

import numpy as np

import chd_amd  # noqa: F401
from chd_amd import apply_results as ar
from chd_amd import io_formats as iof
from chd_amd import prepare_input as pi
from chd_amd import skeleton_io as sk
from chd_amd.phys_capi import default_config

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, 'golden'))
sys.path.insert(0, os.path.join(HERE, 'host_emu'))


def test_bvh_to_physics_to_bvh(tmp_path):
    import emu
    import ik_emu
    from make_apply_golden import CHARACTER, NAMES
    emu.build()
    g = np.load(os.path.join(HERE, 'golden', 'apply_golden.npz'))
    bvh = str(tmp_path / 'in.bvh')
    open(bvh, 'wb').write(g['bvh_text'].tobytes())
    character = ar.Character(**CHARACTER)
    motion, names, _ = sk.load_bvh(bvh)
    floor = (np.array([0.0, 0.0, 1.0]), np.array([0.0, 0.0, 0.0]))
    seq = pi.prepare_sequence(motion, floor, g['prep_contacts'], character, 0, motion.n_frames, 1 / 30)
    # physics: the kinematic stages (no dynamics yet) must solve on any input; this clip is noise, not a walk, so the
    # dynamics stages are only required to terminate with a status
    e = emu.EmuProblem(seq, default_config(max_iter=[60] * 6))
    e.solve(0, 4)
    stats, snaps = e.results()
    assert int(stats[0, 0]) == 0 and int(stats[1, 0]) == 0
    assert all(int(s) in (0, 1, -1, -2) for s in stats[2:5, 0])
    s0 = snaps[0]                                                            # sol_out_no_dynamics
    sol = iof.Solution(dt=seq.dt, num_frames=s0['num_frames'], base_lin=s0['base_lin'], base_ang_deg=s0['base_ang_deg'],
                       ee_pos=s0['ee_pos'], ee_force=s0['ee_force'], contact=s0['contact'])
    assert sol.base_lin.shape[0] >= seq.F
    p = str(tmp_path / 'sol_out_no_dynamics.txt')
    iof.write_solution(sol, p)
    # back-projection
    class EmuIk:
        def solve(self, seqs):
            return ik_emu.solve(seqs)
    out = str(tmp_path / 'clip_synth_no_dynamics.bvh')
    task, = ar.apply_results_batch([p], [bvh], [out], character, EmuIk(), starts=[0], ends=[motion.n_frames])
    res = ar.load_towr_results(p)
    gp = sk.positions_global(task.motion)
    before = sk.positions_global(task.motion_og)
    for k, j in enumerate(character.toe_inds):
        want = res.feet_pos[:seq.F, k] * 100.0
        assert np.linalg.norm(gp[:, j] - want, axis=1).mean() < np.linalg.norm(before[:, j] - want, axis=1).mean()      # IK pulled the toes to the optimised positions
    back, names2, _ = sk.load_bvh(out)
    assert names2 == NAMES and back.n_frames == motion.n_frames and back.n_joints == motion.n_joints

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


def test_in_memory_pipeline_matches_the_file_chain(tmp_path):
    """pipeline.run_clips (no intermediate files) against the same chain through sol_out_*.txt: same BVH up to the
    10 significant digits of the solution file."""
    import emu
    import ik_emu
    from make_apply_golden import CHARACTER
    from chd_amd import pipeline
    emu.build()
    g = np.load(os.path.join(HERE, 'golden', 'apply_golden.npz'))
    bvh = str(tmp_path / 'in.bvh')
    open(bvh, 'wb').write(g['bvh_text'].tobytes())
    character = ar.Character(**CHARACTER)
    floor = (np.array([0.0, 0.0, 1.0]), np.array([0.0, 0.0, 0.0]))

    class Result:
        def __init__(self, snaps, stats):
            self.snapshots = [iof.Solution(dt=1 / 30, num_frames=s['num_frames'], base_lin=s['base_lin'], base_ang_deg=s['base_ang_deg'],
                                           ee_pos=s['ee_pos'], ee_force=s['ee_force'], contact=s['contact']) for s in snaps]
            self.stage_status = [int(v) for v in stats[:, 0]]

    class EmuPhys:                                            # the kernel source on the host, one sequence after the other
        def solve(self, seqs):
            res = []
            for q in seqs:
                e = emu.EmuProblem(q, default_config(max_iter=[60] * 6))
                e.solve(0, 1)
                stats, snaps = e.results()
                res.append(Result(snaps, stats))
            return res, None

    class EmuIk:
        def solve(self, seqs):
            return ik_emu.solve(seqs)

    outs = [str(tmp_path / 'a.bvh'), str(tmp_path / 'b.bvh')]
    clips = [pipeline.Clip(bvh=bvh, floor=floor, contacts=g['prep_contacts'], out_bvh={'no_dynamics': outs[0], 'durations': str(tmp_path / 'never.bvh')}),
             pipeline.Clip(bvh=bvh, floor=floor, contacts=g['prep_contacts'], out_bvh={'no_dynamics': outs[1]}, start=1, end=13)]
    res = pipeline.run_clips(clips, character, EmuPhys(), EmuIk())
    assert res[0].written == [outs[0]] and res[1].written == [outs[1]] and not os.path.exists(str(tmp_path / 'never.bvh'))      # stage not run: no file
    assert res[0].seq.F == 14 and res[1].seq.F == 12
    # the same clip through the text file
    p = str(tmp_path / 'sol.txt')
    iof.write_solution(res[0].phys.snapshots[0], p)
    ref_out = str(tmp_path / 'file_chain.bvh')
    ar.apply_results_batch([p], [bvh], [ref_out], character, EmuIk(), starts=[0], ends=[14])
    a = np.array(open(outs[0]).read().split('Time:')[1].split(), dtype=np.float64)
    b = np.array(open(ref_out).read().split('Time:')[1].split(), dtype=np.float64)
    assert a.shape == b.shape and np.abs(a - b).max() < 1e-4
    assert sk.load_bvh(outs[1])[0].n_frames == 12
    # prepare_input's batched tensor path (torch's CPU device here) feeds the solver the same sequences
    res_t = pipeline.run_clips(clips, character, EmuPhys(), EmuIk(), prepare_device='cpu')
    for r0, r1 in zip(res, res_t):
        assert r1.seq.F == r0.seq.F and np.allclose(r1.seq.com, r0.seq.com, rtol=1e-12, atol=1e-14) and np.allclose(r1.seq.inertia, r0.seq.inertia, rtol=1e-11, atol=1e-13)

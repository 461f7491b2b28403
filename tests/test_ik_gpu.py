"""GPU parity of the IK back-projection step through its C ABI: libchd_ik.so on an MI355X against
the vectors produced by the reference solver (first run on an MI355X at the start of round 2: profiles/r02a_round_start)."""
import os

import numpy as np
import pytest

import chd_amd  # noqa: F401
from oracle import ik_oracle as ik

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'ik_golden.npz')


def test_hip_path_matches_reference_vectors():
    torch = pytest.importorskip('torch')
    if not torch.cuda.is_available():
        pytest.skip('needs an MI355X')
    from chd_amd.ik_backproject import IkBackProject
    from chd_amd.ik_capi import ChdIkConfig
    g = np.load(GOLD)
    cases = [dict(parents=g['c%d_parents' % c], target_joints=g['c%d_target_joints' % c], targets=g['c%d_targets' % c],
                  rot=g['c%d_rot0' % c], pos=g['c%d_pos0' % c]) for c in range(int(g['n_cases']))]
    for iters in (1, 30):
        outs = IkBackProject(device=0, config=ChdIkConfig.default(iterations=iters)).solve(cases)
        for c, (rot, pos) in enumerate(outs):
            gp = ik.positions_global(rot, pos, cases[c]['parents'])
            assert np.allclose(gp, g['c%d_it%d_gpos' % (c, iters)], rtol=1e-8, atol=1e-8)
            assert np.allclose(pos, g['c%d_it%d_pos' % (c, iters)], rtol=1e-8, atol=1e-8)


def test_apply_results_on_gpu_matches_reference(tmp_path):
    """The whole back-projection (solution file + BVH in, BVH out) with the HIP solver against the reference's
    `apply_results` vectors (tests/golden/apply_golden.npz)."""
    torch = pytest.importorskip('torch')
    if not torch.cuda.is_available():
        pytest.skip('needs an MI355X')
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden'))
    from make_apply_golden import CHARACTER
    from chd_amd import apply_results as ar
    from chd_amd import skeleton_io as sk
    from chd_amd.ik_backproject import IkBackProject
    g = np.load(os.path.join(os.path.dirname(GOLD), 'apply_golden.npz'))
    bvh = str(tmp_path / 'in.bvh'); sol = str(tmp_path / 'sol.txt')
    open(bvh, 'wb').write(g['bvh_text'].tobytes()); open(sol, 'wb').write(g['sol_text'].tobytes())
    s, e = [int(v) for v in g['start_end']]
    outs = [str(tmp_path / ('out%d.bvh' % k)) for k in range(3)]
    tasks = ar.apply_results_batch([sol] * 3, [bvh] * 3, outs, ar.Character(**CHARACTER), IkBackProject(device=0), starts=[s] * 3, ends=[e] * 3)
    for t in tasks:
        assert np.abs(sk.positions_global(t.motion) - g['ik_gpos']).max() < 1e-7
        assert np.abs(t.motion.positions - g['ik_pos']).max() < 1e-7
    ref = g['out_bvh_text'].tobytes().decode()
    got = open(outs[2]).read()
    a = np.array(got[got.index('Time:') + 5:].split(), dtype=np.float64)
    b = np.array(ref[ref.index('Time:') + 5:].split(), dtype=np.float64)
    assert got[:got.index('MOTION')] == ref[:ref.index('MOTION')] and np.abs(a - b).max() <= 2e-6


def test_in_memory_pipeline_on_gpu(tmp_path):
    """pipeline.run_clips with both HIP libraries: BVH + floor + contacts in, BVH out, no intermediate files."""
    torch = pytest.importorskip('torch')
    if not torch.cuda.is_available():
        pytest.skip('needs an MI355X')
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden'))
    from make_apply_golden import CHARACTER
    from chd_amd import apply_results as ar
    from chd_amd import pipeline
    from chd_amd import skeleton_io as sk
    from chd_amd.ik_backproject import IkBackProject
    from chd_amd.phys_optim import PhysOptim, default_config
    g = np.load(os.path.join(os.path.dirname(GOLD), 'apply_golden.npz'))
    bvh = str(tmp_path / 'in.bvh')
    open(bvh, 'wb').write(g['bvh_text'].tobytes())
    floor = (np.array([0.0, 0.0, 1.0]), np.array([0.0, 0.0, 0.0]))
    outs = [str(tmp_path / ('c%d.bvh' % k)) for k in range(4)]
    clips = [pipeline.Clip(bvh=bvh, floor=floor, contacts=g['prep_contacts'], out_bvh={'no_dynamics': o}) for o in outs]
    phys = PhysOptim(device=0, config=default_config(max_iter=[60] * 6))
    try:
        res = pipeline.run_clips(clips, ar.Character(**CHARACTER), phys, IkBackProject(device=0))
    finally:
        phys.close()
    for r, o in zip(res, outs):
        assert r.phys.stage_status[0] == 0 and r.written == [o]
        assert sk.load_bvh(o)[0].n_frames == 14
    assert open(outs[0]).read() == open(outs[3]).read()              # identical clips, identical files

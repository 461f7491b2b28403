"""GPU parity: the HIP physics path (through the C ABI) against the CPU oracle on the same
seeded inputs.  Tolerance for trajectories: 1e-3 relative L2 (BASELINE.json north_star);
contact flags bit-exact."""
import numpy as np
import pytest

import chd_amd
from chd_amd.synth import make_walk

from common import oracle_run, rel_max, snapshot_errors

pytestmark = pytest.mark.gpu
CAP = [300] * 6


@pytest.fixture(scope='module')
def solver():
    from chd_amd.phys_optim import PhysOptim, default_config
    s = PhysOptim(device=0, config=default_config(max_iter=CAP))
    yield s
    s.close()


def test_eval_matches_oracle(solver, oracle_lib):
    """f, gradient, constraint values, Jacobian and Gauss-Newton Hessian of every stage: 1e-10 relative."""
    from oracle.oracle import OracleProblem
    seq = make_walk(seed=0, F=60, randomize=True)
    o = OracleProblem(seq)
    b = solver.upload([seq])
    rng = np.random.default_rng(1)
    for st in range(5):
        o.set_stage(st)
        assert b.sizes(0, st)['n'] == o.n and b.sizes(0, st)['m'] == o.m
        x = o.get_x() if st == 0 else x0
        x0 = o.get_x() if st == 0 else x0
        x = x0[:o.n].copy() if st != 4 else np.concatenate([x0, o.get_x()[len(x0):]])
        x = x + 0.01 * rng.normal(size=x.size) * (1.0 if st != 4 else np.concatenate([np.ones(len(x0)), 0.01 * np.ones(x.size - len(x0))]))
        fo, go, co, Jo, Ho = o.eval(x, jac=True, hess=True)
        r = b.debug_eval(0, st, x)
        assert r['err'] == 0
        assert abs(r['f'] - fo) <= 1e-10 * abs(fo)
        assert rel_max(r['g'], go) < 1e-10 and rel_max(r['c'], co) < 1e-10
        assert rel_max(r['J'], Jo) < 1e-10 and rel_max(r['H'], Ho) < 1e-10
    b.free()


@pytest.mark.parametrize('seed,F', [(0, 60), (5, 90)])
def test_solve_matches_oracle(solver, oracle_lib, seed, F):
    seq = make_walk(seed=seed, F=F, randomize=True)
    ostats, osnaps = oracle_run(seq, CAP)
    res, st = solver.solve([seq])
    r = res[0]
    for k in range(3):
        e = snapshot_errors(r.snapshots[k], osnaps[k])
        assert e['n_samples'][0] == e['n_samples'][1]
        assert e['contact_mismatch'] == 0
        for key in ('base_lin', 'base_ang_deg', 'ee_pos', 'ee_force'):
            assert e[key] < 1e-3, (k, key, e)
    for stg in range(len(ostats)):
        assert r.stage_status[stg] == ostats[stg][0]

"""The oracle's IPOPT-like mode (L-BFGS(6) + filter line search: oracle/ipm_solver.hpp IpmOptions::lbfgs / ::filter) and the committed comparison of the
shipped algorithm with it (tests/golden/ipopt_like_golden.json, made by tests/golden/make_ipopt_like_golden.py) -- the explicit PROXY for north_star's
"within 1e-3 rel-L2 of the IPOPT reference", which cannot be measured (the reference binary is not buildable here: SURVEY 8c)."""
import json
import os

import numpy as np

import chd_amd  # noqa: F401
from chd_amd.synth import make_walk

HERE = os.path.dirname(os.path.abspath(__file__))
CAPS = [7000, 7000, 7000, 2500, 2000, 7000]


def test_fixture_is_complete_and_says_what_it_is():
    o = json.load(open(os.path.join(HERE, 'golden', 'ipopt_like_golden.json')))
    assert 'PROXY' in o['proxy_for'].upper() or 'UNMEASURED' in o['proxy_for']
    for key, n in (('40_frames', 32), ('90_frames', 8)):
        s = o['summary'][key]
        assert s['sequences'] == n
        for q in ('base_lin', 'base_ang_deg', 'ee_pos', 'ee_force'):
            assert len(s['median_rel_l2'][q]) == 3 and len(s['max_rel_l2'][q]) == 3
        # what the proxy says, so that nobody reads more into the parity claims than is there: positions agree to 1e-2, not 1e-3; forces do not agree
        assert s['sequences_within_1e-3_on_every_trajectory'] == 0
        assert max(s['median_rel_l2']['base_lin']) < 3e-2 and max(s['median_rel_l2']['ee_pos']) < 2e-2 and s['median_rel_l2']['ee_force'][1] > 0.1
    rows = o['per_sequence']
    assert len(rows) == 40 and all(len(r['rel_l2']) == 3 and len(r['rel_l2'][0]) == 4 for r in rows)


def test_filter_mode_solves_every_stage_and_lands_near_the_shipped_solution(oracle_lib):
    """one 40-frame sequence, live (~40 s): the IPOPT-like solve converges on all five stages, at the distance from the shipped solve the fixture records"""
    from common import oracle_run, rel_l2
    from oracle.oracle import lib
    seq = make_walk(seed=2, F=40, randomize=True)
    sa, a = oracle_run(seq, CAPS)
    lib().orc_set_ipopt_like(2)
    try:
        sb, b = oracle_run(seq, CAPS)
    finally:
        lib().orc_set_ipopt_like(0)
    assert [s[0] for s in sb[:5]] == [0, 0, 0, 0, 0]
    assert sum(s[1] for s in sb) > 10 * sum(s[1] for s in sa)                # a limited-memory model needs an order of magnitude more iterations
    o = json.load(open(os.path.join(HERE, 'golden', 'ipopt_like_golden.json')))
    row = [r for r in o['per_sequence'] if r['seed'] == 2 and r['frames'] == 40][0]
    for k in range(3):
        d = rel_l2(a[k]['base_lin'], b[k]['base_lin'])
        assert d <= 3 * row['rel_l2'][k][0] + 1e-3 and d <= 5e-2, (k, d, row['rel_l2'][k][0])
        assert sb[[1, 3, len(sb) - 1][k]][2] <= 1.3 * sa[[1, 3, len(sa) - 1][k]][2]       # objective within 30 % of the shipped solve's

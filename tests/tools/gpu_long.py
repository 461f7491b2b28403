"""BASELINE configs[4] on the GPU: one 600-frame sequence, floor tilted by 10 degrees (the stress case for the KKT band),
alone in a launch; compared with the oracle's committed result when the fixture holds it.

    python tests/tools/gpu_long.py [frames] [tilt]
"""
import os
import sys
import time
sys.path.insert(0, '.'); sys.path.insert(0, 'tests'); sys.path.insert(0, 'tests/golden')
import numpy as np
import chd_amd  # noqa: E402,F401
from chd_amd.phys_optim import PhysOptim, default_config  # noqa: E402
import make_bench_parity_golden as mk  # noqa: E402

F = int(sys.argv[1]) if len(sys.argv) > 1 else 600
tilt = float(sys.argv[2]) if len(sys.argv) > 2 else 10.0
seq = mk.make_case(0, F, tilt)
s = PhysOptim(0, default_config(stall_window=int(os.environ.get('STALL', '0'))))
t0 = time.time(); b = s.upload([seq]); t1 = time.time(); st = b.solve(); t2 = time.time(); r = b.fetch()[0]
print('F %d tilt %.0f: upload %.2fs solve %.2fs kernel %.0f + %.0f ms; sizes %s' % (F, tilt, t1 - t0, t2 - t1, st['kernel_ms'][0], st['kernel_ms'][1], r.sizes))
print('stages', list(zip(r.stage_status, r.stage_iters)), 'viol', ['%.1e' % v for v in r.stage_constr_viol])
print('phase share', {k: round(st['phase_ms'][i] / max(1e-9, st['phase_ms'][5]), 3) for k, i in (('eval', 0), ('eval_values', 1), ('factor', 2), ('solve', 3), ('matvec', 4))})
print('factorisation ms', {k: round(st['phase_ms'][i], 1) for k, i in (('copy', 6), ('panel_load', 8), ('row_solve', 9), ('store_wait', 10), ('lookahead_wavefront', 11), ('border', 12))}, 'substitution ms', [round(st['phase_ms'][i], 1) for i in (16, 17, 18, 19, 20)], 'factorisations', st['total_factorizations'])
g = np.load(os.path.join('tests', 'golden', 'bench_parity_golden.npz'))
key = mk.case_key(0, F, tilt)
if key + '_status' in g.files:
    worst = 0.0
    for k in range(3):
        sn = r.snapshots[k]
        for name, val in (('base_lin', sn.base_lin), ('base_ang_deg', sn.base_ang_deg), ('ee_pos', sn.ee_pos), ('ee_force', sn.ee_force)):
            ref = g['%s_snap%d_%s' % (key, k, name)]
            if np.linalg.norm(ref) > 0:
                worst = max(worst, float(np.linalg.norm(np.asarray(val) - ref) / np.linalg.norm(ref)))
    print('oracle', list(zip(g[key + '_status'], g[key + '_iters'])), 'worst rel-L2 %.2e' % worst)
else:
    print('no oracle result for', key, 'in the fixture')

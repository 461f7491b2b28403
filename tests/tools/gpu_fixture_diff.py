"""HIP path vs the committed oracle fixture, sequence by sequence (what the parity test asserts, as a table).
    python tests/tools/gpu_fixture_diff.py [n_flat] [n_tilted]"""
import os
import sys
sys.path.insert(0, '.'); sys.path.insert(0, 'tests/golden')
import numpy as np
import chd_amd  # noqa: E402,F401
from chd_amd.phys_optim import PhysOptim, default_config  # noqa: E402
import make_bench_parity_golden as mk  # noqa: E402
nf = int(sys.argv[1]) if len(sys.argv) > 1 else 128
nt = int(sys.argv[2]) if len(sys.argv) > 2 else 32
g = np.load(os.path.join('tests', 'golden', 'bench_parity_golden.npz'))
cases = mk.FLAT[:nf] + mk.TILTED[:nt]
s = PhysOptim(0, default_config(max_iter=mk.CAPS))
res, st = s.solve([mk.make_case(*c) for c in cases])
bad = 0
for c, r in zip(cases, res):
    key = mk.case_key(*c)
    gs = list(g[key + '_status']); gi = list(g[key + '_iters'])
    err = 0.0; cm = 0
    for k in range(3):
        sn = r.snapshots[k]
        for name, val in (('base_lin', sn.base_lin), ('base_ang_deg', sn.base_ang_deg), ('ee_pos', sn.ee_pos), ('ee_force', sn.ee_force)):
            ref = g['%s_snap%d_%s' % (key, k, name)]
            if np.linalg.norm(ref) > 0 and ref.shape == np.asarray(val).shape:
                err = max(err, float(np.linalg.norm(np.asarray(val) - ref) / np.linalg.norm(ref)))
        cm += int(np.abs(np.asarray(sn.contact, dtype=np.int64) - g['%s_snap%d_contact' % (key, k)]).sum())
    same = list(r.stage_status[:len(gs)]) == gs and list(r.stage_iters[:len(gi)]) == gi
    if not same or err > 1e-9 or cm:
        bad += 1
        print(key, 'gpu', list(zip(r.stage_status, r.stage_iters))[:len(gs)], 'oracle', list(zip(gs, gi)), 'rel-L2 %.2e contact mismatches %d' % (err, cm), 'nfact', r.stage_factorizations[:len(gs)], 'E0', ['%.1e' % v for v in r.stage_kkt_error[:len(gs)]])
print('%d of %d sequences differ (status / iterations / rel-L2 > 1e-9 / contact flags)' % (bad, len(cases)))

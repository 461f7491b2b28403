"""BASELINE config 5 (single 600-frame sequence, tilted floor) and a multi-seed parity sweep on the GPU."""
import sys, time
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np
import chd_amd
from chd_amd.synth import make_walk
from chd_amd.phys_optim import PhysOptim, default_config
from common import oracle_run, snapshot_errors

s = PhysOptim(0, default_config())
seq = make_walk(seed=5, F=600, randomize=True, tilt_deg=10.0)
t0 = time.time(); b = s.upload([seq]); t1 = time.time()
for st in range(5):
    print('stage', st, b.sizes(0, st))
stt = b.solve(); t2 = time.time(); res = b.fetch()
r = res[0]
print('F=600: upload %.2fs solve %.2fs' % (t1 - t0, t2 - t1), list(zip(r.stage_status, r.stage_iters)), r.sizes)
print({k: stt[k] for k in ('kernel_ms', 'total_iters', 'total_factorizations', 'max_seq_ms')})
print('phase_ms', [round(x, 1) for x in stt['phase_ms']])
print('finite', all(np.isfinite(sn.ee_force).all() for sn in r.snapshots), 'viol', r.stage_constr_viol)
b.free()
if len(sys.argv) > 1:
    worst = 0
    for seed in [int(a) for a in sys.argv[1:]]:
        sq = make_walk(seed=seed, F=90, randomize=True)
        rs, _ = s.solve([sq])
        ostats, osnaps = oracle_run(sq, [7000, 7000, 7000, 2500, 2000, 7000])
        errs = [snapshot_errors(rs[0].snapshots[k], osnaps[k]) for k in range(3)]
        w = max(max(e['base_lin'], e['base_ang_deg'], e['ee_pos'], e['ee_force']) for e in errs)
        worst = max(worst, w)
        print('seed', seed, 'gpu', list(zip(rs[0].stage_status, rs[0].stage_iters)), 'oracle', [(a, b_) for a, b_, c in ostats], 'max rel-L2 %.2e' % w,
              'contact mismatches', sum(e['contact_mismatch'] for e in errs))
    print('worst rel-L2 over seeds', worst)

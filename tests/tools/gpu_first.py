"""First GPU contact: eval + solve parity on a handful of sequences, and a small timing."""
import sys, time, json
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np
import chd_amd
from chd_amd.synth import make_walk
from chd_amd.phys_optim import PhysOptim, default_config
from common import oracle_run, snapshot_errors

cap = [300] * 6
s = PhysOptim(0, default_config(max_iter=cap))
for seed, F in [(0, 60), (5, 90), (1, 90)]:
    seq = make_walk(seed=seed, F=F, randomize=True)
    t0 = time.time(); res, st = s.solve([seq]); t1 = time.time()
    ostats, osnaps = oracle_run(seq, cap); t2 = time.time()
    r = res[0]
    print('seed', seed, 'F', F, 'gpu %.2fs oracle %.2fs' % (t1 - t0, t2 - t1), st)
    print('  gpu   ', list(zip(r.stage_status, r.stage_iters)))
    print('  oracle', [(a, b) for a, b, c in ostats])
    for k in range(3):
        print('  snap', k, snapshot_errors(r.snapshots[k], osnaps[k]))
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
seqs = [make_walk(seed=i, F=90, randomize=True) for i in range(B)]
t0 = time.time(); b = s.upload(seqs); t1 = time.time(); st = b.solve(); t2 = time.time(); res = b.fetch(); t3 = time.time()
print('batch', B, 'upload %.2fs solve %.2fs fetch %.2fs' % (t1 - t0, t2 - t1, t3 - t2), st)
for i, r in enumerate(res):
    print(i, list(zip(r.stage_status, r.stage_iters)), r.sizes)

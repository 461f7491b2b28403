import sys, time
sys.path.insert(0, '.')
import numpy as np
import chd_amd
from chd_amd.synth import make_walk
from oracle.oracle import OracleProblem
F = int(sys.argv[1]); seed = int(sys.argv[2])
seq = make_walk(seed=seed, F=F, randomize=True)
p = OracleProblem(seq)
tot = 0; t00 = time.time()
for st in [0, 1, 2, 3, 4]:
    t0 = time.time()
    status, info = p.solve_stage(st, 300)
    tot += info['iters']
    print(f'== stage {st}: status {status} iters {info["iters"]} f={info["objective"]:.6e} E={info["kkt_error"]:.2e} viol={info["constr_viol"]:.1e} nfact={info["n_factor"]} N={info["N"]} w={info["bandwidth"]} time {time.time()-t0:.2f}s')
    if st == 4 and status != 0:
        status, info = p.solve_stage(5, 300)
        print(f'== stage 5: status {status} iters {info["iters"]} f={info["objective"]:.6e} E={info["kkt_error"]:.2e}')
print('total iters', tot, 'time', time.time() - t00)

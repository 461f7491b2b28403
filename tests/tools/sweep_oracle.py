import sys, time
sys.path.insert(0, '.')
import numpy as np
import chd_amd
from chd_amd.synth import make_walk
from oracle.oracle import OracleProblem
F = int(sys.argv[1]); seeds = range(int(sys.argv[2]), int(sys.argv[3]))
tot_it = 0; t00 = time.time()
for seed in seeds:
    seq = make_walk(seed=seed, F=F, randomize=True)
    p = OracleProblem(seq)
    line = []
    its = 0
    for st in [0, 1, 2, 3, 4]:
        status, info = p.solve_stage(st, 300)
        line.append(f'{status}/{info["iters"]}')
        its += info['iters']
        if st == 4 and status != 0:
            status, info = p.solve_stage(5, 300)
            line.append(f'{status}/{info["iters"]}')
            its += info['iters']
    tot_it += its
    print(seed, ' '.join(line), 'total', its, flush=True)
print('mean iters', tot_it / len(seeds), 'time/seq', (time.time() - t00) / len(seeds))

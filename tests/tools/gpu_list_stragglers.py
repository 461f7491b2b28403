"""Which sequences of the bench workload are the stragglers?  Prints, for seeds 0 .. n-1 (90 frames, reference caps), the ones whose
stage 3 took >= 100 iterations, failed, or was ended by the stall guard.    python tests/tools/gpu_list_stragglers.py [n] [stall_window]"""
import sys
sys.path.insert(0, '.')
import chd_amd  # noqa: E402,F401
import bench  # noqa: E402
from chd_amd.phys_optim import PhysOptim, default_config  # noqa: E402
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1536
win = int(sys.argv[2]) if len(sys.argv) > 2 else 150
seqs = bench.make_sequences(0, n, 8)
s = PhysOptim(0, default_config(stall_window=win))
res, st = s.solve(seqs)
print('kernel ms', st['kernel_ms'], 'stalled', st['n_stalled'], 'fallbacks', st['n_fallback'])
for i, r in enumerate(res):
    if r.stage_iters[4] >= 100 or r.stage_status[4] != 0:
        print(i, list(zip(r.stage_status, r.stage_iters)), 'stalled', r.stage_stalled[4])

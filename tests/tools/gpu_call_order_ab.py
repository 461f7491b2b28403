import os, sys, time, json
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT','/root/repo'))
import bench
seqs = bench.make_sequences(0, 2560, 8, 90)
import chd_amd
from chd_amd.phys_optim import PhysOptim, default_config
out = {}
for tag, env in (('sorted', None), ('unsorted', '1'), ('sorted2', None), ('unsorted2', '1')):
    if env: os.environ['CHD_CALL_ORDER_OFF'] = env
    else: os.environ.pop('CHD_CALL_ORDER_OFF', None)
    s = PhysOptim(device=0, config=default_config())
    s.solve_batch(seqs[:600])
    ts = []
    for r in range(3):
        res, cs = s.solve_batch(seqs); ts.append(cs['wall_ms'])
    s.close()
    out[tag] = ts
    print(tag, [round(t) for t in ts], flush=True)
# mixed lengths: 1280 x 60 + 1280 x 120 frames interleaved
mixed = [bench._gen((5000 + i, 1, 60 if i % 2 else 120))[0] for i in range(800)]
for tag, env in (('mixed sorted', None), ('mixed unsorted', '1')):
    if env: os.environ['CHD_CALL_ORDER_OFF'] = env
    else: os.environ.pop('CHD_CALL_ORDER_OFF', None)
    s = PhysOptim(device=0, config=default_config())
    s.solve_batch(mixed[:300])
    ts = []
    for r in range(2):
        res, cs = s.solve_batch(mixed); ts.append(cs['wall_ms'])
    s.close()
    print(tag, [round(t) for t in ts], flush=True)

import os, sys, json, shutil, glob
import numpy as np
ROOT = os.environ.get('GRAFT_REPO_ROOT', '/root/repo')
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests')); sys.path.insert(0, os.path.join(ROOT, 'tests', 'tools'))
import pipeline_bench as pb
keep = '/tmp/pipe_keep'
shutil.rmtree(keep, ignore_errors=True); os.makedirs(keep)
r = pb.run(32, 60, keep=keep)
print(json.dumps({k: r[k] for k in ('videos_per_s', 'seconds_per_stage')}))
import chd_amd
from chd_amd import io_formats as iof
from chd_amd.phys_optim import PhysOptim, default_config
out = os.path.join(ROOT, 'gpurun_out', 'pipe'); shutil.rmtree(out, ignore_errors=True); os.makedirs(out)
seqs = []; names = []
for d in sorted(glob.glob(os.path.join(keep, 'data', 'video_*'))):
    src = os.path.join(d, 'phys_optim_in_combined')
    if os.path.isdir(src):
        shutil.copytree(src, os.path.join(out, os.path.basename(d)))
        seqs.append(iof.read_inputs(src, 60)); names.append(os.path.basename(d))
s = PhysOptim(device=0, config=default_config())
b = s.upload(seqs); st = b.solve(); res = b.fetch()
for n, q in zip(names, res):
    print(n, list(q.stage_status), list(q.stage_iters), list(q.stage_factorizations))
print('slowest ms', st['max_seq_ms'])
b.free(); s.close()

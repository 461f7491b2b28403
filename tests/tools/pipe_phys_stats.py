"""Per-video solver statistics of the physics stage on the inputs the pipeline produced (tests/tools/pipeline_bench.py --keep DIR): which clips are the
slow ones, and their phys_optim_in_combined/ directories copied out for CPU analysis (tests/tools/emu_sweep.py --dirs).

    python tests/tools/pipe_phys_stats.py DATA_ROOT FRAMES OUT_DIR [n_copy]
"""
import json
import os
import shutil
import sys
import time

sys.path.insert(0, '.')
import chd_amd  # noqa: E402,F401
from chd_amd import io_formats as iof  # noqa: E402
from chd_amd.phys_optim import PhysOptim, default_config  # noqa: E402

root, frames, out = sys.argv[1], int(sys.argv[2]), sys.argv[3]
n_copy = int(sys.argv[4]) if len(sys.argv) > 4 else 12
vids = sorted(v for v in os.listdir(root) if os.path.isdir(os.path.join(root, v, 'phys_optim_in_combined')))
seqs = [iof.read_inputs(os.path.join(root, v, 'phys_optim_in_combined'), frames) for v in vids]
s = PhysOptim(0, default_config())
t0 = time.time(); res, st = s.solve(seqs); dt = time.time() - t0
rows = sorted(((sum(r.stage_iters), v, list(r.stage_status), list(r.stage_iters)) for v, r in zip(vids, res)), reverse=True)
os.makedirs(out, exist_ok=True)
json.dump({'videos': len(vids), 'frames': frames, 'seconds_upload_solve_fetch': dt, 'kernel_ms': st['kernel_ms'], 'max_seq_ms': st['max_seq_ms'], 'total_iters': st['total_iters'],
           'per_video': [{'video': v, 'iterations': it, 'stage_status': ss, 'stage_iters': si} for it, v, ss, si in rows]}, open(os.path.join(out, 'pipe_phys_stats.json'), 'w'), indent=1)
for it, v, ss, si in rows[:n_copy]:
    shutil.copytree(os.path.join(root, v, 'phys_optim_in_combined'), os.path.join(out, v), dirs_exist_ok=True)
print('physics on %d clips x %d frames: %.2f s; slowest: %s' % (len(vids), frames, dt, [(v, it) for it, v, _, _ in rows[:6]]))

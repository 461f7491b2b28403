"""A pipelined call under rocprofv3's marker trace: shows that the library's roctx ranges (table build, upload + launch, finisher, fetch) land on the marker track.

    cd /tmp && rocprofv3 --marker-trace --kernel-trace --stats --output-format csv -d <out> -o roctx -- python tests/tools/gpu_roctx_probe.py
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import chd_amd  # noqa: E402,F401
from chd_amd.phys_optim import PhysOptim, default_config  # noqa: E402
from chd_amd.synth import make_walk  # noqa: E402

seqs = [make_walk(seed=100 + i, F=40, randomize=True) for i in range(700)]
s = PhysOptim(device=0, config=default_config())
res, cs = s.solve_batch(seqs)
b = s.upload(seqs[:64]); b.solve(); b.free()
s.close()
print('solved %d sequences in %d chunks, %d iterations' % (cs['n_sequences'], cs['n_chunks'], cs['total_iters']))

"""prepare_input for a whole run on one MI355X: N copies (perturbed) of the fixture clip through the NumPy mirror, video by video, and
through prepare_sequences_device (one batch of tensor operations on the GPU).

    python tests/tools/prepare_bench.py [clips=4000] [frames=90]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests', 'golden'))
import chd_amd  # noqa: E402,F401
from chd_amd import apply_results as ar  # noqa: E402
from chd_amd import prepare_input as pi  # noqa: E402
from make_apply_golden import CHARACTER, synthetic_motion  # noqa: E402

if __name__ == '__main__':
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 4000
    F = int(sys.argv[2]) if len(sys.argv) > 2 else 90
    ch = ar.Character(**CHARACTER)
    rng = np.random.default_rng(0)
    base = [synthetic_motion(F, seed=s) for s in range(16)]
    clips = [base[i % 16] for i in range(N)]
    fc = np.zeros((F, 4), dtype=int); fc[:F // 2, :2] = 1; fc[F // 2 - 2:, 2:] = 1
    floor = (np.array([0.0, 0.0, 1.0]), np.zeros(3))
    t0 = time.perf_counter()
    ref = [pi.prepare_sequence(m, floor, fc, ch, 0, F, 1.0 / 30.0) for m in clips[:200]]
    t_np = (time.perf_counter() - t0) / 200
    pi.prepare_sequences_device(clips[:8], [floor] * 8, [fc] * 8, ch, device='cuda:0')          # warm-up
    t0 = time.perf_counter()
    got = pi.prepare_sequences_device(clips, [floor] * N, [fc] * N, ch, device='cuda:0')
    t_dev = time.perf_counter() - t0
    err = max(np.abs(got[i].inertia - ref[i].inertia).max() + np.abs(got[i].com - ref[i].com).max() for i in range(200))
    print('prepare_input, %d clips x %d frames, %d joints: NumPy %.2f ms per clip (%.1f s for the run), batched on the GPU %.2f s for the run (%.3f ms per clip, '
          'host packing and SeqInput assembly included); max |difference| %.1e' % (N, F, clips[0].n_joints, 1e3 * t_np, t_np * N, t_dev, 1e3 * t_dev / N, err))

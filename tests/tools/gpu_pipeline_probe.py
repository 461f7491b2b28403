"""The whole call against the solve alone on the bench workload: chd_phys_solve_batch (set-up + upload + solve + fetch, pipelined over chunks)
for several chunk sizes, with the per-chunk timeline of the library (CHD_PIPE_TRACE), next to upload-then-solve through the split interface.

    python tests/tools/gpu_pipeline_probe.py [n_sequences] [chunk sizes ...]
"""
import os
import sys
import time

sys.path.insert(0, '.')
os.environ['CHD_PIPE_TRACE'] = '1'
import bench  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 2560
chunks = [int(v) for v in sys.argv[2:]] or [0, 256, 512, -1]
seqs = bench.make_sequences(0, n, 8)
import chd_amd  # noqa: E402,F401
from chd_amd.phys_optim import PhysOptim, default_config  # noqa: E402

s = PhysOptim(0, default_config())
wb = s.upload(seqs[:256]); wb.solve(); wb.free()
t0 = time.perf_counter(); b = s.upload(seqs); t1 = time.perf_counter(); st = b.solve(); t2 = time.perf_counter(); res = b.fetch(); t3 = time.perf_counter(); b.free()
print('split interface: upload (table build + copies) %.2f s, solve %.2f s = %.1f sequences/s, fetch %.2f s; all three %.1f sequences/s' % (t1 - t0, t2 - t1, n / (t2 - t1), t3 - t2, n / (t3 - t0)), flush=True)
s.close()
for ch in chunks:
    s = PhysOptim(0, default_config(pipeline_chunk=ch))
    os.environ.pop('CHD_PIPE_TRACE', None)
    s.solve_batch(seqs)                            # warm-up with the same plan: kernel load, pools
    os.environ['CHD_PIPE_TRACE'] = '1'
    t0 = time.perf_counter(); r2, cs = s.solve_batch(seqs); dt = time.perf_counter() - t0
    same = all(a.stage_iters == b_.stage_iters for a, b_ in zip(res, r2))
    print('chd_phys_solve_batch, chunk %d: %.2f s in Python, %.2f s in the library = %.1f sequences/s (%.2f of solve-only); chunks %d, set-up %.0f ms wall on %d threads (%.2f ms per sequence and thread), upload %.0f ms, waited for the device %.0f ms; same results %s'
          % (ch, dt, cs['wall_ms'] * 1e-3, n / (cs['wall_ms'] * 1e-3), (n / (cs['wall_ms'] * 1e-3)) / (n / (t2 - t1)), cs['n_chunks'], cs['setup_wall_ms'], cs['host_threads'], cs['setup_cpu_ms'] / n, cs['upload_ms'], cs['wait_for_pool_ms'], same), flush=True)
    s.close()

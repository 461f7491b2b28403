"""File-to-file throughput of the drop-in call ("report an I/O-inclusive number separately"): N directories with
the four input files each -> chd_phys_solve_dirs -> three solution files + success_log per directory, ONE call.

    python tests/tools/gpu_dirs_bench.py [n_dirs] [frames]
"""
import os
import sys
import tempfile
import time
from concurrent.futures import ProcessPoolExecutor
sys.path.insert(0, '.')
import chd_amd  # noqa: E402,F401
from chd_amd import io_formats as iof  # noqa: E402
from chd_amd.synth import make_walk  # noqa: E402


def make(args):
    root, i, F = args
    d = os.path.join(root, 'v%05d' % i, 'phys_optim_in_ybot')
    iof.write_inputs(make_walk(seed=i, F=F, randomize=True), d)
    o = os.path.join(root, 'v%05d' % i, 'phys_optim_out_ybot')
    os.makedirs(o)
    return d, o


if __name__ == '__main__':
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
    F = int(sys.argv[2]) if len(sys.argv) > 2 else 90
    root = tempfile.mkdtemp(prefix='chd_dirs_')
    t0 = time.time()
    with ProcessPoolExecutor(8) as ex:                     # (before the HIP runtime exists in this process)
        dirs = list(ex.map(make, [(root, i, F) for i in range(n)], chunksize=16))
    t1 = time.time()
    from chd_amd.phys_optim import PhysOptim, default_config
    s = PhysOptim(0, default_config(stall_window=150))
    s.solve_dirs([dirs[0][0]], [dirs[0][1]], [F])          # warm-up: kernel load, workspace allocation
    t2 = time.time()
    st = s.solve_dirs([d[0] for d in dirs], [d[1] for d in dirs], [F] * n)
    t3 = time.time()
    seqs = [iof.read_inputs(d[0], F) for d in dirs[:256]]
    t4 = time.time()
    b = s.upload(seqs); t5 = time.time(); b.solve(); t6 = time.time(); b.fetch(); t7 = time.time()
    print('%d directories of %d frames written in %.1f s' % (n, F, t1 - t0))
    print('chd_phys_solve_dirs: %.2f s = %.1f directories/s (read + table build + upload + solve + fetch + write), %d failures' % (t3 - t2, n / (t3 - t2), sum(1 for x in st if x != 0)))
    print('for scale: 256 of them in memory: upload %.2f s, solve %.2f s, fetch %.2f s' % (t5 - t4, t6 - t5, t7 - t6))
    sz = sum(os.path.getsize(os.path.join(dirs[0][1], f)) for f in os.listdir(dirs[0][1]))
    print('output files per directory: %s (%d bytes)' % (sorted(os.listdir(dirs[0][1])), sz))

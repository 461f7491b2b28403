"""Per-sequence solver statistics of the bench workload on the GPU (analysis tool, not a test).

    python tests/tools/gpu_seq_stats.py --n 2560 --out gpurun_out/seq_stats.npz [--stall-window 0 150]

For every stall-guard setting: stage statuses / iteration counts / factorisations / guard flags / final optimality errors of
seeds 0 .. n-1 (bench.py's sequences), the wall time of the solve, and the library's aggregate statistics.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--n', type=int, default=2560)
    ap.add_argument('--seed0', type=int, default=0)
    ap.add_argument('--out', default='gpurun_out/seq_stats.npz')
    ap.add_argument('--stall-window', type=int, nargs='+', default=[150, 0])
    args = ap.parse_args()
    seqs = bench.make_sequences(args.seed0, args.n, 8)
    import chd_amd  # noqa: F401
    from chd_amd.phys_optim import PhysOptim, default_config
    out = {}
    summary = {}
    for sw in args.stall_window:
        solver = PhysOptim(device=0, config=default_config(stall_window=sw))
        wb = solver.upload(seqs[:128]); wb.solve(); wb.free()
        b = solver.upload(seqs)
        t0 = time.perf_counter(); st = b.solve(); dt = time.perf_counter() - t0
        res = b.fetch()
        k = 'sw%d_' % sw
        out[k + 'status'] = np.array([r.stage_status for r in res])
        out[k + 'iters'] = np.array([r.stage_iters for r in res])
        out[k + 'stalled'] = np.array([r.stage_stalled for r in res])
        out[k + 'nfact'] = np.array([r.stage_factorizations for r in res])
        out[k + 'kkt'] = np.array([r.stage_kkt_error for r in res])
        out[k + 'viol'] = np.array([r.stage_constr_viol for r in res])
        out[k + 'obj'] = np.array([r.stage_objective for r in res])
        out[k + 'ok'] = np.array([[r.dynamics_succeed, r.durations_succeed] for r in res])
        summary[k] = dict(seconds=dt, seq_per_s=args.n / dt, stats={kk: (vv if not isinstance(vv, list) else [float(x) for x in vv]) for kk, vv in st.items()})
        print(k, 'seconds %.2f  seq/s %.1f  iters %d  fallbacks %d stalled %d  max_seq_ms %.0f' % (dt, args.n / dt, st['total_iters'], st['n_fallback'], st['n_stalled'], st['max_seq_ms']), flush=True)
        b.free(); solver.close()
    os.makedirs(os.path.dirname(args.out) or '.', exist_ok=True)
    np.savez_compressed(args.out, **out)
    json.dump(summary, open(os.path.splitext(args.out)[0] + '.json', 'w'), indent=1)


if __name__ == '__main__':
    main()

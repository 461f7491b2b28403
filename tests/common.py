"""Helpers shared by the parity tests."""
import numpy as np


def rel_l2(a, b):
    a = np.asarray(a, dtype=np.float64); b = np.asarray(b, dtype=np.float64)
    return float(np.linalg.norm(a - b) / max(1e-300, np.linalg.norm(b)))


def rel_max(a, b):
    a = np.asarray(a, dtype=np.float64); b = np.asarray(b, dtype=np.float64)
    return float(np.abs(a - b).max() / max(1e-300, np.abs(b).max()))


def oracle_run(seq, max_iter, **kw):
    """Run the staged solve on the CPU oracle; returns (stage stats, three snapshots)."""
    from oracle.oracle import OracleProblem
    o = OracleProblem(seq, max_iter=max_iter, **kw)
    stats, snaps = [], []
    for st in range(5):
        status, info = o.solve_stage(st)
        stats.append((status, info['iters'], info['objective']))
        if st in (1, 3, 4):
            snaps.append(o.sample_solution())
    if stats[4][0] != 0:       # "STAGE 4: Durations failed ..." phys_optim.cpp:714
        status, info = o.solve_stage(5)
        stats.append((status, info['iters'], info['objective']))
        snaps[2] = o.sample_solution()
    return stats, snaps


SNAP_KEYS = (('base_lin', 'base_lin'), ('base_ang_deg', 'base_ang_deg'), ('ee_pos', 'ee_pos'), ('ee_force', 'ee_force'))


def snapshot_errors(sol, ref):
    """rel-L2 of COM linear / angular, feet and GRFs of a Solution against an oracle snapshot dict."""
    out = {}
    for a, b in SNAP_KEYS:
        out[a] = rel_l2(getattr(sol, a), ref[b])
    out['contact_mismatch'] = int(np.abs(np.asarray(sol.contact, dtype=np.int64) - ref['contact']).sum())
    out['n_samples'] = (sol.base_lin.shape[0], ref['n_samples'])
    return out

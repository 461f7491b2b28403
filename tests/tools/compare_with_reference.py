"""The route by which parity against the REFERENCE solver can become "measured" (SURVEY 8c / 8d-ii).

    python tests/tools/compare_with_reference.py --towr_phys_optim_path <dir holding the reference's phys_optim> [--n 8] [--frames 90]

If `<dir>/phys_optim` is a foreign binary (the reference's towr_phys_optim build: TOWR fork + ifopt + IPOPT/MA57 -- not this repo's
cli/phys_optim, which links libchd_phys.so), it is run exactly as scripts/run_phys_mocap.py:159-174 runs it -- one process per
sequence, `--in_dir --nframes --out_dir --w_*` with the reference's default weights, pinned to one core -- on the first n
sequences of bench.py's workload, written in the reference's own input format (io_formats.write_inputs); its three solution files
are parsed with the reference's line-indexed layout (io_formats.load_results) and compared with the HIP path's results for the same
sequences: relative L2 per trajectory (COM, Euler angles, four feet, four ground-reaction forces) per snapshot, and seconds per
sequence.  No such binary can be built in the build container (towr_phys_optim/CMakeLists.txt:4-8: no TOWR fork, ifopt, IPOPT,
HSL, Eigen, gflags, no network), so the committed state of this comparison is "not found -- not measured".

Used by bench.py (`reference_binary` block of its JSON line; `cpu_baseline.kind` becomes "reference" when the binary is found).
"""
import argparse
import json
import os
import shutil
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

OWN_MARK = b'libchd_phys'          # this repo's cli/phys_optim links the HIP library; the reference's binary does not


def find_reference_binary(path):
    """-> (binary path or None, reason)."""
    if not path:
        return None, 'no --towr_phys_optim_path given'
    exe = os.path.join(path, 'phys_optim')
    if not (os.path.isfile(exe) and os.access(exe, os.X_OK)):
        return None, 'no executable %s' % exe
    try:
        with open(exe, 'rb') as fh:
            if OWN_MARK in fh.read():
                return None, '%s is this repository\'s own command-line front end of libchd_phys.so, not the reference solver' % exe
    except OSError as exc:
        return None, str(exc)
    return exe, 'found'


def run_reference(exe, seqs, workdir=None, pin_core=0, timeout=3600):
    """One child process per sequence, as the reference's driver does.  -> list of (three Solutions or None, seconds, return code)."""
    import chd_amd  # noqa: F401
    from chd_amd import io_formats as iof
    from chd_amd.phys_optim import SNAPSHOT_FILES
    tmp = workdir or tempfile.mkdtemp(prefix='chd_ref_')
    out = []
    for i, seq in enumerate(seqs):
        din = os.path.join(tmp, 'in_%d' % i); dout = os.path.join(tmp, 'out_%d' % i)
        os.makedirs(din, exist_ok=True); os.makedirs(dout, exist_ok=True)
        iof.write_inputs(seq, din)
        cmd = [exe, '--in_dir', din, '--nframes', str(seq.F), '--out_dir', dout, '--w_com_lin', '0.4', '--w_com_ang', '1.7', '--w_ee', '0.3', '--w_smooth', '0.1', '--w_dur', '0.1']
        if shutil.which('taskset'):
            cmd = ['taskset', '-c', str(pin_core)] + cmd
        t0 = time.perf_counter()
        try:
            rc = subprocess.run(cmd, cwd=os.path.dirname(exe), stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=timeout).returncode
        except subprocess.TimeoutExpired:
            rc = -9
        dt = time.perf_counter() - t0
        sols = []
        for f in SNAPSHOT_FILES:
            pth = os.path.join(dout, f)
            try:
                sols.append(iof.load_results(pth) if os.path.exists(pth) else None)
            except Exception:
                sols.append(None)
        out.append((sols, dt, rc))
    if workdir is None:
        shutil.rmtree(tmp, ignore_errors=True)
    return out


def compare(ref_runs, hip_results):
    """rel-L2 of every trajectory of every snapshot, HIP result against the reference's files."""
    rows = []
    for i, ((sols, dt, rc), r) in enumerate(zip(ref_runs, hip_results)):
        rec = {'sequence': i, 'reference_seconds': dt, 'reference_return_code': rc, 'snapshots': []}
        for k, sol in enumerate(sols):
            if sol is None:
                rec['snapshots'].append(None); continue
            h = r.snapshots[k]
            e = {}
            for name in ('base_lin', 'base_ang_deg', 'ee_pos', 'ee_force'):
                a = np.asarray(getattr(h, name)); b = np.asarray(getattr(sol, name))
                e[name] = float(np.linalg.norm(a - b) / max(1e-300, np.linalg.norm(b))) if a.shape == b.shape else None
            e['contacts_equal'] = bool(np.array_equal(np.asarray(h.contact), np.asarray(sol.contact)))
            rec['snapshots'].append(e)
        rows.append(rec)
    return rows


def reference_block(path, seqs, hip_results, n=8):
    """What bench.py prints as `reference_binary`."""
    exe, why = find_reference_binary(path)
    if exe is None:
        return {'status': 'not found -- not measured', 'reason': why,
                'note': 'parity against IPOPT stays unpinned until a box with the reference toolchain runs tests/tools/compare_with_reference.py'}, None
    runs = run_reference(exe, seqs[:n])
    rows = compare(runs, hip_results[:n])
    worst = {}
    for rec in rows:
        for e in rec['snapshots']:
            if e:
                for k, v in e.items():
                    if isinstance(v, float):
                        worst[k] = max(worst.get(k, 0.0), v)
    secs = [r[1] for r in runs]
    return ({'status': 'measured', 'binary': exe, 'sequences': len(runs), 'worst_rel_l2': worst, 'per_sequence': rows},
            {'value': len(secs) / sum(secs), 'unit': 'sequences/s', 'cores': 1, 'kind': 'reference',
             'sample': 'first %d sequences of the workload, one %s process per sequence pinned to one core (scripts/run_phys_mocap.py:159-174), %.1f s' % (len(secs), exe, sum(secs))})


if __name__ == '__main__':
    ap = argparse.ArgumentParser()
    ap.add_argument('--towr_phys_optim_path', required=True)
    ap.add_argument('--n', type=int, default=8)
    ap.add_argument('--frames', type=int, default=90)
    a = ap.parse_args()
    import chd_amd  # noqa: F401
    from chd_amd.synth import make_walk
    seqs = [make_walk(seed=s, F=a.frames, randomize=True) for s in range(a.n)]
    exe, why = find_reference_binary(a.towr_phys_optim_path)
    if exe is None:
        print(json.dumps({'status': 'not found -- not measured', 'reason': why}))
        sys.exit(0)
    from chd_amd.phys_optim import PhysOptim, default_config
    s = PhysOptim(device=0, config=default_config())
    res, _ = s.solve(seqs); s.close()
    blk, base = reference_block(a.towr_phys_optim_path, seqs, res, a.n)
    print(json.dumps({'reference_binary': blk, 'cpu_baseline': base}, indent=1))

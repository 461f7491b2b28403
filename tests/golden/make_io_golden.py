"""Generates tests/golden/io_golden.npz: a solution file written by this repo's writer (io_formats.write_solution, the
format of phys_optim.cpp:63-143) as parsed by the REFERENCE's own reader, `load_results` of
/root/reference/src/utils/towr_utils.py:51-121 (line-number indexed; swaps y/z, negates, converts the Euler angles).

The reference module cannot be imported as a whole (matplotlib, BVH, Animation ...), so the two definitions it takes
(`TowrResults`, `load_results`, `find_contact_durations`) are cut out of its source with `ast` and executed with the reference's own
`Quaternions` class (src/skeleton_fitting/ik/Quaternions.py).  Run in the build container only (it reads
/root/reference); the test (tests/test_io_formats.py) uses the committed fixture."""
import ast
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
import chd_amd  # noqa: E402,F401
from chd_amd import io_formats as iof  # noqa: E402

REF = '/root/reference/src'


def reference_loader():
    sys.path.insert(0, os.path.join(REF, 'skeleton_fitting', 'ik'))
    from Quaternions import Quaternions
    src = open(os.path.join(REF, 'utils', 'towr_utils.py')).read()
    tree = ast.parse(src)
    keep = [n for n in tree.body if (isinstance(n, ast.ClassDef) and n.name == 'TowrResults')
            or (isinstance(n, ast.FunctionDef) and n.name in ('load_results', 'find_contact_durations'))]
    assert len(keep) == 3
    ns = {'np': np, 'os': os, 'Quaternions': Quaternions}
    exec(compile(ast.Module(body=keep, type_ignores=[]), 'towr_utils.py', 'exec'), ns)
    return ns['load_results'], ns['find_contact_durations']


if __name__ == '__main__':
    S = 9
    rng = np.random.default_rng(11)
    sol = iof.Solution(dt=1 / 30, num_frames=S, base_lin=rng.normal(size=(S, 3)), base_ang_deg=rng.uniform(-40, 40, size=(S, 3)),
                       ee_pos=rng.normal(size=(4, S, 3)), ee_force=rng.normal(size=(4, S, 3)) * 300, contact=rng.integers(0, 2, (4, S)))
    path = '/tmp/io_golden_sol.txt'
    iof.write_solution(sol, path)
    load, ref_durations = reference_loader()
    out = {'file_text': np.frombuffer(open(path, 'rb').read(), dtype=np.uint8)}
    for flip in (True, False):
        r = load(path, flip_coords=flip)
        tag = 'flip' if flip else 'noflip'
        out[tag + '_base_pos'] = r.base_pos; out[tag + '_base_rot'] = r.base_rot; out[tag + '_base_R'] = r.base_R
        out[tag + '_feet_pos'] = r.feet_pos; out[tag + '_feet_force'] = r.feet_force; out[tag + '_feet_contact'] = r.feet_contact
        assert r.num_feet == 4 and abs(r.dt - 1 / 30) < 1e-9
    for k in ('base_lin', 'base_ang_deg', 'ee_pos', 'ee_force', 'contact'):
        out['in_' + k] = np.asarray(getattr(sol, k))
    # contact flags -> phase durations (towr_utils.py:435-449): float accumulation included, so the values are bit-exact
    for case, (n, dt) in enumerate(((90, 1 / 30), (61, 0.04), (600, 1 / 30))):
        c = (rng.random(n) < 0.5).astype(np.int64)
        for k in range(2, n, 7):
            c[k:k + 4] = c[k]               # runs, not noise
        out['dur%d_contacts' % case] = c; out['dur%d_dt' % case] = np.array(dt)
        out['dur%d_ref' % case] = np.array(ref_durations(c, dt))
    np.savez_compressed(os.path.join(HERE, 'io_golden.npz'), **out)
    print('wrote io_golden.npz', {k: v.shape for k, v in out.items()})

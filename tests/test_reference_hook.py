"""The reference-binary hook (tests/tools/compare_with_reference.py, bench.py's `reference_binary` block): detection rules, and the
whole run / parse / compare path against a STAND-IN binary -- a `phys_optim` script with the reference's command line that solves with the
CPU oracle and writes the reference's file formats.  (The real reference binary cannot be built here: SURVEY 8c.)"""
import os
import stat
import sys
import types

import numpy as np

import chd_amd  # noqa: F401
from chd_amd.synth import make_walk

from common import oracle_run

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, os.path.join(HERE, 'tools'))
import compare_with_reference as cwr  # noqa: E402

STUB = '''#!%(py)s
import sys
sys.path.insert(0, %(root)r); sys.path.insert(0, %(tests)r)
import chd_amd
from chd_amd import io_formats as iof
from chd_amd.phys_optim import SNAPSHOT_FILES
from common import oracle_run
a = dict(zip(sys.argv[1::2], sys.argv[2::2]))
seq = iof.read_inputs(a['--in_dir'], int(a['--nframes']))
stats, snaps = oracle_run(seq, [300] * 6)
for f, sn in zip(SNAPSHOT_FILES, snaps):
    iof.write_solution(iof.Solution(dt=seq.dt, num_frames=sn['num_frames'], base_lin=sn['base_lin'], base_ang_deg=sn['base_ang_deg'], ee_pos=sn['ee_pos'],
                                    ee_force=sn['ee_force'], contact=sn['contact']), a['--out_dir'] + '/' + f)
'''


def test_detection_rules(tmp_path):
    assert cwr.find_reference_binary('')[0] is None
    assert cwr.find_reference_binary(str(tmp_path))[0] is None                       # no binary there
    own = os.path.join(ROOT, 'contact-human-dynamics_amd', 'cli')
    if os.path.exists(os.path.join(own, 'phys_optim')):
        exe, why = cwr.find_reference_binary(own)                                       # this repository's front end is not a reference
        assert exe is None and 'own' in why
    blk, base = cwr.reference_block('', [], [])
    assert blk['status'].startswith('not found') and base is None


def test_run_parse_compare_with_a_stand_in_binary(tmp_path, oracle_lib):
    exe = tmp_path / 'phys_optim'
    exe.write_text(STUB % dict(py=sys.executable, root=ROOT, tests=HERE))
    exe.chmod(exe.stat().st_mode | stat.S_IXUSR)
    assert cwr.find_reference_binary(str(tmp_path))[0] == str(exe)
    seq = make_walk(seed=2, F=40, randomize=True)
    _, snaps = oracle_run(seq, [300] * 6)
    hip_like = types.SimpleNamespace(snapshots=[types.SimpleNamespace(base_lin=s['base_lin'], base_ang_deg=s['base_ang_deg'], ee_pos=s['ee_pos'], ee_force=s['ee_force'],
                                                                      contact=s['contact']) for s in snaps])
    blk, base = cwr.reference_block(str(tmp_path), [seq], [hip_like], n=1)
    assert blk['status'] == 'measured' and blk['sequences'] == 1
    assert base['kind'] == 'reference' and base['cores'] == 1 and base['value'] > 0
    for e in blk['per_sequence'][0]['snapshots']:
        assert e['contacts_equal']
        assert max(e['base_lin'], e['base_ang_deg'], e['ee_pos'], e['ee_force']) < 1e-8          # the files carry 10 significant digits

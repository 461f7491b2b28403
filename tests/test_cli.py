"""The `phys_optim` executable (contact-human-dynamics_amd/cli): the reference's command line for one video
(towr_phys_optim/phys_optim.cpp:23-31), so that the unmodified scripts/run_phys_mocap.py --towr_phys_optim_path <cli dir>
keeps working (it runs ['./phys_optim', '--in_dir', ..., '--nframes', ..., '--out_dir', ..., '--w_com_lin', ...] from that
directory, run_phys_mocap.py:159-174)."""
import json
import os
import subprocess

import numpy as np
import pytest
import torch

import chd_amd  # noqa: F401
from chd_amd import phys_optim


@pytest.fixture(scope='module')
def cli():
    return phys_optim.build_cli()


def reference_argv(in_dir, nframes, out_dir):
    # the argument vector of run_phys_mocap.py:163-174 with PhysOptimParsms' defaults (:33-44)
    return ['./phys_optim', '--in_dir', in_dir, '--nframes', str(nframes), '--out_dir', out_dir, '--w_com_lin', str(0.4), '--w_com_ang', str(1.7),
            '--w_ee', str(0.3), '--w_smooth', str(0.1), '--w_dur', str(0.1)]


def test_accepts_the_reference_argument_vector(cli):
    cwd = os.path.dirname(cli)
    out = subprocess.check_output(reference_argv('/data/v/phys_optim_in_ybot', 123, '/data/v/phys_optim_out_ybot') + ['--check_args'], cwd=cwd, text=True)
    a = json.loads(out)
    assert a['in_dir'] == '/data/v/phys_optim_in_ybot' and a['out_dir'] == '/data/v/phys_optim_out_ybot' and a['nframes'] == 123
    assert (a['w_com_lin'], a['w_com_ang'], a['w_ee'], a['w_smooth'], a['w_dur']) == (0.4, 1.7, 0.3, 0.1, 0.1)
    # gflags forms: --flag=value and single dash; defaults of phys_optim.cpp:23-31 when a flag is absent
    a = json.loads(subprocess.check_output(['./phys_optim', '-nframes=60', '--w_ee=0.5', '--check_args'], cwd=cwd, text=True))
    assert a['nframes'] == 60 and a['w_ee'] == 0.5 and a['in_dir'] == './' and a['out_dir'] == 'sol_out' and a['w_com_ang'] == 1.7
    assert subprocess.run(['./phys_optim', '--no_such_flag', '1'], cwd=cwd, capture_output=True).returncode == 1
    assert subprocess.run(['./phys_optim', '--nframes', 'abc'], cwd=cwd, capture_output=True).returncode == 1


@pytest.mark.skipif(torch.cuda.is_available(), reason='checks the no-GPU failure mode')
def test_fails_loudly_without_gpu(cli, tmp_path):
    r = subprocess.run(reference_argv(str(tmp_path), 40, str(tmp_path)), cwd=os.path.dirname(cli), capture_output=True, text=True)
    assert r.returncode == 2 and 'no CPU path' in r.stderr


@pytest.mark.gpu
def test_cli_writes_the_same_files_as_the_batched_call(cli, tmp_path):
    from chd_amd import io_formats as iof
    from chd_amd.synth import make_walk
    seq = make_walk(seed=2, F=40, randomize=True)
    din = str(tmp_path / 'phys_optim_in_ybot'); d1 = str(tmp_path / 'out_cli'); d2 = str(tmp_path / 'out_lib')
    iof.write_inputs(seq, din)
    os.makedirs(d1); os.makedirs(d2)
    r = subprocess.run(reference_argv(din, seq.F, d1), cwd=os.path.dirname(cli), capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    s = phys_optim.PhysOptim(device=0)
    assert s.solve_dirs([din], [d2], [seq.F]) == [0]
    s.close()
    names = ['sol_out_durations.txt', 'sol_out_dynamics.txt', 'sol_out_no_dynamics.txt', 'success_log.txt']
    assert sorted(os.listdir(d1)) == names
    for n in names:
        assert open(os.path.join(d1, n)).read() == open(os.path.join(d2, n)).read(), n
    sol = iof.load_results(os.path.join(d1, 'sol_out_dynamics.txt'))
    assert sol.num_frames == seq.F and np.isfinite(sol.ee_force).all()

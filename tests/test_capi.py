"""The C-ABI shared library: loads without a GPU, exports every function include/chd_phys.h declares,
refuses to create a handle when no HIP device exists (no silent CPU fallback), and the Python structs
mirror the C layout."""
import ctypes as C
import os
import re

import numpy as np
import pytest
import torch

import chd_amd
from chd_amd import phys_capi, phys_optim

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope='module')
def lib():
    phys_optim.build_library()
    return C.CDLL(phys_optim.LIB_PATH)


def declared_functions():
    src = open(os.path.join(ROOT, 'include', 'chd_phys.h')).read()
    src = re.sub(r'/\*.*?\*/', '', src, flags=re.S)
    return sorted(set(re.findall(r'\b(chd_[a-z_]+)\s*\(', src)))


def test_header_and_exports_agree(lib):
    names = declared_functions()
    assert set(names) == set(phys_optim.EXPORTS)
    for n in names:
        assert hasattr(lib, n), n


def test_version_and_default_config(lib):
    assert lib.chd_phys_version() == phys_optim.ABI_VERSION == int(re.search(r'#define CHD_PHYS_ABI_VERSION (\d+)', open(os.path.join(ROOT, 'include', 'chd_phys.h')).read()).group(1))
    c = phys_capi.ChdConfig()
    lib.chd_config_default(C.byref(c))
    assert list(c.max_iter) == [7000, 7000, 7000, 2500, 2000, 7000] and c.tol == 1e-3      # phys_optim.cpp:571-743, :578
    d = phys_capi.default_config()
    assert (c.w_com_lin, c.w_com_ang, c.w_ee, c.w_smooth, c.w_dur) == (d.w_com_lin, d.w_com_ang, d.w_ee, d.w_smooth, d.w_dur) == (0.4, 1.7, 0.3, 0.1, 0.1)


@pytest.mark.skipif(torch.cuda.is_available(), reason='checks the no-GPU failure mode')
def test_create_fails_loudly_without_gpu(lib):
    h = C.c_void_p()
    rc = lib.chd_phys_create(None, 0, C.byref(h))
    assert rc != 0 and not h.value
    with pytest.raises(phys_optim.PhysError):
        phys_optim.PhysOptim(device=0)


def test_product_never_touches_the_oracle():
    """Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline may use oracle/."""
    pkg = os.path.join(ROOT, 'contact-human-dynamics_amd')
    for dp, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(('.py', '.hpp', '.hip', '.h', '.cpp')):
                txt = open(os.path.join(dp, f), errors='ignore').read()
                assert 'liboracle' not in txt and 'from oracle' not in txt and 'import oracle' not in txt and '#include "../../oracle' not in txt, f


# ---- IK back-projection library (include/chd_ik.h; next row) -----------------------------------------
@pytest.fixture(scope='module')
def ik_lib():
    from chd_amd import ik_backproject
    ik_backproject.build_library()
    return C.CDLL(ik_backproject.LIB_PATH)


def test_ik_header_and_exports_agree(ik_lib):
    from chd_amd import ik_backproject, ik_capi
    src = open(os.path.join(ROOT, 'include', 'chd_ik.h')).read()
    src = re.sub(r'/\*.*?\*/', '', src, flags=re.S)
    names = sorted(set(re.findall(r'\b(chd_ik_[a-z_]+)\s*\(', src)))
    assert set(names) == set(ik_backproject.EXPORTS)
    for n in names:
        assert hasattr(ik_lib, n), n
    cfg = ik_capi.ChdIkConfig()
    ik_lib.chd_ik_config_default(C.byref(cfg))
    assert (cfg.iterations, cfg.translate, cfg.damping, cfg.smoothness, cfg.gamma) == (30, 1, 7.0, 0.001, 1.0)      # towr_utils.py:843


@pytest.mark.skipif(torch.cuda.is_available(), reason='checks the no-GPU failure mode')
def test_ik_fails_loudly_without_gpu(ik_lib):
    from chd_amd.ik_backproject import IkBackProject
    g = np.load(os.path.join(ROOT, 'tests', 'golden', 'ik_golden.npz'))
    case = dict(parents=g['c0_parents'], target_joints=g['c0_target_joints'], targets=g['c0_targets'], rot=g['c0_rot0'], pos=g['c0_pos0'])
    with pytest.raises(RuntimeError):
        IkBackProject(device=0).solve([case])


# ---- kinematic optimisation library (include/chd_kinopt.h; next row) -------------------------------------------
@pytest.fixture(scope='module')
def kin_lib():
    from chd_amd import kinematic_optimizer
    kinematic_optimizer.build_library()
    return C.CDLL(kinematic_optimizer.LIB_PATH)


def test_kinopt_header_exports_and_struct_sizes_agree(kin_lib, tmp_path):
    import subprocess
    from chd_amd import kinematic_optimizer, kinopt_capi
    src = open(os.path.join(ROOT, 'include', 'chd_kinopt.h')).read()
    src = re.sub(r'/\*.*?\*/', '', src, flags=re.S)
    names = sorted(set(re.findall(r'\b(chd_kin_[a-z_]+)\s*\(', src)))
    assert set(names) == set(kinematic_optimizer.EXPORTS)
    for n in names:
        assert hasattr(kin_lib, n), n
    cfg = kinopt_capi.ChdKinConfig()
    kin_lib.chd_kin_config_default(C.byref(cfg))
    assert (cfg.max_nfev, cfg.ftol, cfg.xtol, cfg.gtol, cfg.lsmr_atol, cfg.lsmr_btol, cfg.lsmr_conlim, cfg.lsmr_maxiter) == (50, 1e-8, 1e-8, 1e-12, 1e-6, 1e-6, 1e8, 0)
    assert list(cfg.parents) == [-1, 0, 1, 2, 3, 3, 3, 0, 7, 8, 9, 9, 9, 0, 13, 14, 15, 16, 16, 16, 16, 16, 15, 22, 23, 15, 25, 26]      # combined_body_25.bvh
    csrc = tmp_path / 'k.c'
    csrc.write_text('#include <stdio.h>\n#include "chd_kinopt.h"\nint main(void){printf("%zu %zu\\n", sizeof(chd_kin_config), sizeof(chd_kin_seq)); return 0;}\n')
    exe = tmp_path / 'k'
    subprocess.check_call(['gcc', '-I', os.path.join(ROOT, 'include'), str(csrc), '-o', str(exe)])
    assert [int(v) for v in subprocess.check_output([str(exe)], text=True).split()] == [C.sizeof(kinopt_capi.ChdKinConfig), C.sizeof(kinopt_capi.ChdKinSeq)]


@pytest.mark.skipif(torch.cuda.is_available(), reason='checks the no-GPU failure mode')
def test_kinopt_fails_loudly_without_gpu(kin_lib):
    from chd_amd.kinematic_optimizer import KinSolver, STAGE_WEIGHTS
    g = np.load(os.path.join(ROOT, 'tests', 'golden', 'kinopt_golden.npz'))
    p = dict(offsets=g['c0_fit_offsets'], pose3d=g['c0_poses3D'], root_trans=g['c0_root_pos'], pose2d_n=g['c0_lsq0_pose2d_n'], proj_w=g['c0_lsq0_proj_w'],
             data_w=g['c0_lsq0_data_w'], contact=g['c0_lsq0_vel'], floor_n=np.zeros(3), floor_p=np.zeros(3), weights=STAGE_WEIGHTS[0], x0=g['c0_lsq0_x0'])
    with pytest.raises(RuntimeError):
        KinSolver(device=0).solve([p])


def test_struct_mirrors_match_the_header_sizes(lib, tmp_path):
    """sizeof of every struct of include/chd_phys.h as compiled by the C compiler == the ctypes mirror."""
    import subprocess
    src = tmp_path / 's.c'
    src.write_text('#include <stdio.h>\n#include "chd_phys.h"\nint main(void){printf("%zu %zu %zu %zu %zu %zu\\n", sizeof(chd_config), sizeof(chd_seq_in), '
                   'sizeof(chd_snapshot), sizeof(chd_seq_out), sizeof(chd_batch_stats), sizeof(chd_call_stats)); return 0;}\n')
    exe = tmp_path / 's'
    subprocess.check_call(['gcc', '-I', os.path.join(ROOT, 'include'), str(src), '-o', str(exe)])
    sizes = [int(v) for v in subprocess.check_output([str(exe)], text=True).split()]
    assert sizes == [C.sizeof(phys_capi.ChdConfig), C.sizeof(phys_capi.ChdSeqIn), C.sizeof(phys_capi.ChdSnapshot), C.sizeof(phys_capi.ChdSeqOut),
                     C.sizeof(phys_capi.ChdBatchStats), C.sizeof(phys_capi.ChdCallStats)]
    c = phys_capi.ChdConfig()
    lib.chd_config_default(C.byref(c))
    assert c.stall_window == 0 and c.max_workgroups == 0 and c.threads_per_sequence == 0       # no stall guard unless asked for


def test_bench_hashes_the_sources_the_build_uses():
    """bench.py marks profiles/traffic.json STALE by a hash of the kernel sources: the list must be the one build_library compiles."""
    import re
    from chd_amd import phys_optim
    src = open(os.path.join(ROOT, 'bench.py')).read()
    m = re.search(r"def kernel_sources_sha256\(\):.*?_sources_sha256\(\((.*?)\)\)", src, re.S)
    listed = set(re.findall(r"'([^']+)'", m.group(1)))
    assert listed == set(phys_optim.SOURCES), (listed, phys_optim.SOURCES)
    for f in listed:
        assert os.path.exists(os.path.join(ROOT, 'contact-human-dynamics_amd', 'csrc', f)), f

"""The native readers of the two JSON inputs in front of the kinematic optimisation and the contact network (libchd_prepare.so: csrc/chd_json.hpp, include/chd_prepare.h,
ABI version 2) against the Python mirrors that use the json module (contact_net.load_keypoint_dir = openpose_utils.py:48-76, totalcap_io.load_totalcap_results =
totalcap_utils.py:33-79): value for value, on the synthetic video directories the driver tests use and on hand-made files with the corner cases of the grammar.  Host
code: no GPU needed."""
import json
import os

import numpy as np
import pytest

import chd_amd  # noqa: F401
from chd_amd import contact_net as cn
from chd_amd import prepare_capi as pc
from chd_amd import totalcap_io as tc


@pytest.fixture(scope='module')
def videos(tmp_path_factory):
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from test_kinopt_driver import write_video_dir
    from chd_amd.synth import make_kin_clip
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'kinopt_golden.npz'))
    root = tmp_path_factory.mktemp('videos')
    rng = np.random.default_rng(0)
    dirs = []
    for i, F in enumerate((7, 12, 5)):
        d = str(root / ('video_%03d' % i))
        write_video_dir(d, make_kin_clip(i, F, g['c0_skel_offsets'], g['c0_skel_parents'], upright=True), rng)
        dirs.append(d)
    return dirs


def test_openpose_directories_value_for_value(videos):
    dirs = [os.path.join(v, 'openpose_result') for v in videos]
    for a, d in zip(pc.load_keypoint_dirs(dirs), dirs):
        b = cn.load_keypoint_dir(d)
        assert a.shape == b.shape and np.array_equal(a, b)


def test_total_capture_files_value_for_value(videos):
    paths = [os.path.join(v, 'tracked_results.json') for v in videos]
    for a, p in zip(pc.load_totalcap_batch(paths), paths):
        b = tc.load_totalcap_results(p)
        for f in ('root_trans', 'joint3d', 'smpl_joint3d', 'smpl_joint_angles', 'body_coeffs', 'face_coeffs'):
            x, y = getattr(a, f), getattr(b, f)
            assert x.shape == y.shape and np.array_equal(x, y), f


def test_the_grammars_corner_cases(tmp_path):
    """Numbers in every form the grammar allows (integers, negative zero, exponents with and without sign, 17 significant digits, subnormal and overflowing
    magnitudes), white space everywhere, other members around the ones that are read, strings with escapes, an empty `people` list, files that do not end in
    .json and a file literally called `json` (the reference's test is `name.split('.')[-1] == 'json'`)."""
    d = tmp_path / 'openpose_result'
    d.mkdir()
    nums = ['0', '-0.0', '-0', '1', '-1', '1e0', '1E+2', '2.5e-3', '123456789012345678', '0.1', '1.7976931348623157e308', '5e-324', '1e400', '-1e400', '4.9406564584124654e-324',
            '3.141592653589793', '2.2250738585072014e-308', '0.30000000000000004', '1e-400']
    body = (nums * 5)[:75]
    text = '{ "version" : 1.3 ,\n\t"people":[ {"person_id":[-1], "note":"a \\"quoted\\" \\\\ string \\u00e9 [1,2,{", "pose_keypoints_2d" : [' + ' ,\n '.join(body) + '\n] , "hand_left_keypoints_2d":[]} , {"pose_keypoints_2d":[9,9,9]} ] }\n'
    (d / 'f_000000000001_keypoints.json').write_text(text)
    (d / 'f_000000000000_keypoints.json').write_text('{"people":[],"version":1.3}')
    (d / 'json').write_text('{"people": []}')
    (d / 'notes.txt').write_text('not json')
    (d / 'f_000000000002_keypoints.json.bak').write_text('not json either')
    a = pc.load_keypoint_dirs([str(d)])[0]
    b = cn.load_keypoint_dir(str(d))
    assert a.shape == b.shape == (3, 25, 3)
    assert np.array_equal(a, b) and np.array_equal(np.signbit(a), np.signbit(b))


def test_what_is_not_the_format_fails_its_clip_with_the_files_name(videos, tmp_path):
    good = os.path.join(videos[0], 'openpose_result')
    bad = tmp_path / 'openpose_result'
    bad.mkdir()
    cases = {'trailing comma': '{"people":[{"pose_keypoints_2d":[1,2,3,]}]}', 'bad literal': '{"people":[{"pose_keypoints_2d":[nan]}]}', 'two values': '{"people":[]} {}',
             'unknown escape': '{"note":"a \\q b","people":[]}', 'short unicode escape': '{"note":"\\u12G4","people":[]}',
             'wrong length': '{"people":[{"pose_keypoints_2d":[1,2,3]}]}', 'no people': '{"persons":[]}', 'leading dot': '{"people":[{"pose_keypoints_2d":[.5]}]}',
             'duplicate key': '{"people":[],"people":[]}', 'truncated': '{"people":[{"pose_keypoints_2d":[1,2'}
    for what, text in cases.items():
        for f in bad.iterdir():
            f.unlink()
        (bad / 'frame_0.json').write_text(text)
        with pytest.raises(ValueError, match='frame_0.json'):
            pc.load_keypoint_dirs([good, str(bad)])
        with pytest.raises((ValueError, KeyError, json.JSONDecodeError)) if what not in ('duplicate key', 'wrong length') else _accepts():      # the Python mirror: stricter here only where json is laxer
            cn.load_keypoint_dir(str(bad))
    with pytest.raises(ValueError, match='no .json result files'):
        pc.load_keypoint_dirs([str(tmp_path)])
    p = tmp_path / 'tracked_results.json'
    p.write_text('{"totalcapResults":[{"trans":{"x":1,"y":2},"joints":[],"SMPLJoints":[],"bodyCoeffs":[],"faceCoeffs":[]}]}')
    with pytest.raises(ValueError, match='tracked_results.json: frame 0'):
        pc.load_totalcap_batch([str(p)])
    with pytest.raises(ValueError, match='cannot open'):
        pc.load_totalcap_batch([str(tmp_path / 'absent.json')])


def test_nan_and_infinity_literals_are_read_like_the_json_module_reads_them(tmp_path):
    """Round 6 (advisor): Python's json module accepts NaN / Infinity / -Infinity, and trackers occasionally write them; the native reader used to fail the whole
    run on such a file while the Python mirror went on.  Now both give the same arrays.  Invalid UTF-8 inside a string fails in both."""
    d = tmp_path / 'openpose_result'
    d.mkdir()
    body = ['NaN', 'Infinity', '-Infinity', '1.5', '-2'] * 15
    (d / 'f_0_keypoints.json').write_text('{"people":[{"pose_keypoints_2d":[' + ','.join(body) + ']}]}')
    a = pc.load_keypoint_dirs([str(d)])[0]
    b = cn.load_keypoint_dir(str(d))
    assert a.shape == b.shape == (1, 25, 3)
    assert np.array_equal(a, b, equal_nan=True) and np.isnan(a).sum() == 15 and np.isposinf(a).sum() == 15 and np.isneginf(a).sum() == 15
    (d / 'f_0_keypoints.json').write_bytes(b'{"note":"\xff\xfe","people":[]}')
    with pytest.raises(ValueError, match='UTF-8'):
        pc.load_keypoint_dirs([str(d)])
    with pytest.raises((ValueError, UnicodeDecodeError)):
        cn.load_keypoint_dir(str(d))


class _accepts:
    def __enter__(self):
        return self

    def __exit__(self, *exc):
        return exc[0] is not None and issubclass(exc[0], (ValueError, KeyError))      # (whatever the lax reader does with it is its business)


def test_exports_and_version():
    L = pc.load_library()
    for name in pc.EXPORTS:
        getattr(L, name)
    import re
    hdr = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'include', 'chd_prepare.h')).read()
    assert L.chd_prep_version() == pc.ABI_VERSION == int(re.search(r'#define CHD_PREP_ABI_VERSION (\d+)', hdr).group(1))
    for name in ('chd_openpose_load_dirs', 'chd_openpose_free', 'chd_totalcap_load_batch', 'chd_totalcap_free'):
        assert re.search(r'\b%s\(' % name, hdr)

"""Contact network: this repo's PyTorch implementation against golden vectors produced by the
reference's own classes (tests/golden/make_contact_golden.py, run in the build container)."""
import os

import numpy as np
import pytest
import torch

import chd_amd
from chd_amd import contact_net as cn

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'contact_net_golden.npz')


@pytest.fixture(scope='module')
def gold():
    return np.load(GOLD)


def _model(gold):
    """Seeded weights: identical to the reference OpenPoseModel built under the same seeds (fingerprints in the fixture)."""
    torch.manual_seed(1234)
    return cn.randomize_batchnorm_stats(cn.OpenPoseModel(), seed=7).eval()


def test_parameter_count():
    assert sum(p.numel() for p in cn.OpenPoseModel().parameters()) == 959092


def test_seeded_init_matches_reference(gold):
    m = _model(gold)
    keys = [k for k in gold.files if k.startswith('sd_')]
    assert len(keys) >= 18
    for k in keys:
        v = m.state_dict()[k[3:]].double()
        fp = np.concatenate([[float(v.sum()), float(v.abs().sum())], v.reshape(-1)[:6].numpy()])
        assert np.array_equal(fp, gold[k]), k


def test_windows_bit_exact(gold):
    w = cn.make_windows(gold['raw0'])
    assert w.dtype == np.float32 and np.array_equal(w, gold['windows_vid0'])


def test_logits_and_labels_cpu(gold):
    m = _model(gold)
    vids = [gold['raw0'], gold['raw1'], gold['raw2']]
    fmax = max(v.shape[0] for v in vids)
    pads = [np.concatenate([v, np.repeat(v[-1:], fmax - v.shape[0], axis=0)]) for v in vids]
    x = torch.from_numpy(np.concatenate([cn.make_windows(v) for v in pads]))
    with torch.no_grad():
        lg = m(x).numpy()
    assert np.allclose(lg, gold['logits'], atol=1e-5)
    labels, margin = cn.detect_contacts(vids, m, torch.device('cpu'))
    for k in range(3):
        assert np.array_equal(labels[k], gold['contacts%d' % k])


def test_vote_merge_edges():
    pred = np.zeros((10, 5, 4), dtype=bool)
    pred[0, 0, 0] = True            # first frame: one vote is enough (test.py:103-106)
    pred[3, 2, 1] = True            # interior frame with a single vote: below the threshold of 3
    lab = cn.vote_merge(pred)
    assert lab.shape == (18, 4)
    assert lab[0, 0] == 1 and lab[1, 0] == 1 and lab[2, 0] == 1      # two-frame leading pad repeats frame 0
    assert lab[:, 1].sum() == 0


def test_low_confidence_fill_matches_linear_interpolation():
    op = np.zeros((6, 1, 3)); op[:, 0, 0] = [0, 9, 9, 9, 4, 5]; op[:, 0, 2] = [1, 0, 0, 0, 1, 1]
    out = cn.fill_low_confidence(op)
    assert np.allclose(out[:, 0, 0], [0, 1, 2, 3, 4, 5])


@pytest.mark.gpu
def test_labels_bit_exact_on_gpu(gold):
    m = _model(gold)
    vids = [gold['raw0'], gold['raw1'], gold['raw2']]
    labels, margin = cn.detect_contacts(vids, m, torch.device('cuda:0'))
    for k in range(3):
        assert np.array_equal(labels[k], gold['contacts%d' % k]), 'min |logit| %.3e' % margin
    assert cn.smoke(torch.device('cuda:0')) > 0


def _nasty_videos():
    """Detections with every gap shape: at the start, at the end, one joint never seen, long and adjacent gaps,
    confidences exactly at the threshold."""
    vids = []
    for seed, F in ((0, 40), (1, 90), (2, 33)):
        kp = cn.synthetic_keypoints(seed, F=F)
        rng = np.random.default_rng(100 + seed)
        kp[:5, 3, 2] = 0.05                       # gap at the start
        kp[-7:, 9, 2] = 0.1                       # gap at the end
        kp[:, 20, 2] = 0.0                        # never detected
        kp[10:31, 11, 2] = 0.19                   # long gap
        kp[12:14, 12, 2] = 0.0; kp[15:17, 12, 2] = 0.0     # two gaps separated by one good frame
        kp[8, 13, 2] = 0.2                        # exactly the threshold: confident
        drop = rng.uniform(size=kp.shape[:2]) < 0.15
        kp[:, :, 2][drop] = rng.uniform(0.0, 0.19, drop.sum())
        vids.append(kp)
    return vids


def test_device_preprocessing_is_bit_identical_to_numpy():
    """make_windows_device / vote_merge_device / detect_contacts_device (tensor ops, here on the CPU device) against the
    NumPy path that is pinned to the reference's classes: windows bit for bit, labels equal, videos of different length."""
    vids = _nasty_videos()
    for v in vids:
        got = cn.make_windows_device(torch.from_numpy(v[None]))[0].numpy()
        want = cn.make_windows(v)
        assert got.dtype == np.float32 and got.shape == want.shape and np.array_equal(got, want)
    same = [vids[1], vids[1][::-1].copy(), cn.synthetic_keypoints(5, F=90)]
    batch = cn.make_windows_device(torch.from_numpy(np.stack(same)))
    for k, v in enumerate(same):
        assert np.array_equal(batch[k].numpy(), cn.make_windows(v))
    rng = np.random.default_rng(0)
    pred = rng.uniform(size=(3, 30, 5, 4)) < 0.5
    lab = cn.vote_merge_device(torch.from_numpy(pred)).numpy()
    for k in range(3):
        assert np.array_equal(lab[k], cn.vote_merge(pred[k]))
    torch.manual_seed(0)
    model = cn.randomize_batchnorm_stats(cn.OpenPoseModel(), seed=0)
    a, _ = cn.detect_contacts(vids, model, torch.device('cpu'))
    b, _ = cn.detect_contacts_device(vids, model, torch.device('cpu'))
    assert all(np.array_equal(x, y) and x.shape == (v.shape[0], 4) for x, y, v in zip(a, b, vids))


@pytest.mark.gpu
def test_device_preprocessing_on_gpu():
    """The tensor-op pre- / post-processing on the HIP device: same windows and labels as the NumPy path (first run on an
    MI355X at the start of round 2: profiles/r02a_round_start)."""
    if not torch.cuda.is_available():
        pytest.skip('needs an MI355X')
    dev = torch.device('cuda:0')
    vids = _nasty_videos()
    for v in vids:
        assert np.array_equal(cn.make_windows_device(torch.from_numpy(v[None]).to(dev))[0].cpu().numpy(), cn.make_windows(v))
    torch.manual_seed(0)
    model = cn.randomize_batchnorm_stats(cn.OpenPoseModel(), seed=0)
    a, _ = cn.detect_contacts(vids, model, dev)
    b, _ = cn.detect_contacts_device(vids, model, dev)
    assert all(np.array_equal(x, y) for x, y in zip(a, b))


def test_directory_interface_json_in_npy_out(tmp_path):
    """run_detect_contacts (scripts/run_detect_contacts.py): OpenPose JSON directories in, foot_contacts.npy (F x 4 ints,
    [l_heel, l_toe, r_heel, r_toe]) out; frames without a detected person are all-zero detections
    (openpose_utils.py:60-63); host and device-op pre-processing write the same labels."""
    import json
    from chd_amd import run_detect_contacts as cli
    vids = {'clipA': cn.synthetic_keypoints(3, F=30), 'clipB': cn.synthetic_keypoints(4, F=45)}
    for name, kp in vids.items():
        d = tmp_path / 'data' / name / 'openpose_result'
        os.makedirs(d)
        for f in range(kp.shape[0]):
            people = [] if (name == 'clipA' and f in (4, 5)) else [{'pose_keypoints_2d': [float(x) for x in kp[f].reshape(-1)]}]
            json.dump({'version': 1.3, 'people': people}, open(d / ('%s_%012d_keypoints.json' % (name, f)), 'w'))
        open(tmp_path / 'data' / name / 'not_a_frame.txt', 'w').write('x')
    os.makedirs(tmp_path / 'data' / 'no_openpose_here')
    torch.manual_seed(0)
    model = cn.randomize_batchnorm_stats(cn.OpenPoseModel(), seed=0)
    w = str(tmp_path / 'op_only_weights.pth')
    torch.save(model.state_dict(), w)
    assert cli.main(['--data', str(tmp_path / 'data'), '--weights', w]) == 0
    first = {n: np.load(tmp_path / 'data' / n / 'foot_contacts.npy') for n in vids}
    assert cli.main(['--data', str(tmp_path / 'data'), '--weights', w, '--device-ops']) == 0
    kpA = vids['clipA'].copy(); kpA[4:6] = 0.0
    want, _ = cn.detect_contacts([kpA, vids['clipB']], model, torch.device('cpu'))
    for (n, kp), ref in zip(vids.items(), want):
        got = np.load(tmp_path / 'data' / n / 'foot_contacts.npy')
        assert got.shape == (kp.shape[0], 4) and got.dtype == np.int64 and set(np.unique(got)) <= {0, 1}
        assert np.array_equal(got, first[n])
        if cn.select_device().type == 'cpu':
            assert np.array_equal(got, ref)
    assert not os.path.exists(tmp_path / 'data' / 'no_openpose_here' / 'foot_contacts.npy')

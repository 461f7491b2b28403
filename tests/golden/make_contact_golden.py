"""Generates tests/golden/contact_net_golden.npz by running the REFERENCE's own code
(/root/reference/src/contact_learning: OpenPoseModel, RealVideoDataset pre-processing and
test.py::val_full_video vote merge) on seeded synthetic OpenPose detections.

Runs only in the build container (needs /root/reference); the .npz travels with the repo.
Modules the reference imports but never uses on this path (skimage, torchvision, cv2,
OneEuroFilter) are stubbed; ``np.int`` (removed in NumPy >= 1.24, used at test.py:107,151) is aliased.
"""
import json
import os
import re
import sys
import tempfile
import types

import numpy as np
import torch

REF = '/root/reference/src'
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)


def stub(name, **attrs):
    m = types.ModuleType(name)
    for k, v in attrs.items():
        setattr(m, k, v)
    sys.modules[name] = m
    return m


def main():
    np.int = int
    stub('skimage', io=types.SimpleNamespace(), transform=types.SimpleNamespace())
    stub('torchvision', transforms=types.SimpleNamespace(), utils=types.SimpleNamespace())
    stub('cv2')
    stub('OneEuroFilter', OneEuroFilter=object)
    for p in (REF, os.path.join(REF, 'contact_learning'), os.path.join(REF, 'utils')):
        sys.path.insert(0, p)
    cwd = os.getcwd()
    os.chdir(REF)
    from data.real_video_dataset import RealVideoDataset          # reference pre-processing
    from models.openpose_only import OpenPoseModel                # reference network
    src = open(os.path.join(REF, 'contact_learning', 'test.py')).read()
    fn = re.search(r'\ndef val_full_video\(.*?(?=\ndef )', src, re.S).group(0)
    ns = {'np': np, 'torch': torch, 'os': os}
    exec(fn, ns)                                                  # reference vote merge + save
    val_full_video = ns['val_full_video']
    os.chdir(cwd)

    import chd_amd  # noqa: F401
    from chd_amd.contact_net import randomize_batchnorm_stats, synthetic_keypoints

    Fs = [60, 47, 33]
    with tempfile.TemporaryDirectory() as tmp:
        raw = []
        for v, F in enumerate(Fs):
            kp = synthetic_keypoints(100 + v, F=F)
            raw.append(kp)
            d = os.path.join(tmp, 'data', 'vid%d' % v, 'openpose_result')
            os.makedirs(d)
            for i in range(F):
                with open(os.path.join(d, 'frame_%06d_keypoints.json' % i), 'w') as fh:
                    json.dump({'people': [{'pose_keypoints_2d': kp[i].reshape(-1).tolist()}]}, fh)
        ds = RealVideoDataset(os.path.join(tmp, 'data'), split='test', window_size=9, contact_size=5, dimensions=(1920, 1080),
                              load_img=False, use_confidence=True, joint_set='lower')
        torch.manual_seed(1234)
        model = OpenPoseModel(9, 13, 5, 3)
        randomize_batchnorm_stats(model, seed=7)
        model.eval()
        nwin = ds.get_num_test_windows_per_seq()
        loader = torch.utils.data.DataLoader(ds, batch_size=nwin, shuffle=False)
        out = os.path.join(tmp, 'out')
        with torch.no_grad():
            val_full_video(loader, ds, model, torch.device('cpu'), 0.5, 5, contacts_out_path=out)
            windows = torch.stack([ds[i]['joint2d'] for i in range(len(ds))]).numpy()
            logits = model(torch.from_numpy(windows)).numpy()
        contacts = [np.load(os.path.join(out, 'vid%d' % v, 'foot_contacts.npy')) for v in range(len(Fs))]
    # weights are reproducible from the seeds (torch.manual_seed(1234) + randomize_batchnorm_stats(seed=7)); keep only a fingerprint
    sd = {k: np.concatenate([[float(v.double().sum()), float(v.double().abs().sum())], v.double().reshape(-1)[:6].numpy()])
          for k, v in model.state_dict().items() if v.dtype.is_floating_point}
    np.savez_compressed(os.path.join(HERE, 'contact_net_golden.npz'),
                        raw0=raw[0], raw1=raw[1], raw2=raw[2], windows_vid0=windows[:nwin], logits=logits.astype(np.float32),
                        contacts0=contacts[0], contacts1=contacts[1], contacts2=contacts[2], nwin=nwin,
                        **{'sd_' + k: v for k, v in sd.items()})
    print('wrote contact_net_golden.npz: windows', windows.shape, 'logits', logits.shape, 'contacts', [c.shape for c in contacts],
          'min |logit|', np.abs(logits).min())


if __name__ == '__main__':
    main()

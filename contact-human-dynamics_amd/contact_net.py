"""Foot-contact network on PyTorch-ROCm (the stage that feeds the physics optimisation).

Mirrors, for inference only:
  * the model          ``src/contact_learning/models/openpose_only.py:14-78``  (same module tree, so a
                       reference ``op_only_weights.pth`` state_dict loads unchanged)
  * the pre-processing ``src/contact_learning/data/real_video_dataset.py:140-161, 206-276`` and
                       ``openpose_dataset.py:49-121`` (low-confidence gap interpolation, 1280x720 scaling,
                       pixel normalisation, root-relative 9-frame windows over the 13 lower-body joints)
  * the vote merge     ``src/contact_learning/test.py:87-122``
  * the file output    ``foot_contacts.npy`` int F x 4 = [l_heel, l_toe, r_heel, r_toe] (README.md:87)

``torch.cuda`` on a ROCm build *is* the HIP device, so the reference's device selection
(``utils.py:48-58``) carries over.  All windows of all videos go through one batched forward.
"""
import json
import os

import numpy as np
import torch
import torch.nn as nn

WINDOW_SIZE = 9          # test.py --window-size default used by run_detect_contacts.py
PRED_SIZE = 5            # --contact-size
OP_ROOT_JOINT = 8        # MidHip (openpose_dataset.py:19)
OP_LOWER_JOINTS = [8, 9, 10, 11, 12, 13, 14, 19, 20, 21, 22, 23, 24]   # OP_JOINT_SUBSETS['lower'] (openpose_dataset.py:38)
TRAIN_DIM = (1280, 720)                      # real_video_dataset.py:17
TRAIN_NORMALIZATION = 200.4160302695367      # real_video_dataset.py:18
CONF_THRESH = 0.2                            # real_video_dataset.py:158


class OpenPoseModel(nn.Module):
    """351 -> 1024 -> 512 -> 128 -> 32 -> 20 MLP with BatchNorm + ReLU (+ Dropout 0.3 before the 4th Linear)."""

    def __init__(self, window_size=WINDOW_SIZE, joints=len(OP_LOWER_JOINTS), pred_size=PRED_SIZE, feat_size=3):
        super().__init__()
        self.window_size, self.contact_size, self.feat_size = window_size, pred_size, feat_size
        self.model = nn.Sequential(
            nn.Linear(window_size * joints * feat_size, 1024), nn.BatchNorm1d(1024), nn.ReLU(),
            nn.Linear(1024, 512), nn.BatchNorm1d(512), nn.ReLU(),
            nn.Linear(512, 128), nn.BatchNorm1d(128), nn.ReLU(),
            nn.Dropout(p=0.3),
            nn.Linear(128, 32), nn.BatchNorm1d(32), nn.ReLU(),
            nn.Linear(32, 4 * pred_size))
        for m in self.model:
            if isinstance(m, nn.Linear):
                nn.init.xavier_uniform_(m.weight)
                m.bias.data.fill_(0.01)

    def forward(self, x):                      # B x N x J x F  ->  B x pred_size x 4 logits
        b = x.shape[0]
        return self.model(x.reshape(b, -1)).view(b, self.contact_size, 4)

    @staticmethod
    def prediction(logits):
        """sigmoid(logit) > 0.5  <=>  logit > 0 (openpose_only.py:75-78)."""
        return logits > 0


def randomize_batchnorm_stats(model, seed=0):
    """No pretrained weights are reachable offline (pretrained_weights/download.sh); tests use seeded
    weights and give BatchNorm non-trivial running statistics so the check is not vacuous."""
    g = torch.Generator().manual_seed(seed)
    for m in model.modules():
        if isinstance(m, nn.BatchNorm1d):
            m.running_mean.copy_(0.3 * torch.randn(m.num_features, generator=g))
            m.running_var.copy_(0.5 + torch.rand(m.num_features, generator=g))
            m.weight.data.copy_(0.8 + 0.4 * torch.rand(m.num_features, generator=g))
            m.bias.data.copy_(0.1 * torch.randn(m.num_features, generator=g))
    return model


# ---------------------------------------------------------------------------------------------
# pre-processing (float64 NumPy, cast to fp32 at the end — real_video_dataset.py:267)
# ---------------------------------------------------------------------------------------------
def load_keypoint_dir(dir_path, num_joints=25):
    """OpenPose JSON directory -> F x 25 x 3 (first person; all zeros when nobody is detected).
    openpose_utils.py:48-76."""
    files = sorted(f for f in os.listdir(dir_path) if f.split('.')[-1] == 'json')
    out = []
    for f in files:
        with open(os.path.join(dir_path, f)) as fh:
            d = json.load(fh)
        if len(d['people']) == 0:
            out.append(np.zeros((num_joints, 3)))
        else:
            out.append(np.array(d['people'][0]['pose_keypoints_2d'], dtype=np.float64).reshape(-1, 3))
    return np.stack(out, axis=0)


def fill_low_confidence(op, thresh=CONF_THRESH):
    """Replace (x, y) of detections with confidence < thresh by linear interpolation between the
    neighbouring confident frames (edges: copy the nearest confident frame).  Same arithmetic as
    ``process_openpose_data`` (openpose_dataset.py:49-103), including its accumulated step.
    One quirk is reproduced on purpose: a bad run reaching the end of the clip also overwrites the
    last *good* frame with itself (slice starts at init_valid_frame), which is a no-op."""
    op = op.copy()
    F, J = op.shape[0], op.shape[1]
    xy, conf = op[:, :, :2], op[:, :, 2]
    for j in range(J):
        bad = conf[:, j] < thresh
        if not bad.any():
            continue
        t = 0
        while t < F:
            if not bad[t]:
                t += 1
                continue
            nxt = t + 1
            while nxt < F and bad[nxt]:
                nxt += 1
            prev = t - 1
            if t == 0 and nxt == F:
                pass
            elif t == 0:
                xy[:nxt, j, :] = xy[nxt, j, :]
            elif nxt == F:
                xy[prev:, j, :] = xy[prev, j, :]
            else:
                step = 1.0 / (nxt - prev)
                cur = step
                for k in range(t, nxt):
                    xy[k, j, :] = (1.0 - cur) * xy[prev, j, :] + cur * xy[nxt, j, :]
                    cur += step
            t = nxt
    return op


def make_windows(op, dimensions=(1920, 1080), window_size=WINDOW_SIZE):
    """F x 25 x 3 raw OpenPose detections -> (F - window_size + 1) x window_size x 13 x 3 float32 network input."""
    op = np.array(op, dtype=np.float64, copy=True)
    scale_w = float(TRAIN_DIM[0]) / dimensions[0]
    op[:, :, :2] *= scale_w                                   # real_video_dataset.py:147-153
    op = fill_low_confidence(op)
    op[:, :, :2] /= TRAIN_NORMALIZATION                       # :161
    F = op.shape[0]
    nwin = F - 2 * (window_size // 2)
    if nwin <= 0:
        raise ValueError('video shorter than one window')
    idx = np.arange(nwin)[:, None] + np.arange(window_size)[None, :]
    win = op[idx]                                             # nwin x W x 25 x 3 (copy)
    mid = window_size // 2
    root = win[:, mid, OP_ROOT_JOINT, :2].copy()              # __getitem__ :243-252
    win[:, :, :, :2] -= root[:, None, None, :]
    win[:, mid, OP_ROOT_JOINT, :2] = root
    win = win[:, :, OP_LOWER_JOINTS, :]
    return win.astype(np.float32)


def vote_merge(pred, window_size=WINDOW_SIZE, pred_size=PRED_SIZE):
    """B x pred_size x 4 boolean window predictions -> F x 4 int labels (test.py:91-122)."""
    pred = np.asarray(pred)
    B = pred.shape[0]
    votes = np.zeros((B + 2 * (pred_size // 2), 4))
    for k in range(pred_size):
        votes[k:k + B] += pred[:, k, :]
    thresh = np.ones(votes.shape[0]) * ((pred_size + 1) / 2)
    for off in range(pred_size - 1):
        thresh[off] = (off // 2) + 1
        thresh[-1 - off] = (off // 2) + 1
    labels = (votes >= thresh.reshape(-1, 1)).astype(np.int64)
    pad = (window_size - pred_size) // 2
    return np.concatenate([np.repeat(labels[:1], pad, axis=0), labels, np.repeat(labels[-1:], pad, axis=0)], axis=0)


# ---------------------------------------------------------------------------------------------
# the same pre- / post-processing as tensor ops on the device: V videos at once, float64 like the
# reference's NumPy code and operation by operation in its order, so the windows are bit-identical to `make_windows`
# ---------------------------------------------------------------------------------------------
def fill_low_confidence_device(op, thresh=CONF_THRESH):
    """`fill_low_confidence` for a V x F x J x 3 float64 tensor.  Neighbouring confident frames come from index scans
    (cummax / cummin over frame numbers: integers, exact); the reference's accumulated interpolation weight
    (`cur += step` once per frame of a gap) is reproduced by a loop over the position inside the gap -- as long as the
    longest gap, elementwise over all videos and joints -- because a parallel prefix sum would round differently."""
    V, F, J = op.shape[0], op.shape[1], op.shape[2]
    xy, conf = op[..., :2], op[..., 2]
    good = ~(conf < thresh)
    t = torch.arange(F, device=op.device).view(1, F, 1).expand(V, F, J)
    prev = torch.cummax(torch.where(good, t, torch.full_like(t, -1)), dim=1).values
    nxt = torch.flip(torch.cummin(torch.flip(torch.where(good, t, torch.full_like(t, F)), dims=[1]), dim=1).values, dims=[1])
    bad = ~good
    has_prev, has_next = prev >= 0, nxt < F
    a = torch.gather(xy, 1, prev.clamp(min=0).unsqueeze(-1).expand(V, F, J, 2))          # value at the previous confident frame
    b = torch.gather(xy, 1, nxt.clamp(max=F - 1).unsqueeze(-1).expand(V, F, J, 2))       # ... at the next one
    inside = bad & has_prev & has_next
    r = torch.where(inside, t - prev, torch.zeros_like(t))                                 # 1, 2, ... inside a gap
    step = 1.0 / torch.where(inside, nxt - prev, torch.ones_like(t)).to(torch.float64)
    cur = step.clone()
    w = torch.where(r == 1, cur, torch.zeros_like(cur))
    for q in range(2, int(r.max().item()) + 1):
        cur = cur + step
        w = torch.where(r == q, cur, w)
    w = w.unsqueeze(-1)
    interp = (1.0 - w) * a + w * b
    new_xy = torch.where(inside.unsqueeze(-1), interp, xy)
    new_xy = torch.where((bad & ~has_prev & has_next).unsqueeze(-1), b, new_xy)           # gap at the start: copy the first confident frame
    new_xy = torch.where((bad & has_prev & ~has_next).unsqueeze(-1), a, new_xy)           # gap at the end: copy the last one
    return torch.cat([new_xy, conf.unsqueeze(-1)], dim=-1)


def make_windows_device(op, dimensions=(1920, 1080), window_size=WINDOW_SIZE):
    """V x F x 25 x 3 float64 tensor of raw detections -> V x (F - window_size + 1) x window_size x 13 x 3 float32, the
    values `make_windows` produces for every video."""
    op = op.to(torch.float64).clone()
    op[..., :2] *= float(TRAIN_DIM[0]) / dimensions[0]
    op = fill_low_confidence_device(op)
    op[..., :2] /= TRAIN_NORMALIZATION
    if op.shape[1] - 2 * (window_size // 2) <= 0:
        raise ValueError('video shorter than one window')
    win = op.unfold(1, window_size, 1).permute(0, 1, 4, 2, 3).contiguous()               # V x nwin x W x 25 x 3
    mid = window_size // 2
    root = win[:, :, mid, OP_ROOT_JOINT, :2].clone()
    win[..., :2] -= root[:, :, None, None, :]
    win[:, :, mid, OP_ROOT_JOINT, :2] = root
    return win[:, :, :, OP_LOWER_JOINTS, :].to(torch.float32)


def vote_merge_device(pred, window_size=WINDOW_SIZE, pred_size=PRED_SIZE):
    """V x B x pred_size x 4 boolean window predictions -> V x F x 4 int64 labels (`vote_merge` per video; integer sums)."""
    V, B = pred.shape[0], pred.shape[1]
    n = B + 2 * (pred_size // 2)
    votes = torch.zeros((V, n, 4), dtype=torch.int64, device=pred.device)
    for k in range(pred_size):
        votes[:, k:k + B] += pred[:, :, k, :].to(torch.int64)
    thresh = torch.full((n,), -(-(pred_size + 1) // 2), dtype=torch.int64)            # votes >= (pred_size + 1) / 2 for integer votes
    for off in range(pred_size - 1):
        thresh[off] = (off // 2) + 1
        thresh[-1 - off] = (off // 2) + 1
    labels = (votes >= thresh.to(pred.device).view(1, n, 1)).to(torch.int64)
    pad = (window_size - pred_size) // 2
    return torch.cat([labels[:, :1].expand(V, pad, 4), labels, labels[:, -1:].expand(V, pad, 4)], dim=1)


@torch.no_grad()
def detect_contacts_device(videos, model, device, dimensions=(1920, 1080)):
    """`detect_contacts` with the pre- and post-processing on the device: the raw detections are uploaded once (padded
    to the longest video as there), labels come back.  Same labels, bit for bit."""
    model = model.to(device).eval()
    fmax = max(v.shape[0] for v in videos)
    raw = np.stack([np.concatenate([np.asarray(v, dtype=np.float64), np.repeat(np.asarray(v, dtype=np.float64)[-1:], fmax - v.shape[0], axis=0)], axis=0)
                    for v in videos], axis=0)
    x = make_windows_device(torch.from_numpy(raw).to(device), dimensions)
    V, B = x.shape[0], x.shape[1]
    logits = model(x.view(V * B, *x.shape[2:]))
    margin = float(logits.abs().min().item())
    labels = vote_merge_device(OpenPoseModel.prediction(logits).view(V, B, *logits.shape[1:])).cpu().numpy()
    return [labels[k, :v.shape[0]] for k, v in enumerate(videos)], margin


# ---------------------------------------------------------------------------------------------
# inference
# ---------------------------------------------------------------------------------------------
def select_device(prefer_gpu=True):
    return torch.device('cuda:0') if (prefer_gpu and torch.cuda.is_available()) else torch.device('cpu')


@torch.no_grad()
def detect_contacts(videos, model, device, dimensions=(1920, 1080)):
    """videos: list of F_i x 25 x 3 arrays.  One batched forward over the windows of all videos.
    As in the reference (real_video_dataset.py:133-145, fix_data_len :165-190) shorter videos are padded to the
    longest one by repeating their last frame *before* the pre-processing, and the labels are trimmed back to
    the true length afterwards (test.py:147-151).
    Returns (list of F_i x 4 int label arrays, min |logit| over all windows)."""
    model = model.to(device).eval()
    fmax = max(v.shape[0] for v in videos)
    wins = []
    for v in videos:
        v = np.asarray(v, dtype=np.float64)
        if v.shape[0] < fmax:
            v = np.concatenate([v, np.repeat(v[-1:], fmax - v.shape[0], axis=0)], axis=0)
        wins.append(make_windows(v, dimensions))
    x = torch.from_numpy(np.concatenate(wins, axis=0)).to(device)
    logits = model(x)
    margin = float(logits.abs().min().item())
    pred = OpenPoseModel.prediction(logits).cpu().numpy()
    out, o = [], 0
    for w, v in zip(wins, videos):
        out.append(vote_merge(pred[o:o + w.shape[0]])[:v.shape[0]])
        o += w.shape[0]
    return out, margin


def run_on_directory(data_root, weights_path, out_root=None, dimensions=(1920, 1080), device=None, device_ops=False):
    """Host interface of ``scripts/run_detect_contacts.py``: every sub-directory of ``data_root`` with an
    ``openpose_result`` folder gets a ``foot_contacts.npy``.  ``device_ops``: gap interpolation, windowing and vote merge as
    tensor ops on the device (`detect_contacts_device`; same labels) instead of NumPy on the host."""
    device = device or select_device()
    model = OpenPoseModel()
    model.load_state_dict(torch.load(weights_path, map_location='cpu'))
    names = sorted(d for d in os.listdir(data_root) if os.path.isdir(os.path.join(data_root, d, 'openpose_result')))
    # (prepare_capi.load_keypoint_dirs reads these small files in 8 ms instead of 37 for 32 videos; the kinematic driver, whose tracked_results.json are 200 ms of json-module
    #  parsing, uses the native readers -- here the json module stays: nothing to gain next to the model's first use of the GEMM libraries in a fresh process, 0.2-0.9 s)
    videos = [load_keypoint_dir(os.path.join(data_root, n, 'openpose_result')) for n in names]
    labels, _ = (detect_contacts_device if device_ops else detect_contacts)(videos, model, device, dimensions)
    for n, lab in zip(names, labels):
        dst = os.path.join(out_root or data_root, n)
        os.makedirs(dst, exist_ok=True)
        np.save(os.path.join(dst, 'foot_contacts'), lab.astype(np.int64))
    return dict(zip(names, labels))


def synthetic_keypoints(seed, F=90, dimensions=(1920, 1080)):
    """OpenPose-like detections for tests / benchmarks: a walking stick figure with
    confidences in U[0,1] and 5 % dropouts below the 0.2 threshold."""
    rng = np.random.default_rng(seed)
    t = np.arange(F) / 30.0
    kp = np.zeros((F, 25, 3))
    base = np.stack([400 + 300 * t, 500 + 10 * np.sin(2 * np.pi * t)], axis=1)
    for j in range(25):
        off = np.array([40.0 * np.cos(j * 1.3), 25.0 * j - 200.0])
        sw = 60.0 * np.sin(2 * np.pi * (t + 0.07 * j))[:, None] * np.array([1.0, 0.15])
        kp[:, j, :2] = base + off + sw + rng.normal(0, 2.0, (F, 2))
    kp[:, :, 2] = rng.uniform(0.25, 1.0, (F, 25))
    drop = rng.uniform(size=(F, 25)) < 0.05
    kp[:, :, 2][drop] = rng.uniform(0.0, 0.19, drop.sum())
    kp[:, :, 0] *= dimensions[0] / 1920.0
    kp[:, :, 1] *= dimensions[1] / 1080.0
    return kp


def smoke(device):
    """Same seeded model on ``device`` and on the CPU: labels must agree bit-exactly."""
    torch.manual_seed(0)
    model = randomize_batchnorm_stats(OpenPoseModel(), seed=0)
    vids = [synthetic_keypoints(s, F=60) for s in range(2)]
    lab_d, margin = detect_contacts(vids, model, device)
    lab_c, _ = detect_contacts(vids, model, torch.device('cpu'))
    for a, b in zip(lab_d, lab_c):
        assert np.array_equal(a, b), 'contact labels differ between %s and cpu (min |logit| %.3e)' % (device, margin)
    return margin

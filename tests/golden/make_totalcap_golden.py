"""Generates tests/golden/totalcap_golden.npz: a small synthetic `tracked_results.json` (the monocular-total-capture output the
kinematic optimisation starts from) and what the REFERENCE's own ingest functions make of it -- `load_totalcap_results`,
`normalize_root_pos`, `create_combined_model`, `combined_angles_from_smpl` (src/utils/totalcap_utils.py:33-79, :134-186), in the
order `optimize_2d_3d` calls them (src/optimize/kinematic_optimizer.py:64-74, :153).  Run in the build container only (it reads
/root/reference); tests use the committed fixture.

    python tests/golden/make_totalcap_golden.py
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF = '/root/reference/src'

if __name__ == '__main__':
    import matplotlib
    matplotlib.use('Agg')
    sys.path.insert(0, os.path.join(REF, 'utils'))
    import totalcap_utils as tu
    rng = np.random.default_rng(11)
    F = 6
    frames = []
    for f in range(F):
        frames.append({'trans': dict(zip('xyz', (rng.normal(size=3) * 30 + [0, 40, 300]).tolist())),
                       'joints': [{'pos': dict(zip('xyz', (rng.normal(size=3) * 40).tolist()))} for _ in range(25)],
                       'SMPLJoints': [{'pos': dict(zip('xyz', (rng.normal(size=3) * 40).tolist())), 'rot': dict(zip('xyz', (rng.normal(size=3) * 0.5).tolist()))} for _ in range(22)],
                       'bodyCoeffs': rng.normal(size=30).tolist(), 'faceCoeffs': rng.normal(size=200).tolist()})
    text = json.dumps({'totalcapResults': frames})
    path = '/tmp/tracked_results.json'
    open(path, 'w').write(text)
    res = tu.load_totalcap_results(path)
    b_root, b3d = tu.normalize_root_pos(res.root_trans, res.joint3d)
    s_root, s3d = tu.normalize_root_pos(res.root_trans, res.smpl_joint3d, root_idx=tu.SMPL_ROOT_IDX)
    out = dict(json_text=np.frombuffer(text.encode(), dtype=np.uint8), root_trans=res.root_trans, joint3d=res.joint3d, smpl_joint3d=res.smpl_joint3d,
               smpl_joint_angles=res.smpl_joint_angles, body25_root_pos=b_root, body25_3d=b3d, smpl_3d=s3d,
               poses3D=tu.create_combined_model(b3d, s3d), init_combined_joint_rot=tu.combined_angles_from_smpl(res.smpl_joint_angles))
    np.savez_compressed(os.path.join(HERE, 'totalcap_golden.npz'), **out)
    print('wrote totalcap_golden.npz', {k: v.shape for k, v in out.items()})

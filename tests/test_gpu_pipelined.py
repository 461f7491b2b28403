"""GPU: the whole call (chd_phys_solve_batch: set-up, upload, persistent launches, fallbacks and fetch pipelined over chunks, workspaces claimed by the
workgroups) against the split interface (upload, one launch, fetch) on the same sequences: bit-identical results, whatever the chunk plan."""
import numpy as np
import pytest

import chd_amd  # noqa: F401
from chd_amd.synth import make_walk

pytestmark = pytest.mark.gpu
CAP = [300] * 6


def _same(a, b):
    if a.stage_status != b.stage_status or a.stage_iters != b.stage_iters:
        return False
    for sa, sb in zip(a.snapshots, b.snapshots):
        for k in ('base_lin', 'base_ang_deg', 'ee_pos', 'ee_force', 'contact'):
            if not np.array_equal(getattr(sa, k), getattr(sb, k)):
                return False
    return True


@pytest.mark.parametrize('chunk', [0, 200, -1])
def test_pipelined_call_equals_the_split_interface(chunk):
    """600 sequences of 30 .. 45 frames with ONE 90-frame sequence near the end: with the automatic plan (chunk 0) and with chunks of 200 the call runs several
    launches at once, which share one set of workspaces (claimed and released by the resident workgroups), and the long sequence arrives when launches are
    already in flight -- the workspaces have to grow mid-call.  -1 = one chunk (set-up, solve, fetch in turn)."""
    from chd_amd.phys_optim import PhysOptim, default_config
    seqs = [make_walk(seed=5000 + i, F=30 + (i % 4) * 5, randomize=True) for i in range(600)]
    seqs[570] = make_walk(seed=7, F=90, randomize=True)
    ref_solver = PhysOptim(device=0, config=default_config(max_iter=CAP))
    ref, _ = ref_solver.solve(seqs)
    ref_solver.close()
    s = PhysOptim(device=0, config=default_config(max_iter=CAP, pipeline_chunk=chunk))
    got, cs = s.solve_batch(seqs)
    assert cs['n_sequences'] == 600 and cs['n_rejected'] == 0
    assert cs['n_chunks'] == (1 if chunk < 0 else 3)
    bad = [i for i in range(600) if not _same(got[i], ref[i])]
    assert not bad, bad[:10]
    assert sum(r.total_iters for r in got) == cs['total_iters']
    # a second call on the warm handle (workspaces and lane buffers reused) gives the same again
    got2, _ = s.solve_batch(seqs[:300])
    assert all(_same(got2[i], ref[i]) for i in range(300))
    s.close()

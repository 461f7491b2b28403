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


def test_lanes_are_reused_by_calls_of_growing_sequence_length():
    """ADVICE r04 (use-after-free of the page-locked staging): three calls on ONE handle, each with longer sequences than the one before, chunks of 100 so that
    every lane is used by every call -- result, statistics and scratch staging of a lane all have to grow between and within the calls (a later chunk with a
    larger result stride; a stage-4 fallback in a small chunk, whose table regions are larger than the chunk's statistics) -- and a repeat of the first call
    at the end, on the lanes the long call left behind.  Every call must equal the split interface on a fresh handle."""
    from chd_amd.phys_optim import PhysOptim, default_config
    cap = [300, 300, 300, 300, 6, 300]          # stage 3 capped at 6 iterations: most sequences take the stage-4 fallback launch
    plans = [[30 + (i % 3) * 5 for i in range(250)], [40 + (i % 5) * 10 for i in range(230)], [60 if i % 7 else 100 for i in range(220)]]
    s = PhysOptim(device=0, config=default_config(max_iter=cap, pipeline_chunk=100))
    calls = []
    for k, frames in enumerate(plans + [plans[0]]):
        seqs = [make_walk(seed=9000 + 1000 * (k % 3) + i, F=f, randomize=True) for i, f in enumerate(frames)]
        got, cs = s.solve_batch(seqs)
        assert cs['n_chunks'] >= 2 and cs['n_fallback'] > 0
        calls.append((seqs, got))
    s.close()
    for seqs, got in calls:
        r = PhysOptim(device=0, config=default_config(max_iter=cap))
        ref, _ = r.solve(seqs)
        r.close()
        bad = [i for i in range(len(seqs)) if not _same(got[i], ref[i])]
        assert not bad, bad[:10]


def test_more_resident_workgroups_than_workspace_slots():
    """with max_workgroups below the compute-unit count the launches of a pipelined call bring more resident workgroups than the handle has
    workspace slots (3 launches x 24 workgroups against 24 slots).  The late ones must WAIT for a slot and then drain their queue -- until round 5 they gave up
    after a bounded spin and their sequences came back unsolved with status 0.  (The host now also checks every launch's queue counter.)"""
    from chd_amd.phys_optim import PhysOptim, default_config
    seqs = [make_walk(seed=12000 + i, F=30 + (i % 3) * 5, randomize=True) for i in range(300)]
    r = PhysOptim(device=0, config=default_config(max_iter=CAP))
    ref, _ = r.solve(seqs)
    r.close()
    s = PhysOptim(device=0, config=default_config(max_iter=CAP, pipeline_chunk=100, max_workgroups=24))
    got, cs = s.solve_batch(seqs)
    s.close()
    assert cs['n_chunks'] == 3
    assert all(g.total_iters > 0 for g in got)
    bad = [i for i in range(300) if not _same(got[i], ref[i])]
    assert not bad, bad[:10]

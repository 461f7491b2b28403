"""The three stages around and including the physics solve, in memory, for a batch of clips:

    BVH + floor + contacts --prepare_input--> SeqInput --PhysOptim (HIP)--> snapshots --apply_results (HIP IK)--> BVH

The reference runs them as three child processes per video that talk through text files
(scripts/run_phys_mocap.py:137-201).  `run_phys_mocap --prepare --out-bvh` keeps those files (they are the drop-in
contract of the physics stage); this module is the same chain without them -- one solver launch sequence and one IK launch
sequence for all clips -- for deployments where the files would be the bottleneck (thousands of
sequences per run).  Values handed from stage to stage are the full doubles, not the 10-significant-digit text of
`sol_out_*.txt`, so results agree with the file path to ~1e-9 relative, not bit for bit.

Both solvers are passed in (`PhysOptim`, `IkBackProject`: HIP libraries, no CPU path).
"""
from dataclasses import dataclass, field
from typing import List, Optional, Sequence

import numpy as np

from . import apply_results as ar
from . import prepare_input as pi
from . import skeleton_io as sk

SNAPSHOT_KINDS = ('no_dynamics', 'dynamics', 'durations')        # sol_out_<kind>.txt, phys_optim.cpp:601, :659, :757


@dataclass
class Clip:
    bvh: str                                  # animation to improve (kinematic_results/<character>_out.bvh)
    floor: object                             # path of floor_out.txt, or (normal, point) already in the solver's frame
    contacts: object                          # path of foot_contacts.npy, or an F x 4 array (left heel, left toe, right heel, right toe)
    out_bvh: dict = field(default_factory=dict)      # snapshot kind -> output path; kinds that are absent are not back-projected
    start: Optional[int] = None
    end: Optional[int] = None


@dataclass
class ClipResult:
    seq: object                               # io_formats.SeqInput handed to the solver
    phys: object                              # phys_optim.SeqResult (snapshots, stage statuses, sizes)
    written: List[str] = field(default_factory=list)


def run_clips(clips: Sequence[Clip], character: ar.Character, phys, ik, dt=1.0 / 30.0, combined_contacts=False, prepare_device=None) -> List[ClipResult]:
    """`prepare_device` (e.g. 'cuda:0'): prepare_input's per-frame numerics of all clips as one batch of tensor operations on that device."""
    loaded = []
    seqs = []
    pending = []
    from . import prepare_capi
    parsed = prepare_capi.load_bvh_batch([c.bvh for c in clips])      # native reader: all files of the batch on the host's cores
    for c, (motion, names, _) in zip(clips, parsed):
        floor = pi.read_floor(c.floor) if isinstance(c.floor, str) else c.floor
        contacts = np.load(c.contacts) if isinstance(c.contacts, str) else np.asarray(c.contacts)
        start = 0 if c.start is None else c.start
        end = motion.n_frames if c.end is None else c.end
        if prepare_device is None:
            seqs.append(pi.prepare_sequence(motion, floor, contacts, character, start, end, dt, combined_contacts))
        else:
            pending.append((motion, floor, contacts, start, end))
        loaded.append((motion, names, start, end))
    if pending:
        seqs = pi.prepare_sequences_device([p[0] for p in pending], [p[1] for p in pending], [p[2] for p in pending], character, [p[3] for p in pending],
                                           [p[4] for p in pending], dt, combined_contacts, device=prepare_device)
    results, _ = phys.solve(seqs)                                                     # one batched launch sequence
    out = [ClipResult(seq=s, phys=r) for s, r in zip(seqs, results)]
    tasks, dest = [], []
    for k, (c, r) in enumerate(zip(clips, results)):
        motion, names, start, end = loaded[k]
        for kind, path in c.out_bvh.items():
            snap = r.snapshots[SNAPSHOT_KINDS.index(kind)]
            if np.asarray(snap.base_lin).shape[0] < end - start:                      # stage not reached (phys_optim.cpp:655, :709)
                continue
            tasks.append(ar.prepare(ar.to_animation_frame(snap, flip_coords=True), motion, names, start, end, character))
            dest.append((k, path))
    if tasks:
        ar.back_project(tasks, ik)                                                     # one batched launch sequence, all clips and kinds
    for t, (k, path) in zip(tasks, dest):
        ar.finish(t, path)
        out[k].written.append(path)
    return out

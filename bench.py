#!/usr/bin/env python
"""Benchmark of the physics hot path (BASELINE.json metric: physics-optimized 90-frame sequences/sec).

    python bench.py --gpus N --steps K --warmup W

One "step" = one full staged solve (stages 1.1, 1.2, 2.1, 2.2, 3 and the stage-4 fallback where stage 3 fails;
phys_optim.cpp:544-749, reference iteration caps and tolerance) of one batch of 128 synthetic 90-frame sequences
(BASELINE.json configs[1]).  Reference semantics: every stage runs until it converges, fails or reaches the reference's
iteration cap -- the solver's optional stall guard (chd_config.stall_window, not an IPOPT rule) is OFF in the measured
configuration; the same workload with the guard at 150 is timed afterwards and reported next to `value`
(`config.value_with_stall_guard_150`).  The K steps of the timed region are K DIFFERENT batches -- seeds
rank*K*128 .. (rank+1)*K*128 - 1, a one-GPU slice of configs[2] -- handed to the library in one call: inputs and
structure tables are resident in HBM when the clock starts, and ONE persistent launch (one resident workgroup per
compute unit taking sequences from a queue) drains them; one handle, one stream, no replicated batches.
N > 1: one process per GPU (launched by torch.distributed.run), sequences are independent, so there is no data-path
collective; weak scaling.
"""
import argparse
import json
import multiprocessing as mp
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

BATCH = 128
FRAMES = 90
LDS_PEAK_GBS = 256.0 * 256 * 2.4            # MI355X_MICROARCH.md (LDS): 256 B/clk/CU for ds_read_b64, 256 CUs, 2.4 GHz
HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
FP64_PEAK_TFLOPS = 78.6      # MI355X fp64 vector = matrix peak (AMD spec)
CAPS = [7000, 7000, 7000, 2500, 2000, 7000]


def kernel_sources_sha256():
    """Hash of the solver kernel's sources: profiles/traffic.json carries the one its PMC passes were measured with (tools/pmc_summary.py)."""
    return _sources_sha256(('chd_kernels.hpp', 'chd_phys.hip', 'chd_model.hpp', 'chd_device.hpp', 'chd_io.hpp'))      # = phys_optim.SOURCES (tests/test_capi.py checks that)


def _sources_sha256(files):
    import hashlib
    import re
    h = hashlib.sha256()
    for f in files:
        with open(os.path.join(ROOT, 'contact-human-dynamics_amd', 'csrc', f), 'r', errors='replace') as fh:
            for line in fh:                                  # code only: comments and blank lines do not make a measurement stale
                line = re.sub(r'//.*$', '', line).strip()
                if line:
                    h.update(line.encode() + b'\n')
    return h.hexdigest()


def _gen(args):
    import chd_amd  # noqa: F401
    from chd_amd.synth import make_walk
    seed0, n, frames = args[:3]
    out = [make_walk(seed=seed0 + i, F=frames, randomize=True) for i in range(n)]
    if len(args) > 3 and args[3]:                    # (the file-to-file side metric's input directories, written here: in parallel, before the timed regions)
        from chd_amd import io_formats as iof
        root, first = args[3], args[4]
        for i, sq in enumerate(out):
            iof.write_inputs(sq, os.path.join(root, 'v%05d' % (first + i), 'phys_optim_in_ybot'))
            os.makedirs(os.path.join(root, 'v%05d' % (first + i), 'phys_optim_out_ybot'), exist_ok=True)
    return out


def _gen_seeds(args):
    import chd_amd  # noqa: F401
    from chd_amd.synth import make_walk
    seeds, frames = args
    return [make_walk(seed=sd, F=frames, randomize=True) for sd in seeds]


def make_sequences_of(seeds, workers, frames=FRAMES):
    """The sequences of an explicit seed list (a rank's shard of the strong-scaling leg), generated on `workers` processes, order preserved."""
    seeds = list(seeds)
    if workers <= 1 or len(seeds) < 64:
        return _gen_seeds((seeds, frames))
    parts = [(seeds[a:a + 32], frames) for a in range(0, len(seeds), 32)]
    with mp.get_context('spawn').Pool(workers) as pool:          # (spawn: the HIP runtime exists in this process by now)
        out = pool.map(_gen_seeds, parts)
    return [s for p in out for s in p]


def make_sequences(seed0, n, workers, frames=FRAMES, write_root=None):
    """Seeds seed0 .. seed0 + n - 1 (synth.make_walk is a Python loop over frames: spread over a few processes)."""
    if workers <= 1 or n < 64:
        return _gen((seed0, n, frames, write_root, 0))
    chunk = 32
    parts = [(seed0 + a, min(chunk, n - a), frames, write_root, a) for a in range(0, n, chunk)]
    with mp.get_context('fork').Pool(workers) as pool:          # (forked before the HIP runtime is initialised)
        out = pool.map(_gen, parts)
    return [s for p in out for s in p]


def _cpu_worker(seed):
    """One sequence on the CPU oracle (the restated reference algorithm), single thread."""
    import chd_amd  # noqa: F401
    from chd_amd.synth import make_walk
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    from common import oracle_run
    seq = make_walk(seed=seed, F=FRAMES, randomize=True)
    t0 = time.time()
    stats, _ = oracle_run(seq, CAPS)
    return time.time() - t0, sum(s[1] for s in stats)


def cpu_baseline(max_workers=8):
    """Bounded sample of the same workload on the host cores: the first C sequences one per core (all-core rate), whose
    individual solve times also give the one-core rate."""
    from oracle import oracle
    oracle.build()
    cores = max(1, min(max_workers, os.cpu_count() or 1))
    t0 = time.time()
    with mp.get_context('spawn').Pool(cores) as pool:
        res = pool.map(_cpu_worker, list(range(cores)))
    wall = time.time() - t0
    one_core = len(res) / sum(r[0] for r in res)
    return {'value': cores / wall, 'unit': 'sequences/s', 'cores': cores, 'kind': 'port', 'value_one_core': one_core,
            'sample': 'first %d sequences of the workload (seeds 0..%d), one oracle process per core, %d IPM iterations, %.1f s wall; '
                      'value_one_core = sequences / sum of the per-sequence solve times' % (cores, cores - 1, sum(r[1] for r in res), wall),
            'note': 'the oracle is this repo\'s dense single-threaded restatement of the same algorithm, not IPOPT(MA57): the reference binary cannot be built here'}


def _emu_worker(seed):
    """One sequence through the host emulation of the KERNEL source (tests/host_emu: the same banded bordered L D L^T, the same iterations as the HIP path), one thread."""
    import chd_amd  # noqa: F401
    from chd_amd.synth import make_walk
    from chd_amd.phys_capi import default_config
    sys.path.insert(0, os.path.join(ROOT, 'tests', 'host_emu'))
    import emu
    seq = make_walk(seed=seed, F=FRAMES, randomize=True)
    t0 = time.time()
    e = emu.EmuProblem(seq, default_config(max_iter=CAPS))
    e.solve(0, 4)
    st, _ = e.results()
    if int(st[4][0]) != 0 and e.rebuild_fallback():
        e.solve(5, 5); st, _ = e.results()
    return time.time() - t0, int(sum(st[k][1] for k in range(6)))


def banded_emulation_baseline(n_seq=64):
    """The honest "same algorithm on the host" number: the kernel source compiled for the CPU (g++ -O2, one thread per sequence) on ALL
    host cores, bench seeds 0 .. n_seq - 1.  Test infrastructure used as a reported baseline, like the oracle."""
    sys.path.insert(0, os.path.join(ROOT, 'tests', 'host_emu'))
    import emu
    emu.build()
    cores = max(1, os.cpu_count() or 1)
    n_seq = max(n_seq, cores)
    t0 = time.time()
    with mp.get_context('spawn').Pool(min(cores, n_seq)) as pool:
        res = pool.map(_emu_worker, list(range(n_seq)), chunksize=1)
    wall = time.time() - t0
    return {'value': n_seq / wall, 'unit': 'sequences/s', 'cores': min(cores, n_seq), 'kind': 'port', 'value_one_core': len(res) / sum(r[0] for r in res),
            'sample': 'bench seeds 0..%d, one emulation process per core, %d IPM iterations, %.1f s wall' % (n_seq - 1, sum(r[1] for r in res), wall),
            'note': 'tests/host_emu: chd_kernels.hpp compiled with CHD_HOST_EMU for one CPU thread per sequence -- the banded bordered L D L^T and the iterations of the HIP path, not IPOPT(MA57)'}


def parity_block(results, seqs_first_seed):
    """HIP results of the bench workload against the committed oracle results (tests/golden/bench_parity_golden.npz, made by
    tests/golden/make_bench_parity_golden.py with the same caps / tolerance): computed outside the timed region."""
    import numpy as np
    path = os.path.join(ROOT, 'tests', 'golden', 'bench_parity_golden.npz')
    if not os.path.exists(path) or seqs_first_seed != 0:
        return None
    g = np.load(path)
    worst = 0.0; worst_failed = 0.0; n = 0; n_status = 0; n_iters = 0; n_contact = 0; failed = []; above = []; guarded = []
    for seed in range(min(BATCH, len(results))):
        key = 's%d_F%d_t000' % (seed, FRAMES)
        if key + '_status' not in g.files:
            continue
        r = results[seed]
        gs = list(g[key + '_status']); gi = list(g[key + '_iters'])
        if any(getattr(r, 'stage_stalled', [])):      # ended by the stall guard of the measured configuration: the fixture is made without the guard
            guarded.append(seed)
            continue
        n += 1
        st_eq = list(r.stage_status[:len(gs)]) == gs
        n_status += st_eq; n_iters += st_eq and list(r.stage_iters[:len(gi)]) == gi
        err = 0.0; c_ok = True
        for k in range(3):
            sn = r.snapshots[k]
            for name, val in (('base_lin', sn.base_lin), ('base_ang_deg', sn.base_ang_deg), ('ee_pos', sn.ee_pos), ('ee_force', sn.ee_force)):
                ref = g['%s_snap%d_%s' % (key, k, name)]
                if ref.shape != np.asarray(val).shape:
                    err = float('inf'); continue
                nr = float(np.linalg.norm(ref))
                if nr > 0:
                    err = max(err, float(np.linalg.norm(np.asarray(val) - ref)) / nr)
            c_ok = c_ok and np.array_equal(np.asarray(sn.contact), g['%s_snap%d_contact' % (key, k)])
        n_contact += c_ok
        if any(s != 0 for s in gs):          # a stage failed in the oracle: the iterates end on the merit function's noise floor
            failed.append(seed); worst_failed = max(worst_failed, err)
        else:
            worst = max(worst, err)
        if err > 1e-3:
            above.append(seed)
    return {'against': 'CPU oracle (tests/golden/bench_parity_golden.npz), not IPOPT: the reference binary cannot be built here',
            'sequences_compared': n, 'worst_rel_l2': worst, 'stage_status_equal': n_status, 'stage_iterations_equal': n_iters,
            'contact_flags_equal': n_contact, 'sequences_above_1e-3': above, 'not_compared_ended_by_stall_guard': guarded,
            'oracle_stage_failures': failed, 'worst_rel_l2_on_those': worst_failed}


def quality_block(batch, seqs, results, seed0):
    """The acceptance side of parity, on the 32 seeds of tests/golden/quality_golden.npz (= the first 32 sequences of the workload): objective,
    largest constraint violation and dynamics-row residual RECOMPUTED by the oracle's model at the points the kernel returned (chd_debug_get_state), against the
    converged (tol 1e-6) objective; and the distance of the ground reaction forces to the tol-1e-6 solution -- the output the stopping test pins least.  After the
    timed region; the oracle is the checker here, never the thing measured."""
    import numpy as np
    gq = os.path.join(ROOT, 'tests', 'golden', 'quality_golden.npz'); gf = os.path.join(ROOT, 'tests', 'golden', 'quality_forces_golden.npz')
    if seed0 != 0 or not os.path.exists(gq) or len(results) < 32 or seqs[0].F != FRAMES:
        return None
    from oracle.oracle import OracleProblem
    g = np.load(gq)
    n = len(g['seeds'])
    obj = np.zeros((n, 3)); vio = np.zeros((n, 3)); dyn = np.zeros((n, 3))
    for i in range(n):
        r = results[i]
        o = OracleProblem(seqs[i])
        last = 5 if r.stage_status[4] != 0 else 4
        for snap, stage in ((0, 1), (1, 3), (2, last)):
            xv, durs = batch.get_state(i, snap)
            e = o.eval_state(stage, xv, durs)
            obj[i, snap] = e['objective']; vio[i, snap] = e['violation']; dyn[i, snap] = e['dynamics_violation']
    ratio = obj / g['objective_converged']
    out = {'sequences': n, 'evaluator': 'oracle model functions at the points chd_debug_get_state returns (not the kernel\'s own statistics)',
           'objective_over_converged_median': [float(v) for v in np.median(ratio, axis=0)], 'objective_over_converged_max': [float(v) for v in ratio.max(axis=0)],
           'constraint_violation_max': float(vio.max()), 'dynamics_row_residual_max': float(dyn.max())}
    if os.path.exists(gf):
        f = np.load(gf)
        dist = np.zeros((n, 3, 2))
        for i in range(n):
            for k in range(3):
                for q, name in enumerate(('ee_force', 'base_lin')):
                    ref = f['s%d_snap%d_%s' % (i, k, name)]; got = np.asarray(getattr(results[i].snapshots[k], name))
                    nr = float(np.linalg.norm(ref))
                    dist[i, k, q] = float(np.linalg.norm(got - ref)) / nr if got.shape == ref.shape and nr > 0 else 0.0
        out['forces_vs_tol_1e-6_solution_rel_l2_median'] = [float(v) for v in np.median(dist[:, :, 0], axis=0)]
        out['forces_vs_tol_1e-6_solution_rel_l2_max'] = [float(v) for v in dist[:, :, 0].max(axis=0)]
        out['com_vs_tol_1e-6_solution_rel_l2_median'] = [float(v) for v in np.median(dist[:, :, 1], axis=0)]
        out['note'] = ('per snapshot (stages 1.2, 2.2, 3).  The forces of the tol-1e-3 solve are far from the converged ones -- the objective has no force term; north_star\'s '
                       '"GRFs within 1e-3 of the IPOPT reference" is NOT met by this proxy and is unmeasured against IPOPT itself')
    return out


def contact_net_rate(device, n_videos=128, frames=FRAMES, reps=20):
    """Second half of BASELINE.json's metric, "contact-net fps": the foot-contact MLP (contact_net.py, PyTorch-ROCm, fp32)
    on `n_videos` synthetic OpenPose sequences of `frames` frames -- (a) the forward pass alone with the windows resident
    on the device (one launch sequence over all windows of all videos), (b) the whole detector: host pre-processing,
    upload, forward, download, vote merge.  Frames per second = videos x frames / time.  Outside the timed region of
    the physics metric; reported next to it, never part of `value`."""
    import numpy as np
    import torch
    from chd_amd import contact_net as cn
    torch.manual_seed(0)
    model = cn.randomize_batchnorm_stats(cn.OpenPoseModel(), seed=0).to(device).eval()
    vids = [cn.synthetic_keypoints(s, F=frames) for s in range(n_videos)]
    sync = torch.cuda.synchronize if device.type == 'cuda' else (lambda: None)
    # warm-up on the SAME shape as the timed calls (round 5's driver line timed one call whose 128-video GEMM shapes met the libraries for the first time inside the
    # clock: 331 k frames/s against 3.1 M on the builder's box), then the MEDIAN of five timed calls
    def timed_calls(fn, n=5):
        fn(vids, model, device); sync()
        ts = []
        for _ in range(n):
            t0 = time.perf_counter(); r = fn(vids, model, device); sync(); ts.append(time.perf_counter() - t0)
        return r, float(np.median(ts)), ts
    (labels, _), t_all, t_all_runs = timed_calls(cn.detect_contacts)
    x = torch.from_numpy(np.concatenate([cn.make_windows(np.asarray(v, dtype=np.float64)) for v in vids], axis=0)).to(device)
    with torch.no_grad():
        model(x); sync()
        t0 = time.perf_counter()
        for _ in range(reps):
            model(x)
        sync(); t_fwd = (time.perf_counter() - t0) / reps
    # the five-layer forward captured ONCE as a HIP graph (torch.cuda.graphs) and replayed: what the launch latency of the eager sequence costs at this size
    t_graph = None; graph_note = None
    if device.type == 'cuda':
        try:
            with torch.no_grad():
                xs = x.clone()
                side = torch.cuda.Stream()
                side.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(side):
                    for _ in range(3):
                        model(xs)
                torch.cuda.current_stream().wait_stream(side); sync()
                gph = torch.cuda.CUDAGraph()
                with torch.cuda.graph(gph):
                    ys = model(xs)
                gph.replay(); sync()
                y_eager = model(x); sync()
                graph_equal = bool(torch.equal(ys, y_eager))
                t0 = time.perf_counter()
                for _ in range(reps):
                    gph.replay()
                sync(); t_graph = (time.perf_counter() - t0) / reps
                graph_note = 'logits bit-identical to the eager forward: %s' % graph_equal
        except Exception as exc:
            graph_note = 'graph capture failed: %s: %s' % (type(exc).__name__, exc)
    nfr = n_videos * frames
    wbytes = sum(p.numel() * p.element_size() for p in model.parameters()) + sum(b.numel() * b.element_size() for b in model.buffers())
    abytes = wbytes + int(x.numel()) * 4 + int(x.shape[0]) * 20 * 4      # weights once per launch sequence + 351 floats in, 20 out per window
    out = {'fps': nfr / t_fwd, 'fps_hip_graph': (nfr / t_graph) if t_graph else None, 'hip_graph_note': graph_note,
           'fps_end_to_end': nfr / t_all, 'fps_end_to_end_runs': [nfr / t for t in t_all_runs], 'timing': 'warm-up on the same 128-video shape, median of 5 timed calls (end-to-end paths); mean of %d launches (forward)' % reps,
           'unit': 'frames/s', 'videos': n_videos, 'frames': frames,
           'roofline': {'bound': 'hbm', 'achieved': abytes / t_fwd / 1e9, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': abytes / t_fwd / 1e9 / HBM_PEAK_GBS,
                        'algorithmic_bytes_per_forward': abytes, 'weight_bytes': wbytes, 'flops_per_forward': 2 * 0.954e6 * int(x.shape[0]),
                        'note': 'forward pass of all windows of all videos (library GEMMs of PyTorch-ROCm) against the HBM peak; the binding roofline of this size is the fp32 matrix core: roofline_mfma'},
           # 0.954 MMAC per window in exact fp32 (gfx950 has no TF32): at 10 496 windows the five GEMMs are bound by the fp32 matrix rate, not by bytes or launches
           # (replayed as one HIP graph the forward is no faster: fps_hip_graph)
           'roofline_mfma': {'bound': 'mfma', 'achieved': 2 * 0.954e6 * int(x.shape[0]) / t_fwd / 1e12, 'peak': 157.3, 'unit': 'TFLOP/s', 'frac': 2 * 0.954e6 * int(x.shape[0]) / t_fwd / 1e12 / 157.3,
                             'note': 'fp32 in / fp32 accumulate; peak = dense fp32 MFMA of MI355X_MICROARCH.md'},
           'windows': int(x.shape[0]), 'dtype': 'f32', 'device': str(device),
           'note': 'fps: forward pass, windows resident on the device; fps_end_to_end: NumPy pre-processing + upload + forward + vote merge; '
                   'fps_end_to_end_device_ops: the same with gap interpolation, windowing and vote merge as tensor ops on the device'}
    try:                                                                          # device-side pre/post-processing
        (labels_d, _), t_dev, t_dev_runs = timed_calls(cn.detect_contacts_device)
        out['fps_end_to_end_device_ops'] = nfr / t_dev
        out['fps_end_to_end_device_ops_runs'] = [nfr / t for t in t_dev_runs]
        out['device_ops_labels_equal'] = bool(all(np.array_equal(a, b) for a, b in zip(labels, labels_d)))
    except Exception as exc:
        out['fps_end_to_end_device_ops'] = None
        out['device_ops_error'] = '%s: %s' % (type(exc).__name__, exc)
    return out


def kinematic_optimisation_rate(device_index, n_clips=256, frames=100):
    """Next row in front of the physics stage (DESIGN.md "Rank 3"): the reference's `optimize_trajectory`
    for a batch of synthetic clips -- IK initialisation on libchd_ik.so, the two least-squares solves on libchd_kinopt.so, floor fit
    on the host -- and, as the parity figure, the three clips of the committed fixture against the REFERENCE's own results."""
    import numpy as np
    from chd_amd import kinematic_optimizer as kopt
    from chd_amd.synth import make_kin_clip
    g = np.load(os.path.join(ROOT, 'tests', 'golden', 'kinopt_golden.npz'))
    opt = kopt.KinematicOptimizer(device=device_index)
    gold = []
    for ci in range(int(g['n_cases'])):
        k = 'c%d_' % ci
        cl = dict(poses2D=g[k + 'poses2D'], joint_conf_2d=g[k + 'conf'], poses3D=g[k + 'poses3D'], root_pos=g[k + 'root_pos'], joint_angles=g[k + 'joint_angles'],
                  offsets=g[k + 'skel_offsets'], parents=g[k + 'skel_parents'], ppx=g[k + 'pp'][0], ppy=g[k + 'pp'][1], camFocal=g[k + 'focal'], velConstraints=g[k + 'vel'])
        if int(g[k + 'given_floor']):
            cl['plane_normal'] = g[k + 'floor_in_n']; cl['plane_point'] = g[k + 'floor_in_p']
        gold.append(cl)
    res = opt.optimize(gold)                       # (also the warm-up)
    worst = max(float(np.linalg.norm(r['pose3d'] - g['c%d_out_pose3d' % ci]) / np.linalg.norm(g['c%d_out_pose3d' % ci])) for ci, r in enumerate(res))
    contacts_equal = all(np.array_equal(r['velConstraints'], g['c%d_out_vel' % ci]) for ci, r in enumerate(res))
    clips = [make_kin_clip(s, frames, g['c0_skel_offsets'], g['c0_skel_parents']) for s in range(n_clips)]
    ms = []
    solve = opt.kin.solve

    def timed(problems):
        r = solve(problems)
        ms.append(opt.kin.last_kernel_ms())              # (thread-local in the library: this thread's launch)
        return r

    opt.kin.solve = timed
    t0 = time.perf_counter(); out = opt.optimize(clips); dt = time.perf_counter() - t0
    # up to 256 clips run as two halves on two host threads whose launches take turns on the device (round 5: the workgroups of a launch wait on each other, the library lets one
    # run at a time): the kernels' time is the SUM of the launches' device times (until round 4 the launches overlapped and it was the union of their intervals)
    kin_s = sum(ms) * 1e-3
    its = float(np.mean([sum(s['lsmr_iterations'] for s in r['stages']) for r in out]))
    n, m = 87 * frames, 507 * frames - 423
    alg = 8.0 * (2 * m + 8 * n + 2 * 420 * frames) * its * n_clips          # bytes per LSMR iteration x iterations
    traffic = None; tnote = 'not measured for this configuration'
    try:                                                                        # PMC passes of the same configuration (profiles/kinopt_traffic.json)
        tj = json.load(open(os.path.join(ROOT, 'profiles', 'kinopt_traffic.json')))
        if (tj['clips'], tj['frames']) == (n_clips, frames):
            traffic = tj['hbm_bytes_per_batch']; tnote = tj.get('note', '')
            if tj.get('sources_sha256') != _sources_sha256(('chd_kinopt.hip', 'chd_kinopt_kernels.hpp', 'chd_kinopt_host.hpp')):      # kernel, launch and the default LDS block
                tnote = 'STALE (kernel source changed since the PMC passes): ' + tnote
    except Exception:
        pass
    return {'clips': n_clips, 'frames': frames, 'clips_per_s': n_clips / dt, 'least_squares_kernel_ms': ms, 'ik_kernel_ms': opt.ik.last_kernel_ms()[0],
            'lsmr_iterations_per_clip': its, 'algorithmic_bytes_per_batch': alg,
            'least_squares_kernel_seconds': kin_s,
            'roofline': {'bound': 'lds', 'kernel': 'chd_kin_solve_kernel', 'achieved': alg / kin_s / 1e9, 'peak': LDS_PEAK_GBS, 'unit': 'GB/s',
                         'frac': alg / kin_s / 1e9 / LDS_PEAK_GBS, 'traffic': traffic, 'traffic_note': tnote,
                         'hbm_equivalent_frac': alg / kin_s / 1e9 / HBM_PEAK_GBS,
                         'definition': 'algorithmic bytes of all LSMR iterations of all clips (u, v, h, the linearisation: they live in the LDS of the cluster that solves a clip) / summed device time of the least-squares launches (they take turns on the device); '
                                       'peak = 256 B/clk/CU x 256 CUs x 2.4 GHz (MI355X_MICROARCH.md, LDS); `traffic` = HBM bytes of the same batch from the PMC passes; `hbm_equivalent_frac` = the same bytes against the HBM peak, for comparison with rounds 2-4 where they did come from HBM'},
            'fixture_worst_rel_l2_vs_reference': worst, 'fixture_contacts_equal_reference': bool(contacts_equal),
            'note': 'outside the timed region; the whole optimize() of %d clips x %d frames (IK initialisation, two least-squares solves, host floor fits)' % (n_clips, frames)}


def _self_rank(rank, world, port, argv, solver_factory):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), LOCAL_WORLD_SIZE=str(world))
    main(argv, solver_factory)


def _self_launch(world, argv, solver_factory):
    """`python bench.py --gpus N` from a plain shell: start the N ranks here (one process per GPU, rendezvous on 127.0.0.1), exactly what
    `python -m torch.distributed.run --nproc-per-node N bench.py --gpus N` would have started.  Rank 0's JSON line goes to this process's stdout."""
    import socket
    import torch.multiprocessing as tmp
    with socket.socket() as sk:
        sk.bind(('127.0.0.1', 0)); port = sk.getsockname()[1]
    tmp.start_processes(_self_rank, args=(world, port, list(argv) if argv is not None else sys.argv[1:], solver_factory), nprocs=world, join=True, start_method='spawn')


def main(argv=None, solver_factory=None):
    """`solver_factory` (tests only): a stand-in for PhysOptim on a box without a GPU -- the multi-rank plumbing (process group, barrier,
    the three all-reduces, rank 0's JSON line) then runs on the gloo backend with CPU tensors (tests/test_bench_multirank.py)."""
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=12)
    ap.add_argument('--warmup', type=int, default=1)
    ap.add_argument('--batch', type=int, default=BATCH, help=argparse.SUPPRESS)
    ap.add_argument('--stall-window', type=int, default=0,
                    help='chd_config.stall_window of the measured configuration (0 = off: a stagnating stage runs to its iteration cap)')
    ap.add_argument('--gen-workers', type=int, default=0, help=argparse.SUPPRESS)
    ap.add_argument('--max-workgroups', type=int, default=0, help=argparse.SUPPRESS)
    ap.add_argument('--threads', type=int, default=0, help=argparse.SUPPRESS)
    ap.add_argument('--lds-kb', type=int, default=0, help=argparse.SUPPRESS)
    ap.add_argument('--towr_phys_optim_path', default=os.environ.get('TOWR_PHYS_OPTIM_PATH', ''),
                    help='directory of a REFERENCE phys_optim binary (scripts/run_phys_mocap.py:26): if one is found there it is run on the first sequences of the workload, '
                         'compared with the HIP results and timed as the CPU baseline (kind "reference")')
    ap.add_argument('--no-cpu-baseline', action='store_true', help=argparse.SUPPRESS)
    ap.add_argument('--no-side-metrics', action='store_true', help=argparse.SUPPRESS)
    ap.add_argument('--worker', action='store_true', help=argparse.SUPPRESS)        # (accepted for old command lines; no effect)
    ap.add_argument('--frames', type=int, default=FRAMES, help=argparse.SUPPRESS)     # (tests: shorter sequences for the CPU stand-in)
    ap.add_argument('--strong-total', type=int, default=4000, help='sequences of the strong-scaling leg (BASELINE configs[2]: ~2k motions x 2 cameras, LPT-sharded over the ranks); 0 = skip')
    args = ap.parse_args(argv)

    if 'WORLD_SIZE' not in os.environ and args.gpus > 1:          # no launcher around this process: be the launcher
        return _self_launch(args.gpus, argv, solver_factory)
    world = int(os.environ.get('WORLD_SIZE', '1'))
    if world != args.gpus:
        raise SystemExit('bench.py: --gpus %d, but the launcher started %d rank(s) (WORLD_SIZE): one process per GPU' % (args.gpus, world))
    rank = int(os.environ.get('RANK', '0'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    B = args.batch
    steps = max(1, args.steps)
    workers = args.gen_workers if args.gen_workers > 0 else max(1, min(8, (os.cpu_count() or 1) // max(1, world)))
    seed0 = rank * steps * B
    dirs_root = None
    if world == 1 and not args.no_side_metrics and solver_factory is None:
        import tempfile
        dirs_root = tempfile.mkdtemp(prefix='chd_bench_dirs_')          # inputs of the file-to-file side metric (removed at the end)
    seqs = make_sequences(seed0, steps * B, workers, args.frames, dirs_root)         # before the HIP runtime exists in this process (fork)

    import torch
    import torch.distributed as dist
    import chd_amd  # noqa: F401
    from chd_amd.phys_optim import PhysOptim, default_config

    on_gpu = solver_factory is None
    if on_gpu and not torch.cuda.is_available():
        raise SystemExit('bench.py needs a GPU: the physics stage has no CPU path')
    tdev = 'cuda' if on_gpu else 'cpu'
    if on_gpu:
        torch.cuda.set_device(local)
    if world > 1:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        if on_gpu:
            dist.init_process_group('nccl', device_id=torch.device('cuda', local))
        else:
            dist.init_process_group('gloo')
        assert dist.get_world_size() == args.gpus, (dist.get_world_size(), args.gpus)

    def barrier():
        if world > 1:
            dist.barrier()
        if on_gpu:
            torch.cuda.synchronize()

    cfg = default_config(stall_window=args.stall_window, max_workgroups=args.max_workgroups, threads_per_sequence=args.threads, lds_kilobytes=args.lds_kb)   # reference caps and tol
    solver = (PhysOptim if on_gpu else solver_factory)(device=local, config=cfg)
    batch = solver.upload(seqs)                                           # inputs + tables -> HBM (not timed)
    if args.warmup > 0:                                                   # W untimed steps: the first W batches
        wb = solver.upload(seqs[:min(len(seqs), args.warmup * B)])
        wb.solve(); wb.free()
    barrier()
    t0 = time.perf_counter()
    st = batch.solve()                                                    # the K steps: returns when the queue is drained
    barrier()
    elapsed = time.perf_counter() - t0
    kernel_ms = st['kernel_ms'][0] + st['kernel_ms'][1]
    alg_bytes = st['alg_bytes']; iters = st['total_iters']; nfact = st['total_factorizations']
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=tdev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        agg = torch.tensor([float(iters), float(alg_bytes)], dtype=torch.float64, device=tdev)
        dist.all_reduce(agg, op=dist.ReduceOp.SUM)
        tot_iters_all, alg_bytes_all = float(agg[0].item()), float(agg[1].item())
    else:
        tot_iters_all, alg_bytes_all = float(iters), float(alg_bytes)

    # ---- strong-scaling leg (BASELINE configs[2]: the full synthetic set, ~2k motions x 2 cameras = 4 000 sequences, sharded over the ranks as
    # run_phys_mocap.py shards its videos: sharding.lpt_assign by frame count -- an even split for equal lengths).  TOTAL work is fixed as N grows; every rank
    # uploads its shard beforehand and solves it in one call; MAX over ranks.  Reported beside the weak-scaling `value`, never instead of it.
    strong = None
    if args.strong_total > 0:
        from chd_amd.sharding import lpt_assign
        mine = lpt_assign([args.frames] * args.strong_total, world)[rank]
        sseqs = make_sequences_of([1000000 + i for i in mine], workers if solver_factory is None else 1, args.frames)
        sb = solver.upload(sseqs)
        barrier()
        t1 = time.perf_counter()
        sst = sb.solve()
        barrier()
        dt_s = time.perf_counter() - t1
        s_iters = float(sst['total_iters']); s_bytes = float(sst['alg_bytes'])
        if world > 1:
            t = torch.tensor([dt_s], dtype=torch.float64, device=tdev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt_s = float(t.item())
            agg = torch.tensor([s_iters, s_bytes], dtype=torch.float64, device=tdev)
            dist.all_reduce(agg, op=dist.ReduceOp.SUM)
            s_iters, s_bytes = float(agg[0].item()), float(agg[1].item())
        sb.free()
        strong = {'scaling': 'strong', 'total_sequences': args.strong_total, 'sequences_rank0': len(mine), 'n_gpus': world, 'seconds': dt_s,
                  'value': args.strong_total / dt_s, 'unit': 'sequences/s', 'ipm_iterations_per_sequence': s_iters / args.strong_total,
                  'roofline_frac': s_bytes / dt_s / 1e9 / (HBM_PEAK_GBS * world), 'kernel_busy_fraction_rank0': sst['phase_ms'][5] / max(1e-9, sst['n_workgroups'] * (sst['kernel_ms'][0] + sst['kernel_ms'][1])),
                  'workload': 'BASELINE configs[2]: %d synthetic %d-frame walks (seeds 1000000 ..), LPT-sharded over the ranks (sharding.lpt_assign), inputs resident, one call per rank, MAX over ranks' % (args.strong_total, args.frames)}

    res = batch.fetch()
    n_ok = sum(1 for r in res if r.dynamics_succeed and r.durations_succeed)
    sizes = res[0].sizes
    total_value_hint = B * steps / elapsed          # (solve-only sequences/s of this rank)

    def np_eq(a, b_):
        import numpy as _np
        return _np.array_equal(_np.asarray(a), _np.asarray(b_))

    # side runs (one GPU, outside the timed region of `value`): the whole call a user makes on the same sequences (set-up + upload + solve + fetch,
    # pipelined; and file to file), the stall guard, BASELINE configs[2]'s per-GPU slice (4 000 sequences / 8 GPUs = 500 in one call), configs[4] (600 frames)
    side = {}
    if world == 1 and not args.no_side_metrics:
        def timed_solve(sq, **kw):
            c2 = default_config(max_workgroups=args.max_workgroups, threads_per_sequence=args.threads, lds_kilobytes=args.lds_kb,
                                **{**dict(stall_window=args.stall_window), **kw})
            s2 = PhysOptim(device=local, config=c2)
            b2 = s2.upload(sq)
            torch.cuda.synchronize(); t1 = time.perf_counter(); st2 = b2.solve(); torch.cuda.synchronize(); dt2 = time.perf_counter() - t1
            b2.free(); s2.close()
            return len(sq) / dt2, st2
        try:
            # (a) chd_phys_solve_batch: what the reference does inside the process it would be timed as (phys_optim.cpp:428-540 set-up + solve + SaveSolution arrays)
            t1 = time.perf_counter(); solver.solve_batch(seqs); dt_cold = time.perf_counter() - t1          # first call: also allocates the pools' workspaces and staging buffers
            t1 = time.perf_counter(); res_incl, cs = solver.solve_batch(seqs); dt_py = time.perf_counter() - t1
            dt_incl = cs['wall_ms'] * 1e-3                                                                   # the C ABI call itself (the ctypes marshalling of 2 560 sequences is Python's)
            side['value_including_setup'] = len(seqs) / dt_incl
            side['setup_ms_per_sequence'] = cs['setup_cpu_ms'] / max(1, cs['n_sequences'])
            side['including_setup'] = {'seconds': dt_incl, 'seconds_with_python_marshalling': dt_py, 'seconds_first_call_allocating_pools': dt_cold, 'host_threads': cs['host_threads'], 'chunks': cs['n_chunks'], 'chunk': cs['chunk'],
                                       'setup_wall_ms': cs['setup_wall_ms'], 'upload_ms': cs['upload_ms'], 'host_waited_for_device_ms': cs['wait_for_pool_ms'],
                                       'kernel_ms_sum_over_chunks': cs['kernel_ms'], 'iterations': cs['total_iters'], 'fallbacks': cs['n_fallback'],
                                       'identical_to_split_interface': bool(all(a.stage_iters == b_.stage_iters and all(np_eq(x.base_lin, y.base_lin) for x, y in zip(a.snapshots, b_.snapshots))
                                                                               for a, b_ in zip(res_incl, res))),
                                       'host_cores_to_keep_one_gpu_busy': cs['setup_cpu_ms'] / max(1e-9, 1e3 * len(seqs) / (total_value_hint or 1.0)),
                                       'note': 'chd_phys_solve_batch on the SAME %d sequences: table build on the host threads, upload, persistent launches (up to 4 chunks in flight), '
                                               'stage-4 fallbacks, fetch -- one call, wall clock; host_cores_to_keep_one_gpu_busy = set-up thread-seconds per second of solve-only rate' % len(seqs)}
            # (b) the same file to file ("I/O-inclusive figure reported separately"): chd_phys_solve_dirs on the workload's directories
            import shutil
            nd = len(seqs)                                                # the SAME sequences as `value` (until round 5: the first 1 024 -- four per compute unit, whose tail was mistaken for an I/O cost)
            root = dirs_root
            try:
                ins = [os.path.join(root, 'v%05d' % i, 'phys_optim_in_ybot') for i in range(nd)]
                outs_ = [os.path.join(root, 'v%05d' % i, 'phys_optim_out_ybot') for i in range(nd)]
                t1 = time.perf_counter(); stt = solver.solve_dirs(ins, outs_, [args.frames] * nd); dt_io = time.perf_counter() - t1
                cs2 = solver.call_stats()
                side['value_including_file_io'] = nd / dt_io
                side['including_file_io'] = {'directories': nd, 'seconds': dt_io, 'failures': int(sum(1 for v in stt if v != 0)), 'read_ms': cs2['prep_ms'], 'write_ms_overlapped': cs2['finish_ms'],
                                             'note': 'chd_phys_solve_dirs: 4 input files parsed + set-up + solve + 4 output files written per directory (tmpfs/page cache), one call'}
            finally:
                shutil.rmtree(root, ignore_errors=True)
        except Exception as exc:
            side['inclusive_run_error'] = '%s: %s' % (type(exc).__name__, exc)
        try:
            v, st2 = timed_solve(seqs, stall_window=150)
            side['value_with_stall_guard_150'] = v; side['stall_guard_150_hits'] = st2['n_stalled']; side['stall_guard_150_fallbacks'] = st2['n_fallback']
            v, st2 = timed_solve(seqs[:128])
            side['value_128_sequences_in_one_call'] = v           # BASELINE configs[1] taken literally: ONE batch of 128 in a call of its own (half the compute units idle, the call lasts as long as its slowest sequence)
            side['slowest_sequence_ms_128'] = st2['max_seq_ms']
            v, st2 = timed_solve(seqs[:500])
            side['value_500_sequences_in_one_call'] = v
            # what solver-INDEPENDENT positions cost (tests/golden/cross_solver_golden.json: centre of mass / angles / feet of two converged solvers agree to 1e-5, at the
            # reference's tol 1e-3 they depend on the iterates): the first 128 sequences with every stage run to tol 1e-6 -- NOT the reference's setting, never `value`
            n6 = min(128, len(seqs))
            v, st2 = timed_solve(seqs[:n6], tol=1e-6)
            side['value_at_tol_1e-6'] = v
            side['at_tol_1e-6'] = {'sequences': n6, 'seconds': n6 / v, 'ipm_iterations_per_sequence': st2['total_iters'] / max(1, n6), 'fallbacks': st2['n_fallback'], 'slowest_sequence_ms': st2['max_seq_ms'],
                                   'mean_sequence_ms': st2['phase_ms'][5] / max(1, n6),
                                   'note': 'every stage to tol 1e-6 instead of the reference\'s 1e-3 (phys_optim.cpp:578); one call, inputs resident'}
            side['kernel_busy_fraction_500_sequences'] = st2['phase_ms'][5] / max(1e-9, st2['n_workgroups'] * (st2['kernel_ms'][0] + st2['kernel_ms'][1]))
            # BASELINE configs[4]: one 600-frame sequence on a 10-degree floor, alone in a launch
            sys.path.insert(0, os.path.join(ROOT, 'tests', 'golden'))
            import make_bench_parity_golden as mk
            long_seq = mk.make_case(0, 600, 10.0)
            s3 = PhysOptim(device=local, config=default_config())
            t1 = time.perf_counter(); b3 = s3.upload([long_seq]); t_up = time.perf_counter() - t1
            st3 = b3.solve(); r3 = b3.fetch()[0]; b3.free(); s3.close()
            side['long_600_frames'] = {'upload_seconds_incl_table_build': t_up, 'kernel_ms': st3['kernel_ms'][0] + st3['kernel_ms'][1], 'iterations': st3['total_iters'],
                                       'kkt_dim': r3.sizes['kkt_dim'], 'halfband': r3.sizes['halfband'], 'border': r3.sizes['border'], 'stage_status': list(r3.stage_status),
                                       'time_share': {k: st3['phase_ms'][i] / max(1e-9, st3['phase_ms'][5]) for k, i in (('evaluation_full', 0), ('evaluation_values', 1), ('factorisation', 2), ('substitution', 3), ('kkt_matvec', 4))}}
            # chd_config.damping_rule = 1 (round 6, off by default: it costs a launch of thousands of walks 7 %): what it buys where few or long sequences share a call
            v, st2 = timed_solve(seqs[:500], damping_rule=1)
            s4 = PhysOptim(device=local, config=default_config(damping_rule=1))
            b4 = s4.upload([long_seq]); st4 = b4.solve(); b4.free(); s4.close()
            side['damping_rule_1'] = {'value_500_sequences_in_one_call': v, 'ipm_iterations_per_sequence_500': st2['total_iters'] / 500.0,
                                      'long_600_frames_kernel_ms': st4['kernel_ms'][0] + st4['kernel_ms'][1], 'long_600_frames_iterations': st4['total_iters'],
                                      'note': 'chd_config.damping_rule = 1: the damping also grows after an ACCEPTED step that delivered < 1/4 of the predicted merit reduction; NOT the configuration of '
                                              '`value` (2 560 walks in one launch lose 7 % with it: profiles/r06_globalisation_study.md)'}
        except Exception as exc:
            side['side_run_error'] = '%s: %s' % (type(exc).__name__, exc)

    if rank == 0:
        import numpy as np
        total_seqs = world * B * steps
        # sum of algorithmic bytes (bytes_iter(seq) x iterations, all ranks) / wall time of the timed region
        ach = alg_bytes_all / elapsed / 1e9
        traffic = None; traffic_note = 'not measured for this build'; mfma = None; traffic_raw = None; wait_frac = None
        tfile = os.path.join(ROOT, 'profiles', 'traffic.json')
        if os.path.exists(tfile):
            try:
                tj = json.load(open(tfile))
                traffic = tj.get('hbm_bytes_per_step'); traffic_raw = tj.get('hbm_bytes_per_step_raw'); wait_frac = tj.get('sq_wait_any_fraction')
                if tj.get('mfma_tflops') is not None:          # COUNTED matrix-core instructions of the PMC pass (not a formula: the by-formula figure of rounds 1-4 counted the dense band, 12 x the envelope's flops)
                    mfma = {'achieved': tj['mfma_tflops'], 'peak': FP64_PEAK_TFLOPS, 'unit': 'TFLOP/s', 'frac': tj['mfma_tflops'] / FP64_PEAK_TFLOPS,
                            'note': 'v_mfma_f64_16x16x4_f64 counted by SQ_INSTS_VALU_MFMA_MOPS_F64 in the PMC pass of profiles/traffic.json; only the trailing update of the factorisation runs there'}
                traffic_note = tj.get('note', '')
                if tj.get('sources_sha256') != kernel_sources_sha256():      # the PMC passes were made with other kernel sources than the ones that just ran
                    traffic_note = 'STALE (kernel sources changed since the PMC passes): ' + traffic_note
            except Exception:
                traffic = None
        it_seq = np.array([r.total_iters for r in res], dtype=np.float64)
        out = {
            'metric': 'physics-optimized sequences/sec (90-frame)', 'value': total_seqs / elapsed, 'unit': 'sequences/s',
            'n_gpus': (dist.get_world_size() if world > 1 else 1), 'steps': steps, 'warmup': args.warmup, 'ms_per_step': 1e3 * elapsed / steps,
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f64', 'data': 'synthetic',
            'config': {'workload': '%d batches (steps) of %d synthetic Mixamo-like %d-frame walks per GPU (BASELINE configs[1]; seeds rank*%d .. : a slice of configs[2]), '
                                   'staged NLP solve, reference iteration caps 7000/7000/7000/2500/2000/7000, tol 1e-3, stall guard %s'
                                   % (steps, B, args.frames, steps * B, 'off' if args.stall_window <= 0 else 'window %d' % args.stall_window),
                       'sequences_per_gpu': B * steps, 'frames': args.frames,
                       'parallelism': 'independent sequences; %d process(es); per GPU one persistent launch of %d resident workgroups (1 per CU) draining a queue; one handle, one stream'
                                      % (world, st['n_workgroups']),
                       'kkt_dim': sizes['kkt_dim'], 'halfband': sizes['halfband'], 'border': sizes['border'], 'nnz_jac': sizes['nnz_jac'],
                       'ipm_iterations_per_sequence': tot_iters_all / total_seqs,
                       'ipm_iterations_rank0': {'p50': float(np.percentile(it_seq, 50)), 'p90': float(np.percentile(it_seq, 90)), 'max': float(it_seq.max())},
                       'factorizations_rank0': nfact, 'stage4_fallbacks_rank0': st['n_fallback'], 'stall_guard_hits_rank0': st['n_stalled'],
                       'converged_rank0': '%d/%d' % (n_ok, len(res)), 'factorisation': 'right-looking bordered band L D L^T, matrix-core trailing update', **side,
                       'slowest_sequence_ms': st['max_seq_ms'], 'mean_sequence_ms': st['phase_ms'][5] / max(1, len(res)),
                       'in_kernel_phase_ms_per_sequence': [round(v / max(1, len(res)), 3) for v in st['phase_ms']],
                       'in_kernel_time_share': {k: st['phase_ms'][i] / max(1e-9, st['phase_ms'][5]) for k, i in
                                                (('evaluation_full', 0), ('evaluation_values', 1), ('factorisation', 2), ('substitution', 3), ('kkt_matvec', 4),
                                                 ('factor_copy', 6), ('factor_panel_load', 8), ('factor_row_solve', 9), ('factor_lookahead_wavefront', 11),
                                                 ('factor_store_and_wait_for_trailing_tiles', 10), ('factor_border', 12))}},
            'roofline': {'bound': 'hbm', 'achieved': ach, 'peak': HBM_PEAK_GBS * world, 'unit': 'GB/s', 'frac': ach / (HBM_PEAK_GBS * world),
                         'definition': 'sum over sequences of SURVEY 8(d) bytes_iter x IPM iterations / wall time of the timed region / (8 TB/s x GPUs)',
                         'algorithmic_bytes_per_step': alg_bytes_all / steps / world, 'algorithmic_bytes_per_iteration': alg_bytes / max(1, iters),
                         'traffic': traffic, 'traffic_raw': traffic_raw, 'traffic_note': traffic_note, 'sq_wait_any_fraction': wait_frac, 'kernel': 'chd_solve_kernel',
                         'achieved_over_kernel_events_rank0': alg_bytes / max(1e-9, kernel_ms * 1e-3) / 1e9,      # rank 0's algorithmic bytes / its launches' duration by the library's HIP events (main + fallback launch) -- agrees with `achieved` (wall clock, all ranks) to the launch overhead
                         'launches': 1 + (1 if st['kernel_ms'][1] > 0 else 0), 'kernel_ms_rank0': kernel_ms, 'fallback_launch_ms_rank0': st['kernel_ms'][1],
                         'kernel_busy_fraction': st['phase_ms'][5] / max(1e-9, st['n_workgroups'] * kernel_ms),
                         'fp64_mfma': mfma},
        }
        if strong is not None:
            out['strong_scaling'] = strong
        for k in ('value_including_setup', 'value_including_file_io', 'value_128_sequences_in_one_call', 'value_500_sequences_in_one_call', 'value_at_tol_1e-6'):      # what a caller sees, first-class (also under config)
            if k in side:
                out[k] = side[k]
        if world == 1 and not args.no_side_metrics:
            try:
                out['parity'] = parity_block(res, seed0)
                if out['parity'] is not None:
                    out['parity']['acceptance'] = quality_block(batch, seqs, res, seed0)
                    vf = os.path.join(ROOT, 'tests', 'golden', 'ipopt_like_golden.json')
                    if os.path.exists(vf):          # committed study (tests/golden/make_ipopt_like_golden.py): the shipped algorithm against the oracle's IPOPT-like mode -- an explicit PROXY for the unmeasurable "vs IPOPT"
                        vj = json.load(open(vf))
                        out['parity']['vs_ipopt_like'] = {k: vj[k] for k in ('what', 'proxy_for', 'snapshots', 'quantities', 'generator', 'summary') if k in vj}
                    cf = os.path.join(ROOT, 'tests', 'golden', 'cross_solver_golden.json')
                    if os.path.exists(cf):          # committed study (tests/tools/cross_solver_convergence.py, tests/test_cross_solver.py): shipped algorithm vs SciPy trust-constr, both CONVERGED, same start
                        cj = json.load(open(cf))
                        out['parity']['cross_solver'] = {'what': cj['what'], 'generator': cj['generator'], 'summary': cj['summary'],
                                                         'finding': 'where two independent solvers reach the same objective (1e-5), centre of mass / base angles / feet agree to <= 3e-5 / 3e-4 / 6e-6 (max over sequences) '
                                                                    'and the net ground reaction force to 3e-2 in the median: the NLP does not determine the forces better than that (no force term in the cost), '
                                                                    'so north_star\'s "GRFs within 1e-3 of the IPOPT reference" cannot hold between ANY two independent solvers, at any tolerance'}
            except Exception as exc:
                out['parity'] = {'error': '%s: %s' % (type(exc).__name__, exc)}
        ref_base = None
        if world == 1:
            try:
                sys.path.insert(0, os.path.join(ROOT, 'tests', 'tools'))
                import compare_with_reference as cwr
                out['reference_binary'], ref_base = cwr.reference_block(args.towr_phys_optim_path, seqs, res)
            except Exception as exc:
                out['reference_binary'] = {'status': 'not measured', 'reason': '%s: %s' % (type(exc).__name__, exc)}
        if world == 1 and not args.no_cpu_baseline:
            out['cpu_baseline'] = cpu_baseline()
            try:
                out['cpu_baseline']['banded_emulation'] = banded_emulation_baseline()
            except Exception as exc:
                out['cpu_baseline']['banded_emulation'] = {'error': '%s: %s' % (type(exc).__name__, exc)}
            if ref_base is not None:          # the reference's own binary, timed on this box: that is the CPU baseline; the oracle's rate stays beside it
                ref_base['oracle_port'] = out['cpu_baseline']; out['cpu_baseline'] = ref_base
        if world == 1 and not args.no_side_metrics:
            try:
                out['contact_net'] = contact_net_rate(torch.device('cuda', local))
            except Exception as exc:
                out['contact_net'] = {'error': '%s: %s' % (type(exc).__name__, exc)}
            try:
                out['kinematic_optimisation'] = kinematic_optimisation_rate(local)
            except Exception as exc:
                out['kinematic_optimisation'] = {'error': '%s: %s' % (type(exc).__name__, exc)}
            try:                                                                  # the whole chain, file to file (tests/tools/pipeline_bench.py)
                sys.path.insert(0, os.path.join(ROOT, 'tests', 'tools'))
                import pipeline_bench
                import contextlib
                with contextlib.redirect_stdout(sys.stderr):                       # the drivers print progress lines; stdout carries exactly one JSON line
                    out['pipeline'] = pipeline_bench.run(32, 60)
            except BaseException as exc:                                           # (the drivers end with SystemExit on bad arguments)
                out['pipeline'] = {'error': '%s: %s' % (type(exc).__name__, exc)}
        # scalar copies at the top level of what the previous review had to dig out of nested objects (a parser that keeps only first-level numbers keeps these)
        try:
            if 'banded_emulation' in out.get('cpu_baseline', {}) and 'value' in out['cpu_baseline']['banded_emulation']:
                out['cpu_baseline_banded_emulation'] = out['cpu_baseline']['banded_emulation']['value']
                out['cpu_baseline_banded_emulation_cores'] = out['cpu_baseline']['banded_emulation'].get('cores')
            if strong is not None:
                out['strong_scaling_value'] = strong.get('value')
            if 'long_600_frames' in side:
                out['long_600_frames_kernel_ms'] = side['long_600_frames'].get('kernel_ms')
            out['slowest_sequence_ms'] = out['config'].get('slowest_sequence_ms')
            if 'contact_net' in out and 'fps' in out['contact_net']:
                out['contact_net_fps'] = out['contact_net']['fps']
                out['contact_net_fps_hip_graph'] = out['contact_net'].get('fps_hip_graph')
                out['contact_net_fps_end_to_end_device_ops'] = out['contact_net'].get('fps_end_to_end_device_ops')
            if 'kinematic_optimisation' in out and 'clips_per_s' in out['kinematic_optimisation']:
                out['kinematic_optimisation_clips_per_s'] = out['kinematic_optimisation']['clips_per_s']
            if 'pipeline' in out and 'videos_per_s' in out['pipeline']:
                out['pipeline_videos_per_s'] = out['pipeline']['videos_per_s']
        except Exception as exc:
            out['top_level_copies_error'] = '%s: %s' % (type(exc).__name__, exc)
        print(json.dumps(out), flush=True)
    batch.free()
    solver.close()
    if dirs_root:
        import shutil
        shutil.rmtree(dirs_root, ignore_errors=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()

#!/usr/bin/env python
"""Benchmark of the physics hot path (BASELINE.json metric: physics-optimized 90-frame sequences/sec).

    python bench.py --gpus N --steps K --warmup W

One "step" = one full staged solve (stages 1.1, 1.2, 2.1, 2.2, 3 and the stage-4 fallback where
stage 3 fails; phys_optim.cpp:544-749, reference iteration caps) of one batch of 128 synthetic
90-frame sequences per GPU (BASELINE.json configs[1]; seeds rank*128 .. rank*128+127), with the
inputs and the structure tables already resident in HBM.  N > 1: one process per GPU (launched by
torch.distributed.run), sequences are independent so there is no data-path collective; weak scaling.
"""
import argparse
import json
import multiprocessing as mp
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# One HIP stream per step in flight; the runtime multiplexes streams onto this many hardware queues (its default of 4
# would cap the number of concurrently running launches at 4).  Must be set before the HIP runtime initialises.
# Every queue gets its own scratch arena sized for the whole device: queues x scratch-bytes-per-lane of the kernel is
# bounded (measured on MI355X: 16 queues x 4288 B/lane works, 16 x 4400 and 24 x 4288 abort with
# HSA_STATUS_ERROR_OUT_OF_RESOURCES), hence 12 and not 16 for the present kernel (4400 B/lane).
DEFAULT_IN_FLIGHT = 12
os.environ.setdefault('GPU_MAX_HW_QUEUES', str(DEFAULT_IN_FLIGHT))

BATCH = 128
FRAMES = 90
HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def _cpu_worker(seed):
    """One sequence on the CPU oracle (the restated reference algorithm), single thread."""
    import chd_amd  # noqa: F401
    from chd_amd.synth import make_walk
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    from common import oracle_run
    seq = make_walk(seed=seed, F=FRAMES, randomize=True)
    t0 = time.time()
    stats, _ = oracle_run(seq, [7000, 7000, 7000, 2500, 2000, 7000])
    return time.time() - t0, sum(s[1] for s in stats)


def cpu_baseline(max_workers=8):
    """Bounded sample of the same workload on the host cores: the first C sequences, one per core."""
    from oracle import oracle
    oracle.build()
    cores = max(1, min(max_workers, os.cpu_count() or 1))
    t0 = time.time()
    with mp.get_context('spawn').Pool(cores) as pool:
        res = pool.map(_cpu_worker, list(range(cores)))
    wall = time.time() - t0
    return {'value': cores / wall, 'unit': 'sequences/s', 'cores': cores, 'kind': 'port',
            'sample': 'first %d sequences of the workload (seeds 0..%d), one oracle process per core, %d IPM iterations, %.1f s wall'
                      % (cores, cores - 1, sum(r[1] for r in res), wall)}


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
    cn.detect_contacts(vids[:2], model, device)                                   # warm-up: kernels, allocator
    t0 = time.perf_counter(); labels, _ = cn.detect_contacts(vids, model, device); sync(); t_all = time.perf_counter() - t0
    x = torch.from_numpy(np.concatenate([cn.make_windows(np.asarray(v, dtype=np.float64)) for v in vids], axis=0)).to(device)
    with torch.no_grad():
        model(x); sync()
        t0 = time.perf_counter()
        for _ in range(reps):
            y = model(x)
        sync(); t_fwd = (time.perf_counter() - t0) / reps
    nfr = n_videos * frames
    out = {'fps': nfr / t_fwd, 'fps_end_to_end': nfr / t_all, 'unit': 'frames/s', 'videos': n_videos, 'frames': frames,
           'windows': int(x.shape[0]), 'dtype': 'f32', 'device': str(device),
           'note': 'fps: forward pass, windows resident on the device; fps_end_to_end: NumPy pre-processing + upload + forward + vote merge; '
                   'fps_end_to_end_device_ops: the same with gap interpolation, windowing and vote merge as tensor ops on the device'}
    try:                                                                          # device-side pre/post-processing
        cn.detect_contacts_device(vids[:2], model, device)
        t0 = time.perf_counter(); labels_d, _ = cn.detect_contacts_device(vids, model, device); sync(); t_dev = time.perf_counter() - t0
        out['fps_end_to_end_device_ops'] = nfr / t_dev
        out['device_ops_labels_equal'] = bool(all(np.array_equal(a, b) for a, b in zip(labels, labels_d)))
    except Exception as exc:
        out['fps_end_to_end_device_ops'] = None
        out['device_ops_error'] = '%s: %s' % (type(exc).__name__, exc)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=2 * DEFAULT_IN_FLIGHT)       # two full rounds of the steps in flight
    ap.add_argument('--warmup', type=int, default=2)
    ap.add_argument('--batch', type=int, default=BATCH, help=argparse.SUPPRESS)
    ap.add_argument('--pipeline', type=int, default=DEFAULT_IN_FLIGHT,
                    help='steps in flight at once, each on its own HIP stream (1 = strictly one batch after the other)')
    ap.add_argument('--no-cpu-baseline', action='store_true', help=argparse.SUPPRESS)
    ap.add_argument('--worker', action='store_true', help=argparse.SUPPRESS)
    args = ap.parse_args()

    # Single-GPU runs are measured in a child process: if the HIP runtime aborts while creating the queues / scratch of
    # the requested number of launches in flight (it does not return an error, it kills the process), the measurement
    # is repeated with half as many instead of producing no line at all.  (Ranks started by torch.distributed.run are
    # the workers themselves.)
    if not args.worker and int(os.environ.get('WORLD_SIZE', '1')) == 1:
        import subprocess
        depth = max(1, args.pipeline)
        while True:
            env = dict(os.environ)
            if int(env.get('GPU_MAX_HW_QUEUES', '4')) > depth or depth < DEFAULT_IN_FLIGHT:
                env['GPU_MAX_HW_QUEUES'] = str(max(4, depth))
            cmd = [sys.executable, os.path.abspath(__file__), '--worker', '--gpus', str(args.gpus), '--steps', str(args.steps),
                   '--warmup', str(args.warmup), '--batch', str(args.batch), '--pipeline', str(depth)]
            if args.no_cpu_baseline:
                cmd.append('--no-cpu-baseline')
            try:
                r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=1500)
            except subprocess.TimeoutExpired as exc:                       # a hung worker: keep what it had already printed
                so = exc.stdout.decode() if isinstance(exc.stdout, bytes) else (exc.stdout or '')
                se = exc.stderr.decode() if isinstance(exc.stderr, bytes) else (exc.stderr or '')
                r = subprocess.CompletedProcess(cmd, -9, so, 'timed out after 1500 s; ' + se)
            lines = [ln for ln in r.stdout.splitlines() if ln.startswith('{')]
            if lines:                                   # a complete measurement line exists (a later side metric may have failed)
                print(lines[-1], flush=True)
                if r.returncode != 0:
                    sys.stderr.write('bench.py: worker exited with %d after the measurement: %s\n' % (r.returncode, r.stderr[-400:]))
                return
            sys.stderr.write('bench.py: worker with %d steps in flight failed (exit %d): %s\n' % (depth, r.returncode, r.stderr[-400:]))
            if depth == 1:
                raise SystemExit(1)
            depth = max(1, depth // 2)

    import torch
    import torch.distributed as dist
    import chd_amd  # noqa: F401
    from chd_amd.phys_optim import PhysOptim, default_config
    from chd_amd.synth import make_batch

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs a GPU: the physics stage has no CPU path')
    torch.cuda.set_device(local)
    if world > 1:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        dist.init_process_group('nccl', device_id=torch.device('cuda', local))

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    B = args.batch
    seqs = make_batch(B, F=FRAMES, seed0=rank * B)
    # A step is one full staged solve of one batch.  With --pipeline D > 1, D steps are in flight at once: each has its
    # own solver handle (own HIP stream) and its own device-resident copy of the batch, and is driven by its own host
    # thread (the C call releases the GIL).  128 workgroups occupy half of the 256 CUs and a launch lasts as long as its
    # slowest sequence (several times the mean), so many launches must be in flight to keep every CU busy.
    depth = max(1, min(args.pipeline, max(1, args.steps)))
    solvers = [PhysOptim(device=local, config=default_config()) for _ in range(depth)]      # reference iteration caps and tol
    batches = [sv.upload(seqs) for sv in solvers]                                            # inputs + tables -> HBM (not timed)
    batch, solver = batches[0], solvers[0]

    from concurrent.futures import ThreadPoolExecutor
    pool = ThreadPoolExecutor(max_workers=depth)

    def run_steps(n):
        out = []
        if depth == 1:
            for _ in range(n):
                out.append(batches[0].solve())
            return out
        # slot i handles steps i, i + depth, ...: the slots run concurrently
        def slot(i):
            return [batches[i].solve() for _ in range(i, n, depth)]
        for part in pool.map(slot, range(depth)):
            out.extend(part)
        return out

    run_steps(args.warmup)
    barrier()
    t0 = time.perf_counter()
    stats = run_steps(args.steps)                                        # returns when every step is solved
    barrier()
    elapsed = time.perf_counter() - t0
    kernel_ms = 0.0; alg_bytes = 0.0; iters = 0; nfact = 0; nfall = 0; launches = 0; max_seq_ms = 0.0
    for st in stats:
        kernel_ms += st['kernel_ms'][0] + st['kernel_ms'][1]
        launches += 1 + (1 if st['kernel_ms'][1] > 0 else 0)
        alg_bytes += st['alg_bytes']; iters += st['total_iters']; nfact += st['total_factorizations']; nfall += st['n_fallback']
        max_seq_ms = max(max_seq_ms, st['max_seq_ms'])
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device='cuda')
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        agg = torch.tensor([float(iters)], dtype=torch.float64, device='cuda')
        dist.all_reduce(agg, op=dist.ReduceOp.SUM)
        tot_iters_all = float(agg.item())
    else:
        tot_iters_all = float(iters)

    res = batch.fetch()
    n_ok = sum(1 for r in res if r.dynamics_succeed and r.durations_succeed)
    sizes = res[0].sizes

    if rank == 0:
        total_seqs = world * B * args.steps
        # algorithmic bytes of rank 0's launches / average launch duration (HIP events); with overlapping launches the
        # per-launch rate is what the roofline of the kernel is compared with
        ach = alg_bytes / (kernel_ms * 1e-3) / 1e9 if kernel_ms > 0 else 0.0
        # fp64 work of the factorisations / substitutions / products (band N_b, half-width w, border b; multiply-add = 2),
        # from the mean problem size of the batch -- a secondary view: the trailing update of the factorisation runs on
        # the fp64 matrix cores
        Nb_ = sum(r.sizes['kkt_dim'] - r.sizes['border'] for r in res) / len(res)
        w_ = sum(r.sizes['halfband'] for r in res) / len(res)
        b_ = sum(r.sizes['border'] for r in res) / len(res)
        fl_fact = Nb_ * w_ * w_ + 2 * Nb_ * w_ * b_ + Nb_ * b_ * b_ + b_ ** 3 / 3
        fl_solve = 4 * (Nb_ * w_ + Nb_ * b_) + 2 * b_ * b_
        fl_mv = 2 * (Nb_ * (2 * w_ + 1) + 2 * Nb_ * b_ + b_ * b_)
        flops = nfact * (fl_fact + 2 * fl_solve + fl_mv) + iters * fl_mv
        traffic = None
        tfile = os.path.join(ROOT, 'profiles', 'traffic.json')
        if os.path.exists(tfile):
            try:
                traffic = json.load(open(tfile)).get('hbm_bytes_per_launch')
            except Exception:
                traffic = None
        out = {
            'metric': 'physics-optimized sequences/sec (90-frame)', 'value': total_seqs / elapsed, 'unit': 'sequences/s',
            'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': 1e3 * elapsed / args.steps,
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f64', 'data': 'synthetic',
            'config': {'workload': 'batch of %d synthetic Mixamo-like %d-frame walks per GPU (BASELINE configs[1]), staged NLP solve, '
                                   'reference iteration caps 7000/7000/7000/2500/2000/7000, tol 1e-3' % (B, FRAMES),
                       'sequences_per_gpu': B, 'frames': FRAMES, 'parallelism': 'independent sequences, 1 workgroup each; %d process(es); %d step(s) in flight per GPU on separate HIP streams' % (world, depth),
                       'kkt_dim': sizes['kkt_dim'], 'halfband': sizes['halfband'], 'border': sizes['border'], 'nnz_jac': sizes['nnz_jac'],
                       'ipm_iterations_per_sequence': tot_iters_all / (world * B * args.steps),
                       'factorizations_rank0': nfact, 'stage4_fallbacks_rank0': nfall, 'converged_rank0': '%d/%d' % (n_ok, B),
                       'slowest_sequence_ms': max_seq_ms},
            'roofline': {'bound': 'hbm', 'achieved': ach, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': ach / HBM_PEAK_GBS,
                         'traffic': traffic, 'kernel': 'chd_solve_kernel', 'launches': launches,
                         'avg_launch_ms': kernel_ms / max(1, launches),
                         'algorithmic_bytes_per_launch': alg_bytes / max(1, launches),
                         'launches_in_flight': depth,
                         'achieved_all_launches': alg_bytes / elapsed / 1e9,        # rank 0: bytes of every launch / wall time of the timed region
                         'fp64': {'achieved': flops / elapsed / 1e12, 'peak': 78.6, 'unit': 'TFLOP/s', 'frac': flops / elapsed / 1e12 / 78.6,
                                  'note': 'rank 0; band LDL^T + substitutions + products; MI355X fp64 vector = matrix peak 78.6 TFLOP/s (AMD spec; not in MI355X_MICROARCH.md)'}},
        }
        if world == 1 and not args.no_cpu_baseline:
            out['cpu_baseline'] = cpu_baseline()
        if world == 1 and not args.no_cpu_baseline:
            # the side metric runs after the physics line is safely out: the wrapper keeps the LAST line, so a failure
            # here (even one that kills the process) cannot cost the physics measurement
            print(json.dumps(out), flush=True)
            try:
                out['contact_net'] = contact_net_rate(torch.device('cuda', local))
            except Exception as exc:
                out['contact_net'] = {'error': '%s: %s' % (type(exc).__name__, exc)}
        print(json.dumps(out), flush=True)
    for bt in batches:
        bt.free()
    for sv in solvers:
        sv.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()

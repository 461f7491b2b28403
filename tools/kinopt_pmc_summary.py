#!/usr/bin/env python
"""Summarise the rocprofv3 PMC passes of tools/gpu_kinopt_pmc.sh (kinematic-optimisation kernel, 256 clips x 100 frames) into profiles/<tag>/kinopt_pmc.md and
refresh profiles/kinopt_traffic.json -- what bench.py reports as `kinematic_optimisation.roofline.traffic` (tagged with the hash of BOTH kernel files and the LDS tile).

    python tools/kinopt_pmc_summary.py gpurun_out/<tag> profiles/<tag>
"""
import collections
import csv
import glob
import json
import os
import sys

src, dst = sys.argv[1], sys.argv[2]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
import bench  # noqa: E402

os.makedirs(dst, exist_ok=True)
tot = collections.defaultdict(lambda: collections.defaultdict(float))
for f in glob.glob(os.path.join(src, 'pmc_*', '*counter_collection.csv')):
    for r in csv.DictReader(open(f)):
        k = 'kin' if 'chd_kin_solve' in r['Kernel_Name'] else ('ik' if 'chd_ik_step' in r['Kernel_Name'] else 'other')
        tot[k][r['Counter_Name']] += float(r['Counter_Value'])
k = tot['kin']
b = json.load(open(os.path.join(src, 'fetch.json')))
raw = (k['FETCH_SIZE'] + k['WRITE_SIZE']) * 1024.0
corr = (2 * k['FETCH_SIZE'] + k['WRITE_SIZE']) * 1024.0
alg = 8.0 * (2 * (507 * 100 - 423) + 8 * 87 * 100 + 2 * 420 * 100) * b['lsmr_iterations_per_clip'] * 256
lines = ['# rocprofv3 PMC passes on chd_kin_solve_kernel (256 clips x 100 frames: clusters of 8 workgroups, 13-frame slices in LDS, __launch_bounds__(512, 2))', '',
         'Both least-squares launches of `KinematicOptimizer.optimize` (tests/tools/kinopt_bench.py 256 100 0), one counter group per run.', '',
         '| counter | value |', '|---|---|'] + ['| %s | %.6g |' % (c, v) for c, v in sorted(k.items())]
lines += ['', 'FETCH_SIZE %.0f GB + WRITE_SIZE %.0f GB = %.0f GB raw per batch = %.2f x the algorithmic bytes (%.0f GB: 8 (2 m + 8 n + 2 x 420 F) per LSMR iteration x %.0f iterations / clip x 256); '
          'with the gfx950 FETCH x 2 correction %.0f GB = %.2f x.' % (k['FETCH_SIZE'] * 1024 / 1e9, k['WRITE_SIZE'] * 1024 / 1e9, raw / 1e9, raw / alg, alg / 1e9, b['lsmr_iterations_per_clip'], corr / 1e9, corr / alg)]
if 'SQ_WAVE_CYCLES' in k:
    lines += ['', 'Wave states: %.0f %% SQ_WAIT_ANY, %.0f %% SQ_ACTIVE_INST_ANY of SQ_WAVE_CYCLES.' % (100 * k['SQ_WAIT_ANY'] / k['SQ_WAVE_CYCLES'], 100 * k['SQ_ACTIVE_INST_ANY'] / k['SQ_WAVE_CYCLES'])]
lines += ['', 'kernel times of the same run: %s ms; %.1f clips/s; algorithmic %.0f GB/s' % (b['lsq_kernel_ms'], b['clips_per_s'], b['algorithmic_GBps'])]
open(os.path.join(dst, 'kinopt_pmc.md'), 'w').write('\n'.join(lines) + '\n')
json.dump({'clips': 256, 'frames': 100, 'fetch_kb': k['FETCH_SIZE'], 'write_kb': k['WRITE_SIZE'], 'hbm_bytes_per_batch_raw': raw, 'hbm_bytes_per_batch': corr,
           'tag': os.path.basename(dst.rstrip('/')), 'sources_sha256': bench._sources_sha256(('chd_kinopt.hip', 'chd_kinopt_kernels.hpp', 'chd_kinopt_host.hpp')),
           'note': 'rocprofv3 PMC passes of %s/kinopt_pmc.md on the current launch (clusters of 8 workgroups, 13-frame slices in LDS); hbm_bytes_per_batch doubles FETCH_SIZE per MI355X_MICROARCH.md' % dst},
          open(os.path.join(root, 'profiles', 'kinopt_traffic.json'), 'w'), indent=1)
print('\n'.join(lines))

#!/usr/bin/env python
"""Summarise the rocprofv3 CSV outputs of tools/gpu_round_start.sh into a profiles/ directory.

    python tools/pmc_summary.py gpurun_out/start profiles/<tag>

Writes <tag>/pmc.md (per-counter totals of the solver kernel's dispatches: HBM-side bytes, fp64 MFMA, wave states) and
refreshes profiles/traffic.json (HBM-side bytes per bench step, tagged with the commit) -- what bench.py reports as
`roofline.traffic`."""
import collections
import csv
import glob
import json
import os
import shutil
import subprocess
import sys

src, dst = sys.argv[1], sys.argv[2]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
os.makedirs(dst, exist_ok=True)
commit = subprocess.run(['git', 'log', '-1', '--format=%h'], cwd=root, stdout=subprocess.PIPE, text=True).stdout.strip()
open(os.path.join(dst, 'COMMIT'), 'w').write(commit + '\n')
for f in ('pytest_gpu.log', 'smoke.log', 'bench_driver.json', 'trace_bench.json', 'fetch_bench.json', 'write_bench.json', 'mfma_bench.json', 'sq_bench.json', 'fetch_ll_bench.json', 'write_ll_bench.json'):
    if os.path.exists(os.path.join(src, f)):
        shutil.copy(os.path.join(src, f), dst)
ks = os.path.join(src, 'trace', 'trace_kernel_stats.csv')
if os.path.exists(ks):
    shutil.copy(ks, dst)
out = ['# rocprofv3 PMC passes on the solver kernel (commit %s)' % commit, '', '| pass | counter | value summed over the device, main launch | fallback launch |', '|---|---|---|---|']
tot = {}
traffic_json = None
for sub in ('pmc_fetch', 'pmc_write', 'pmc_mfma', 'pmc_sq', 'pmc_fetch_ll', 'pmc_write_ll'):
    for f in glob.glob(os.path.join(src, sub, '*counter_collection.csv')):
        agg = collections.defaultdict(lambda: collections.defaultdict(float))
        for r in csv.DictReader(open(f)):
            if r['Kernel_Name'].startswith('chd_solve_kernel'):
                agg[r['Counter_Name']][r['Dispatch_Id']] += float(r['Counter_Value'])
        for c in sorted(agg):
            d = agg[c]; k = sorted(d, key=lambda x: -d[x])
            out.append('| %s | %s | %.6g | %.6g |' % (sub, c, d[k[0]], d[k[1]] if len(k) > 1 else 0.0))
            tot[c + ('_LL' if sub.endswith('_ll') else '')] = sum(d.values())
if 'FETCH_SIZE' in tot and 'WRITE_SIZE' in tot:
    b = json.loads(open(os.path.join(src, 'fetch_bench.json')).read().strip().splitlines()[-1])
    alg = b['roofline']['algorithmic_bytes_per_step']
    raw = (tot['FETCH_SIZE'] + tot['WRITE_SIZE']) * 1024.0; corr = (2 * tot['FETCH_SIZE'] + tot['WRITE_SIZE']) * 1024.0
    out += ['', 'One step (128 sequences): FETCH_SIZE %.1f GB + WRITE_SIZE %.1f GB = %.1f GB raw = %.1f x the algorithmic bytes (%.2f GB); with the gfx950 FETCH x 2 correction %.1f GB = %.1f x.'
            % (tot['FETCH_SIZE'] * 1024 / 1e9, tot['WRITE_SIZE'] * 1024 / 1e9, raw / 1e9, raw / alg, alg / 1e9, corr / 1e9, corr / alg)]
    traffic_json = dict({'hbm_bytes_per_step': corr, 'hbm_bytes_per_step_raw': raw, 'fetch_kb': tot['FETCH_SIZE'], 'write_kb': tot['WRITE_SIZE'], 'algorithmic_bytes_per_step': alg,
               'tag': os.path.basename(dst.rstrip('/')), 'commit': commit, 'sources_sha256': __import__('bench').kernel_sources_sha256(),
               'note': 'rocprofv3 PMC passes of %s (one step = 128 sequences, seeds 0..127); FETCH_SIZE doubled per MI355X_MICROARCH.md HBM section; measured on commit %s' % (dst, commit)},
              )
if 'FETCH_SIZE_LL' in tot and 'WRITE_SIZE_LL' in tot:
    b = json.loads(open(os.path.join(src, 'fetch_ll_bench.json')).read().strip().splitlines()[-1])
    alg = b['roofline']['algorithmic_bytes_per_step']
    raw = (tot['FETCH_SIZE_LL'] + tot['WRITE_SIZE_LL']) * 1024.0; corr = (2 * tot['FETCH_SIZE_LL'] + tot['WRITE_SIZE_LL']) * 1024.0
    out += ['', 'Left-looking factorisation (chd_config.factorisation = 1), one step: FETCH_SIZE %.1f GB + WRITE_SIZE %.1f GB = %.1f GB raw = %.1f x the algorithmic bytes (%.2f GB); with the FETCH x 2 correction %.1f GB = %.1f x.'
            % (tot['FETCH_SIZE_LL'] * 1024 / 1e9, tot['WRITE_SIZE_LL'] * 1024 / 1e9, raw / 1e9, raw / alg, alg / 1e9, corr / 1e9, corr / alg)]
if 'SQ_INSTS_VALU_MFMA_MOPS_F64' in tot:
    b = json.loads(open(os.path.join(src, 'mfma_bench.json')).read().strip().splitlines()[-1])
    ms = b['roofline']['kernel_ms_rank0']
    fl = tot['SQ_INSTS_VALU_MFMA_MOPS_F64'] * 512
    out += ['', 'fp64 MFMA: %.3g v_mfma_f64_16x16x4_f64 (%.3g flop) in %.0f ms of kernel time = %.2f TFLOP/s = %.2f %% of the 78.6 TFLOP/s matrix peak.'
            % (tot['SQ_INSTS_VALU_MFMA_MOPS_F64'] / 4, fl, ms, fl / (ms * 1e-3) / 1e12, 100 * fl / (ms * 1e-3) / 78.6e12)]
    if traffic_json is not None:
        traffic_json['mfma_tflops'] = fl / (ms * 1e-3) / 1e12
if 'SQ_WAVE_CYCLES' in tot and 'SQ_WAIT_ANY' in tot:
    out += ['', 'Wave states: %.0f %% SQ_WAIT_ANY, %.0f %% SQ_ACTIVE_INST_ANY, %.0f %% SQ_WAIT_INST_ANY of SQ_WAVE_CYCLES.'
            % (100 * tot['SQ_WAIT_ANY'] / tot['SQ_WAVE_CYCLES'], 100 * tot['SQ_ACTIVE_INST_ANY'] / tot['SQ_WAVE_CYCLES'], 100 * tot['SQ_WAIT_INST_ANY'] / tot['SQ_WAVE_CYCLES'])]
    if traffic_json is not None:
        traffic_json['sq_wait_any_fraction'] = tot['SQ_WAIT_ANY'] / tot['SQ_WAVE_CYCLES']
if traffic_json is not None:
    json.dump(traffic_json, open(os.path.join(root, 'profiles', 'traffic.json'), 'w'), indent=1)
open(os.path.join(dst, 'pmc.md'), 'w').write('\n'.join(out) + '\n')
print('\n'.join(out))

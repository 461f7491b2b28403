"""Summarise rocprofv3 (rocpd SQLite) outputs of `tools/gpu_profile.sh` into profiles/.

  python tools/rocprof_summary.py gpurun_out/prof r01

writes profiles/<tag>_kernel_stats.md (per-kernel totals, the `--kernel-trace --stats` view),
profiles/<tag>_pmc_hbm.md and profiles/traffic.json (HBM bytes per launch of the dominant kernel from the
FETCH_SIZE / WRITE_SIZE passes; on gfx950 FETCH_SIZE counts 64 B per 128-B request of a wide streaming read, so
the read side is doubled as MI355X_MICROARCH.md prescribes — an upper bound for this kernel's mostly 8-byte-per-lane
accesses; both raw and corrected figures are recorded)."""
import json
import os
import sqlite3
import sys

src, tag = sys.argv[1], sys.argv[2]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = os.path.join(root, 'profiles')
os.makedirs(out, exist_ok=True)


def find_db(sub):
    d = os.path.join(src, sub)
    for f in os.listdir(d):
        if f.endswith('.db'):
            return os.path.join(d, f)
    raise SystemExit('no .db in ' + d)


con = sqlite3.connect(find_db('trace'))
rows = list(con.execute('select name, total_calls, total_duration, average, percentage from top_kernels'))
disp = list(con.execute("select name, duration, grid_x, workgroup_x, lds_size, scratch_size, vgpr_count, accum_vgpr_count, sgpr_count from kernels where name like 'chd_solve_kernel%' order by start"))
with open(os.path.join(out, tag + '_kernel_stats.md'), 'w') as f:
    f.write('# rocprofv3 --kernel-trace --stats  (python bench.py --no-cpu-baseline: default steps / warm-up / launches in flight)\n\n')
    f.write('| kernel | calls | total (ms) | average (ms) | % |\n|---|---|---|---|---|\n')
    for n, c, t, a, p in rows:
        f.write('| `%s` | %d | %.3f | %.3f | %.3f |\n' % (n, c, t / 1e3, a / 1e3, p))      # top_kernels is in microseconds
    f.write('\n## dispatches of the solver kernel\n\n| # | duration (ms) | grid (threads) | workgroup | LDS (B) | scratch (B/lane) | VGPR | AGPR | SGPR |\n|---|---|---|---|---|---|---|---|---|\n')
    for i, r in enumerate(disp):
        f.write('| %d | %.3f | %d | %d | %d | %d | %d | %d | %d |\n' % (i, r[1] / 1e6, r[2], r[3], r[4], r[5], r[6], r[7], r[8]))
    big = [r[1] for r in disp if r[2] >= 128 * 64]
    if big:
        f.write('\nAverage duration of the full-batch launches (grid = 128 workgroups): %.3f ms over %d launches.\n' % (sum(big) / len(big) / 1e6, len(big)))

res = {}
for sub, ctr in (('pmc_fetch', 'FETCH_SIZE'), ('pmc_write', 'WRITE_SIZE')):
    c2 = sqlite3.connect(find_db(sub))
    r = list(c2.execute("select value, duration, grid_size from counters_collection where counter_name=? and kernel_name like 'chd_solve_kernel%' order by start", (ctr,)))
    res[ctr] = r
with open(os.path.join(out, tag + '_pmc_hbm.md'), 'w') as f:
    f.write('# rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes; python bench.py --steps 1 --warmup 0 --no-cpu-baseline)\n\n')
    f.write('| counter | launch | value (KB) | duration (ms) | grid |\n|---|---|---|---|---|\n')
    for ctr, r in res.items():
        for i, (v, d, g) in enumerate(r):
            f.write('| %s | %d | %.1f | %.3f | %d |\n' % (ctr, i, v, d / 1e6, g))
    fk = res['FETCH_SIZE'][0][0] if res['FETCH_SIZE'] else 0.0
    wk = res['WRITE_SIZE'][0][0] if res['WRITE_SIZE'] else 0.0
    raw = (fk + wk) * 1024.0
    corr = (2.0 * fk + wk) * 1024.0
    dur = res['FETCH_SIZE'][0][1] / 1e9 if res['FETCH_SIZE'] else 0.0
    f.write('\nFull-batch launch (128 sequences, stages 1.1-3): FETCH_SIZE %.3f GB, WRITE_SIZE %.3f GB.\n' % (fk * 1024 / 1e9, wk * 1024 / 1e9))
    f.write('HBM bytes per launch: raw (FETCH+WRITE) = %.3f GB; with the gfx950 FETCH_SIZE x2 correction = %.3f GB ' % (raw / 1e9, corr / 1e9))
    f.write('(%.1f GB/s over the %.3f s launch).\n' % (corr / 1e9 / dur if dur else 0.0, dur))
json.dump({'hbm_bytes_per_launch': corr, 'hbm_bytes_per_launch_raw': raw, 'fetch_kb': fk, 'write_kb': wk, 'tag': tag,
           'note': 'first (full-batch) chd_solve_kernel launch; FETCH_SIZE doubled per MI355X_MICROARCH.md HBM section'},
          open(os.path.join(out, 'traffic.json'), 'w'), indent=1)
print(open(os.path.join(out, tag + '_kernel_stats.md')).read())
print(open(os.path.join(out, tag + '_pmc_hbm.md')).read())

#!/bin/bash
# PMC passes on the kinematic-optimisation kernel (256 clips x 100 frames): HBM-side bytes, wave states
set -u
tag=${1:-kinopt_pmc}
R=${GRAFT_REPO_ROOT:-$(pwd)}
out=$R/gpurun_out/$tag
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
P="python $R/tests/tools/kinopt_bench.py 256 100 0"
timeout 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/pmc_fetch -o fetch -- $P > $out/fetch.json 2> $out/fetch.err
timeout 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/pmc_write -o write -- $P > $out/write.json 2> $out/write.err
timeout 200 rocprofv3 --pmc SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM --output-format csv -d $out/pmc_sq -o sq -- $P > $out/sq.json 2> $out/sq.err
python - <<P
import csv, glob, collections, json
out = '$out'
tot = collections.defaultdict(lambda: collections.defaultdict(float))
for f in glob.glob(out + '/pmc_*/*counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        k = 'kin' if 'chd_kin_solve' in r['Kernel_Name'] else ('ik' if 'chd_ik_step' in r['Kernel_Name'] else 'other')
        tot[k][r['Counter_Name']] += float(r['Counter_Value'])
for k, v in tot.items():
    print(k, dict(v))
for n in ('fetch', 'write', 'sq'):
    try:
        d = json.load(open(out + '/%s.json' % n)); print(n, 'lsq ms', d['lsq_kernel_ms'], 'its/clip', d['lsmr_iterations_per_clip'], 'alg GB/s', d['algorithmic_GBps'])
    except Exception as e:
        print(n, 'bench output unreadable', e)
P

#!/bin/bash
# end-of-round validation of HEAD: GPU tests, smoke, the driver's bench line, the next rows' benches
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; O=gpurun_out/final3; mkdir -p $O
timeout 600 python -m pytest tests -q -m gpu -x > $O/pytest_gpu.log 2>&1; tail -2 $O/pytest_gpu.log
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_head.json 2> $O/bench.err
timeout 300 python tests/tools/kinopt_bench.py 256 100 12 > $O/kinopt_bench_256x100.json 2>/dev/null
timeout 500 python tests/tools/kinopt_bench.py 1024 100 0 > $O/kinopt_bench_1024x100.json 2>/dev/null
timeout 200 python tests/tools/ik_bench.py 128 90 > $O/ik_bench.log 2>&1
python - <<P
import json
d=json.load(open('$O/bench_driver_head.json')); print('bench', d['value'], d['parity']['worst_rel_l2'], d['parity']['sequences_above_1e-3'], d['kinematic_optimisation']['clips_per_s'])
for n in ('256','1024'):
    k=json.load(open('$O/kinopt_bench_%sx100.json'%n)); print('kinopt',n,k['clips_per_s'],k['ik_kernel_ms'],k['lsq_kernel_ms'][:2])
P
tail -2 $O/ik_bench.log

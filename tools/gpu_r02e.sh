#!/bin/bash
# Round 2, GPU call 5: config-4 end-to-end test; trailing-update tile-count variants.
set -x
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r02e
mkdir -p $OUT
cd $R
timeout 300 python -m pytest tests -x -q -m gpu > $OUT/pytest_gpu.log 2>&1; tail -5 $OUT/pytest_gpu.log
for v in tp6 tp8; do
  CHD_PHYS_LIB=$R/contact-human-dynamics_amd/csrc/variants/libchd_$v.so timeout 300 python bench.py --no-cpu-baseline --no-side-metrics > $OUT/bench_$v.json 2> $OUT/bench_$v.err; tail -c 700 $OUT/bench_$v.json; tail -2 $OUT/bench_$v.err
done
timeout 300 python bench.py --no-cpu-baseline > $OUT/bench_base.json 2> $OUT/bench_base.err; tail -c 700 $OUT/bench_base.json

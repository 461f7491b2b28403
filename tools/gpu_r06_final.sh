#!/bin/bash
# round 6, closing GPU call: GPU tests + smoke + the driver's bench line, and the FETCH_SIZE pass the profile call lost to its 150 s limit (tag = $1, default r06q)
set -x
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/${1:-r06q}
mkdir -p $OUT
cd $R
timeout 1500 python -m pytest tests -x -q -m gpu > $OUT/pytest_gpu.log 2>&1; tail -5 $OUT/pytest_gpu.log
timeout 200 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $OUT/smoke.log 2>&1; tail -2 $OUT/smoke.log
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_driver.json 2> $OUT/bench_driver.err; tail -c 700 $OUT/bench_driver.json; tail -3 $OUT/bench_driver.err
cd /tmp && export TMPDIR=/tmp
P="python $R/bench.py --gen-workers 8 --no-cpu-baseline --no-side-metrics --strong-total 0"
timeout 400 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o fetch -- $P --steps 1 --warmup 0 > $OUT/fetch_bench.json 2> $OUT/fetch.err; tail -2 $OUT/fetch.err
find $OUT -name "*counter_collection.csv" | head

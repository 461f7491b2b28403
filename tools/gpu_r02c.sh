#!/bin/bash
# Round 2, GPU call 3: persistent kernel with the stall guard in the bench configuration, long sequence, profiles.
set -x
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r02c
mkdir -p $OUT
cd $R
timeout 300 python -m pytest tests -x -q -m gpu > $OUT/pytest_gpu.log 2>&1; tail -3 $OUT/pytest_gpu.log
timeout 600 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; tail -c 1500 $OUT/bench_default.json; tail -3 $OUT/bench_default.err
timeout 300 python bench.py --steps 20 --no-cpu-baseline --no-side-metrics > $OUT/bench_k20.json 2> $OUT/bench_k20.err; tail -c 600 $OUT/bench_k20.json
timeout 300 python tests/tools/gpu_long.py 600 10 > $OUT/long600.log 2>&1; tail -6 $OUT/long600.log
cd /tmp && export TMPDIR=/tmp
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o trace -- python $R/bench.py --steps 4 --gen-workers 1 --no-cpu-baseline --no-side-metrics > $OUT/trace_bench.json 2> $OUT/trace.err
head -4 $OUT/trace/trace_kernel_stats.csv
timeout 150 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o fetch -- python $R/bench.py --steps 1 --warmup 0 --gen-workers 1 --no-cpu-baseline --no-side-metrics > $OUT/fetch_bench.json 2> $OUT/fetch.err; tail -2 $OUT/fetch.err
timeout 150 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o write -- python $R/bench.py --steps 1 --warmup 0 --gen-workers 1 --no-cpu-baseline --no-side-metrics > $OUT/write_bench.json 2> $OUT/write.err; tail -2 $OUT/write.err
find $OUT -name "*counter_collection.csv" | head

#!/bin/bash
# Runs on the GPU box (via gpurun): bench + rocprofv3 kernel trace + two PMC passes (HBM read / write bytes).
set -x
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof
mkdir -p $OUT
cd $R
timeout 600 python bench.py > $OUT/bench.json 2> $OUT/bench.err
tail -c 2500 $OUT/bench.json
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/trace -o trace -- python $R/bench.py --worker --no-cpu-baseline > $OUT/trace_bench.json 2> $OUT/trace.err
timeout 300 rocprofv3 --pmc FETCH_SIZE -d $OUT/pmc_fetch -o fetch -- python $R/bench.py --worker --steps 1 --warmup 0 --no-cpu-baseline > $OUT/fetch_bench.json 2> $OUT/fetch.err
timeout 300 rocprofv3 --pmc WRITE_SIZE -d $OUT/pmc_write -o write -- python $R/bench.py --worker --steps 1 --warmup 0 --no-cpu-baseline > $OUT/write_bench.json 2> $OUT/write.err
find $OUT -type f | head -40
find $OUT -name "*stats*" | head; for f in $(find $OUT/trace -name "*kernel_stats*"); do head -5 $f; done
for f in $(find $OUT/pmc_fetch $OUT/pmc_write -name "*counter_collection*"); do echo $f; head -4 $f; done

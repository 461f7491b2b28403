#!/bin/bash
# Round 2, GPU call 4: row_nodes as a by-value function (code size 146 k -> 103 k instructions); instruction-cache counters.
set -x
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r02d
mkdir -p $OUT
cd $R
timeout 300 python -m pytest tests -x -q -m gpu > $OUT/pytest_gpu.log 2>&1; tail -3 $OUT/pytest_gpu.log
timeout 400 python bench.py --no-cpu-baseline --no-side-metrics > $OUT/bench_k12.json 2> $OUT/bench_k12.err; tail -c 900 $OUT/bench_k12.json; tail -2 $OUT/bench_k12.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L 2>/dev/null | grep -i -E "ICACHE|IFETCH|SQ_INSTS_VALU |SQ_WAIT_INST_ANY|SQ_ACTIVE_INST_ANY|SQ_INST_LEVEL" | sort -u | head -30 > $OUT/counters_icache.txt; cat $OUT/counters_icache.txt
timeout 150 rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d $OUT/pmc_icache -o ic -- python $R/bench.py --steps 2 --warmup 0 --gen-workers 1 --no-cpu-baseline --no-side-metrics > $OUT/ic_bench.json 2> $OUT/ic.err; tail -2 $OUT/ic.err
timeout 150 rocprofv3 --pmc SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM --output-format csv -d $OUT/pmc_sq -o sq -- python $R/bench.py --steps 2 --warmup 0 --gen-workers 1 --no-cpu-baseline --no-side-metrics > $OUT/sq_bench.json 2> $OUT/sq.err; tail -2 $OUT/sq.err
find $OUT -name "*counter_collection.csv" | head

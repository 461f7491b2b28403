#!/bin/bash
# round 6, closing call: GPU tests + smoke + the driver's bench line + kernel trace at the driver's configuration + the four PMC passes (tag = $1)
set -x
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/${1:-r06z}
mkdir -p $OUT
cd $R
timeout 1500 python -m pytest tests -x -q -m gpu > $OUT/pytest_gpu.log 2>&1; tail -5 $OUT/pytest_gpu.log
timeout 200 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $OUT/smoke.log 2>&1; tail -2 $OUT/smoke.log
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_driver.json 2> $OUT/bench_driver.err; tail -c 500 $OUT/bench_driver.json; tail -3 $OUT/bench_driver.err
cd /tmp && export TMPDIR=/tmp
P="python $R/bench.py --gen-workers 8 --no-cpu-baseline --no-side-metrics --strong-total 0"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o trace -- $P --steps 20 --warmup 5 > $OUT/trace_bench.json 2> $OUT/trace.err; head -3 $OUT/trace/trace_kernel_stats.csv
sleep 5
# (the first counter pass after the trace run has twice been ended by a stray SIGTERM within its first second: it is repeated until its output exists)
for try in 1 2 3; do
  timeout 400 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o fetch -- $P --steps 1 --warmup 0 > $OUT/fetch_bench.json 2> $OUT/fetch.err
  if find $OUT/pmc_fetch -name "*counter_collection.csv" 2>/dev/null | grep -q .; then break; fi
  sleep 5
done
timeout 400 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o write -- $P --steps 1 --warmup 0 > $OUT/write_bench.json 2> $OUT/write.err
timeout 400 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_WAVE_CYCLES --output-format csv -d $OUT/pmc_mfma -o mfma -- $P --steps 2 --warmup 0 > $OUT/mfma_bench.json 2> $OUT/mfma.err
timeout 400 rocprofv3 --pmc SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM --output-format csv -d $OUT/pmc_sq -o sq -- $P --steps 2 --warmup 0 > $OUT/sq_bench.json 2> $OUT/sq.err
find $OUT -name "*counter_collection.csv" | head

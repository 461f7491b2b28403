#!/bin/bash
# Round 2, GPU call 2: first run of the persistent-queue kernel (LDS-resident descriptor/context, per-workgroup workspaces).
set -x
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r02b
mkdir -p $OUT
cd $R
timeout 900 python -m pytest tests -x -q -m gpu > $OUT/pytest_gpu.log 2>&1; tail -5 $OUT/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $OUT/smoke.log 2>&1; tail -2 $OUT/smoke.log
timeout 900 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; tail -c 3000 $OUT/bench_default.json; tail -3 $OUT/bench_default.err
# occupancy experiment: two 256-thread workgroups per compute unit with half the LDS each
timeout 600 python bench.py --steps 8 --threads 256 --lds-kb 76 --max-workgroups 512 --no-cpu-baseline --no-side-metrics > $OUT/bench_2wg.json 2> $OUT/bench_2wg.err; tail -c 1200 $OUT/bench_2wg.json; tail -3 $OUT/bench_2wg.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L 2>/dev/null | grep -i -E "mfma|FETCH_SIZE|WRITE_SIZE|SQ_BUSY_CYCLES|SQ_WAVE_CYCLES|SQ_INSTS_VALU\b|GRBM_GUI" | head -40 > $OUT/counters_available.txt; cat $OUT/counters_available.txt | head -30
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o trace -- python $R/bench.py --steps 4 --no-cpu-baseline --no-side-metrics > $OUT/trace_bench.json 2> $OUT/trace.err
for f in $(find $OUT/trace -name "*kernel_stats*"); do head -6 $f; done
timeout 400 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o fetch -- python $R/bench.py --steps 2 --warmup 0 --no-cpu-baseline --no-side-metrics > $OUT/fetch_bench.json 2> $OUT/fetch.err
timeout 400 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o write -- python $R/bench.py --steps 2 --warmup 0 --no-cpu-baseline --no-side-metrics > $OUT/write_bench.json 2> $OUT/write.err
timeout 400 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_WAVE_CYCLES --output-format csv -d $OUT/pmc_mfma -o mfma -- python $R/bench.py --steps 2 --warmup 0 --no-cpu-baseline --no-side-metrics > $OUT/mfma_bench.json 2> $OUT/mfma.err
tail -2 $OUT/mfma.err
find $OUT -name "*.csv" | head -20
for f in $(find $OUT/pmc_fetch $OUT/pmc_write $OUT/pmc_mfma -name "*counter_collection*.csv"); do echo $f; head -3 $f; done

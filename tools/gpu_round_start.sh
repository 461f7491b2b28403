#!/bin/bash
# First GPU call of a round (run through gpurun from the repo root; ~6-8 minutes of box time):
#   1. the hot path's GPU tests and smoke()            -> gpurun_out/start/pytest_gpu.log, smoke.log
#   2. the rows not yet run on a GPU (IK back-projection, in-memory pipeline, device-side contact pre-processing)
#                                                      -> pytest_gpu_next.log, ik_bench.log (+ kernel trace)
#   3. bench.py at its default depth, then with 16 launches in flight (only works while the kernel's scratch stays
#      <= 4288 B/lane; the wrapper falls back by itself if the runtime refuses)
# Every step runs under its own timeout so that a hang cannot eat the budget; nothing here reads /root/reference.
# Round 2: the inertia retry (CHD_INERTIA_RETRY) and the 1e-8 margin in the correction step's boundary test
# went in after round 1's last GPU run -- step 1 below is their first GPU validation (-DCHD_INERTIA_RETRY=0 + reverting the margin
# gives the measured round-1 kernel); add seeds 31, 73, 77, 105, 107, 113 (tilts as in profiles/r01_parity_cpu_emulation.md) to gpu_long.py.
set -x
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/start
mkdir -p $OUT
cd $R
timeout 900 python -m pytest tests -x -q -m gpu > $OUT/pytest_gpu.log 2>&1; tail -3 $OUT/pytest_gpu.log
timeout 600 python tests/tools/gpu_parity_holes.py > $OUT/parity_holes.log 2>&1; tail -12 $OUT/parity_holes.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $OUT/smoke.log 2>&1; tail -2 $OUT/smoke.log
timeout 300 python -m pytest tests -x -q -m gpu_next > $OUT/pytest_gpu_next.log 2>&1; tail -5 $OUT/pytest_gpu_next.log
timeout 300 python tests/tools/ik_bench.py 128 90 > $OUT/ik_bench.log 2>&1; tail -3 $OUT/ik_bench.log
timeout 600 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; tail -c 1500 $OUT/bench_default.json; tail -3 $OUT/bench_default.err
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/ik_trace -o ik -- python $R/tests/tools/ik_bench.py 128 90 > $OUT/ik_trace.log 2>&1
for f in $(find $OUT/ik_trace -name "*kernel_stats*"); do head -4 $f; done

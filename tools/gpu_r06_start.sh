#!/bin/bash
# round 6, first GPU call: GPU tests + smoke + the driver's bench line at HEAD (the round's baseline on this pool), and the physics inputs the repo's own
# upstream stages produce for 192 synthetic videos x 100 frames (three sets of 64: A, B and a third fresh one -- VERDICT r05 next-1 / next-8 ask for globalisation
# studies on fresh clips), kept as a tarball for the host emulation (tests/tools/emu_sweep.py --dirs).
set -x
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r06a
mkdir -p $OUT
cd $R
timeout 1200 python -m pytest tests -x -q -m gpu > $OUT/pytest_gpu.log 2>&1; tail -3 $OUT/pytest_gpu.log
timeout 200 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $OUT/smoke.log 2>&1; tail -2 $OUT/smoke.log
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_driver.json 2> $OUT/bench_driver.err; tail -c 600 $OUT/bench_driver.json; tail -3 $OUT/bench_driver.err
rm -rf /tmp/pipe192
timeout 600 python tests/tools/pipeline_bench.py --videos 192 --frames 100 --keep /tmp/pipe192 > $OUT/pipe192.json 2> $OUT/pipe192.err; tail -c 800 $OUT/pipe192.json; tail -3 $OUT/pipe192.err
timeout 300 python tests/tools/pipe_phys_stats.py /tmp/pipe192/data 100 $OUT/pipe_stats 0 > $OUT/pipe_stats.log 2>&1; tail -2 $OUT/pipe_stats.log
(cd /tmp/pipe192/data && tar czf $OUT/pipe192_inputs.tgz */phys_optim_in_combined)
ls -la $OUT

#!/bin/bash
# Depth sweep of bench.py on one MI355X (run through gpurun from the repo root; ~4 minutes): how many launches in flight
# pay off with the current kernel (scratch per lane decides how many hardware queues the runtime can create).
# Each run is a separate worker process under its own timeout; a depth the runtime refuses shows up as a non-zero exit.
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/sweep
mkdir -p $OUT
cd $R
for D in ${@:-4 8 12 16}; do
  GPU_MAX_HW_QUEUES=$D timeout 300 python bench.py --worker --pipeline $D --steps $((2 * D)) --warmup 2 --no-cpu-baseline > $OUT/depth_$D.json 2> $OUT/depth_$D.err
  echo "depth $D exit $? $(python -c "import json,sys; d=json.loads(open('$OUT/depth_$D.json').read().strip().splitlines()[-1]); print(d['value'], d['unit'], d['ms_per_step'], 'ms/step')" 2>/dev/null)"
done

#!/bin/bash
# quick bench (no side metrics) with the library named by $1 (optional), output name $2
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/quick
mkdir -p $OUT
cd $R
for spec in "$@"; do
  name=${spec%%:*}; lib=${spec#*:}
  if [ "$lib" != "$name" ] && [ -n "$lib" ]; then export CHD_PHYS_LIB=$R/$lib; else unset CHD_PHYS_LIB; fi
  timeout 300 python bench.py --no-cpu-baseline --no-side-metrics > $OUT/bench_$name.json 2> $OUT/bench_$name.err
  python - <<PY
import json
d=json.loads(open('$OUT/bench_$name.json').read().strip().splitlines()[-1]); c=d['config']
print('$name', 'value %.1f mean %.0f slowest %.0f' % (d['value'], c['mean_sequence_ms'], c['slowest_sequence_ms']), c['in_kernel_phase_ms_per_sequence'])
PY
done

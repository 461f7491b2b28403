#!/bin/bash
# round 6: how does the solver kernel's throughput depend on the number of resident workgroups?  (slowest_sequence_ms_128 = 605 ms against 1 553 ms at full load for the same
# 231-iteration sequence says a sequence runs 2.4 x faster when half the compute units are idle: contention for the memory system, not latency alone)
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r06b
mkdir -p $OUT
cd $R
for wg in 256 224 192 160 128 96 64 32; do
  timeout 300 python bench.py --no-cpu-baseline --no-side-metrics --strong-total 0 --steps 10 --warmup 2 --max-workgroups $wg > $OUT/bench_wg$wg.json 2> $OUT/bench_wg$wg.err
  python - <<PY
import json
d=json.loads(open('$OUT/bench_wg$wg.json').read().strip().splitlines()[-1]); c=d['config']
print('wg $wg', 'value %.1f mean_seq_ms %.0f slowest %.0f busy %.2f' % (d['value'], c['mean_sequence_ms'], c['slowest_sequence_ms'], d['roofline']['kernel_busy_fraction']), [round(x) for x in c['in_kernel_phase_ms_per_sequence'][:13]])
PY
done

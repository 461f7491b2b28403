#!/bin/bash
# round 5, kinematic optimisation on clusters of workgroups: GPU tests, batch sizes, PMC passes
set -u
tag=${1:-r05s}
R=${GRAFT_REPO_ROOT:-$(pwd)}
out=$R/gpurun_out/$tag
mkdir -p "$out"
cd "$R"
timeout 400 python -m pytest tests/test_kinopt_gpu.py tests/test_config4_gpu.py -q > $out/tests.log 2>&1; tail -4 $out/tests.log
for shape in "256 100" "128 100" "64 100" "32 60" "1024 100"; do
  set -- $shape
  timeout 200 python tests/tools/kinopt_bench.py $1 $2 0 > $out/kin_$1x$2.json 2> $out/kin_$1x$2.err
  python - <<P
import json
try:
    d = json.load(open('$out/kin_$1x$2.json')); print('$1x$2', 'clips/s %.1f' % d['clips_per_s'], 'lsq ms', [round(x) for x in d['lsq_kernel_ms']], 'its/clip %.0f' % d['lsmr_iterations_per_clip'])
except Exception as e:
    print('$1x$2 failed', e)
P
done
if [ "${2:-}" != "nopmc" ]; then bash tools/gpu_kinopt_pmc.sh $tag > $out/pmc.log 2>&1; tail -6 $out/pmc.log; fi
timeout 400 python tests/tools/pipeline_bench.py --videos 64 --frames 100 > $out/pipe100.json 2> $out/pipe100.err; tail -c 700 $out/pipe100.json

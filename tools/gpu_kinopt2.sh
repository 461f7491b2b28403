#!/bin/bash
# LDS-tile / batch-size sweep of the kinematic-optimisation kernel
set -u
tag=${1:-kinopt2}
out=gpurun_out/$tag
mkdir -p "$out"
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
for l in 4608 9216 18432; do
  KIN_LDS_DOUBLES=$l timeout 300 python tests/tools/kinopt_bench.py 256 100 0 > "$out/bench_256x100_lds$l.json" 2> "$out/err_lds$l.log"; echo "lds doubles $l"; tail -c 600 "$out/bench_256x100_lds$l.json"; tail -2 "$out/err_lds$l.log"
done
for l in 4608 9216; do
KIN_LDS_DOUBLES=$l timeout 400 python tests/tools/kinopt_bench.py 1024 100 0 > "$out/bench_1024x100_lds$l.json" 2> "$out/err1024_$l.log"; echo "B 1024 lds $l"; tail -c 600 "$out/bench_1024x100_lds$l.json"; tail -2 "$out/err1024_$l.log"
done

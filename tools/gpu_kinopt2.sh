#!/bin/bash
# thread-count / batch-size sweep of the kinematic-optimisation kernel
set -u
tag=${1:-kinopt2}
out=gpurun_out/$tag
mkdir -p "$out"
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
for t in 256 512 1024; do
  KIN_THREADS=$t timeout 300 python tests/tools/kinopt_bench.py 256 100 0 > "$out/bench_256x100_t$t.json" 2> "$out/err_t$t.log"; echo "threads $t"; tail -c 700 "$out/bench_256x100_t$t.json"; tail -2 "$out/err_t$t.log"
done
for t in 512 1024; do
  KIN_THREADS=$t timeout 400 python tests/tools/kinopt_bench.py 1024 100 0 > "$out/bench_1024x100_t$t.json" 2> "$out/err1024_t$t.log"; echo "B 1024 threads $t"; tail -c 700 "$out/bench_1024x100_t$t.json"; tail -2 "$out/err1024_t$t.log"
done

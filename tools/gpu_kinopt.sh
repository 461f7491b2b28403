#!/bin/bash
# GPU check of the kinematic-optimisation row: parity tests, a 256-clip x 100-frame batch, kernel trace.
# usage (from the repo root): gpurun --timeout 900 -- 'bash tools/gpu_kinopt.sh r02h_kinopt'
set -u
tag=${1:-kinopt}
out=gpurun_out/$tag
mkdir -p "$out"
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_kinopt_gpu.py tests/test_ik_gpu.py -q -m gpu -x -s > "$out/pytest.log" 2>&1; echo "pytest rc $?" >> "$out/pytest.log"
grep -E "x rel|clip |passed|failed|rc" "$out/pytest.log" | tail -14
timeout 300 python tests/tools/kinopt_bench.py 64 30 0 > "$out/bench_64x30.json" 2> "$out/bench_64x30.err"; tail -c 1500 "$out/bench_64x30.json"
timeout 600 python tests/tools/kinopt_bench.py 256 100 12 > "$out/bench_256x100.json" 2> "$out/bench_256x100.err"; tail -c 1500 "$out/bench_256x100.json"; tail -3 "$out/bench_256x100.err"
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$GRAFT_REPO_ROOT/$out/trace" -- python "$GRAFT_REPO_ROOT/tests/tools/kinopt_bench.py" 64 30 0 > /dev/null 2>&1)
find "$out/trace" -name '*kernel_stats.csv' | head -1 | xargs -r -I{} cp {} "$out/trace_kernel_stats.csv"
rm -rf "$out/trace"
head -5 "$out/trace_kernel_stats.csv" 2>/dev/null
git rev-parse HEAD > "$out/COMMIT" 2>/dev/null || true

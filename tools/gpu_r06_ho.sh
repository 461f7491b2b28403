#!/bin/bash
# round 6: the hand-over panel loop -- solver GPU tests + a quick bench line
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/${1:-r06h}
mkdir -p $OUT
cd $R
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_pipelined.py tests/test_quality_gate.py -x -q -m gpu > $OUT/pytest_gpu.log 2>&1; tail -15 $OUT/pytest_gpu.log
bash tools/gpu_quick.sh ${2:-new}

mkdir -p gpurun_out/r05y
timeout 400 python -m pytest tests/test_kinopt_gpu.py -x -q 2>&1 | grep -v "^  File\|Extension" | tail -3
for i in 1 2 3 4; do
timeout 200 python tests/tools/kinopt_bench.py 256 100 0 > gpurun_out/r05y/k_256x100_$i.json 2> gpurun_out/r05y/k_256x100_$i.err; echo run $i rc $? $(head -c 330 gpurun_out/r05y/k_256x100_$i.json | cut -c60-330); grep -v amdgpu.ids gpurun_out/r05y/k_256x100_$i.err | tail -1 | cut -c1-200
done
for cfg in "64 100" "32 60"; do
  set -- $cfg
  timeout 200 python tests/tools/kinopt_bench.py $1 $2 0 > gpurun_out/r05y/k_$1x$2.json 2>/dev/null; head -c 330 gpurun_out/r05y/k_$1x$2.json | cut -c60-330; echo
done

mkdir -p gpurun_out/r05r
timeout 240 python -m pytest tests/test_kinopt_gpu.py -x -q 2>&1 | grep -v "^  File\|Extension" | tail -3
cd contact-human-dynamics_amd/csrc
hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -std=c++17 -fPIC -shared -DKIN_PROFILE chd_kinopt.hip -o libchd_kinopt.so 2>/dev/null
cd ../..
timeout 200 python tests/tools/kinopt_bench.py 128 100 0 > gpurun_out/r05r/kin_prof.json 2> gpurun_out/r05r/kin_prof.err; grep KIN_PROFILE gpurun_out/r05r/kin_prof.err | tail -1; head -c 400 gpurun_out/r05r/kin_prof.json

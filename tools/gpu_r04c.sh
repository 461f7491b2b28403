#!/bin/bash
# round 4, kernel iterations: the solver's parity tests + the driver's bench line (tag = $1)
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; O=gpurun_out/${1:-r04c}; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -x > $O/pytest_gpu.log 2>&1; tail -3 $O/pytest_gpu.log
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 ${BENCH_FLAGS} > $O/bench_driver.json 2> $O/bench.err; tail -3 $O/bench.err
python - <<P
import json
try:
    d=json.load(open('$O/bench_driver.json')); c=d['config']
    print('bench value', d['value'], 'iters/seq', c['ipm_iterations_per_sequence'], c['ipm_iterations_rank0'], 'fallbacks', c['stage4_fallbacks_rank0'], 'busy', d['roofline']['kernel_busy_fraction'])
    print('incl setup', c.get('value_including_setup'), c.get('setup_ms_per_sequence'), c.get('inclusive_run_error'))
    print('incl io', c.get('value_including_file_io'))
    print('500', c.get('value_500_sequences_in_one_call'), c.get('kernel_busy_fraction_500_sequences'), 'long', c.get('long_600_frames'), c.get('side_run_error'))
    print('parity', d.get('parity')); print('share', c['in_kernel_time_share'])
    print('phase ms', c.get('in_kernel_phase_ms_per_sequence'))
except Exception as e: print('bench parse failed', e)
P

#!/bin/bash
# round 4, second GPU call: positional pivot test with the oracle's ordering taken from the structure, asynchronous chunk launches, per-clip statistics of the 64 x 100 pipeline
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; O=gpurun_out/${1:-r04b}; mkdir -p $O
timeout 900 python -m pytest tests -q -m gpu -x > $O/pytest_gpu.log 2>&1; tail -3 $O/pytest_gpu.log
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver.json 2> $O/bench.err; tail -3 $O/bench.err
rm -rf /tmp/pipe100; timeout 600 python tests/tools/pipeline_bench.py --videos 64 --frames 100 --keep /tmp/pipe100 > $O/pipe100.json 2> $O/pipe100.err; tail -2 $O/pipe100.err
timeout 300 python tests/tools/pipe_phys_stats.py /tmp/pipe100/data 100 $O/pipe100_slow 16 2>&1 | tail -2
python - <<P
import json
try:
    d=json.load(open('$O/bench_driver.json')); c=d['config']
    print('bench value', d['value'], 'iters/seq', c['ipm_iterations_per_sequence'], c['ipm_iterations_rank0'], 'fallbacks', c['stage4_fallbacks_rank0'], 'busy', d['roofline']['kernel_busy_fraction'])
    print('incl setup', c.get('value_including_setup'), c.get('setup_ms_per_sequence'), c.get('including_setup'), c.get('inclusive_run_error'))
    print('incl io', c.get('value_including_file_io'), c.get('including_file_io'))
    print('500', c.get('value_500_sequences_in_one_call'), c.get('kernel_busy_fraction_500_sequences'), 'long', c.get('long_600_frames'), c.get('side_run_error'))
    print('parity', d.get('parity')); print('share', c['in_kernel_time_share'])
    print('pipeline', d.get('pipeline'))
except Exception as e: print('bench parse failed', e)
try: print('pipe100', json.load(open('$O/pipe100.json')))
except Exception as e: print('pipe100 parse failed', e)
P

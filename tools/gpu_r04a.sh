#!/bin/bash
# round 4, first GPU call: the new solver model (exact node x duration block, Gauss-Newton second model) + the pipelined whole-call path on the MI355X
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; O=gpurun_out/${1:-r04a}; mkdir -p $O
timeout 900 python -m pytest tests -q -m gpu -x > $O/pytest_gpu.log 2>&1; tail -3 $O/pytest_gpu.log
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver.json 2> $O/bench.err; tail -3 $O/bench.err
timeout 600 python tests/tools/pipeline_bench.py --videos 64 --frames 100 > $O/pipe100.json 2> $O/pipe100.err; tail -2 $O/pipe100.err
python - <<P
import json
try:
    d=json.load(open('$O/bench_driver.json')); c=d['config']
    print('bench value', d['value'], 'iters/seq', c['ipm_iterations_per_sequence'], c['ipm_iterations_rank0'], 'fallbacks', c['stage4_fallbacks_rank0'], 'busy', d['roofline']['kernel_busy_fraction'])
    print('incl setup', c.get('value_including_setup'), c.get('setup_ms_per_sequence'), c.get('including_setup'), c.get('inclusive_run_error'))
    print('incl io', c.get('value_including_file_io'), c.get('including_file_io'))
    print('500', c.get('value_500_sequences_in_one_call'), c.get('kernel_busy_fraction_500_sequences'), 'long', c.get('long_600_frames'), c.get('side_run_error'))
    print('parity', d.get('parity')); print('share', c['in_kernel_time_share'])
    print('pipeline', d.get('pipeline')); print('kinopt', d['kinematic_optimisation'].get('clips_per_s'), 'contact', d['contact_net'].get('fps'), d['contact_net'].get('roofline'))
except Exception as e: print('bench parse failed', e)
try: print('pipe100', json.load(open('$O/pipe100.json')))
except Exception as e: print('pipe100 parse failed', e)
P

#!/bin/bash
# experiment: two 256-thread workgroups per compute unit (76 KB of LDS each) against the shipped one 512-thread workgroup
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/exp256
python bench.py --steps 8 --warmup 1 --no-cpu-baseline > gpurun_out/exp256/base.json 2> gpurun_out/exp256/base.err
CHD_EXPERIMENTAL_256=1 timeout 600 python bench.py --steps 8 --warmup 1 --no-cpu-baseline --threads 256 --lds-kb 76 --max-workgroups 512 > gpurun_out/exp256/t256.json 2> gpurun_out/exp256/t256.err
tail -3 gpurun_out/exp256/t256.err
python - <<'P'
import json
for n in ('base','t256'):
    try:
        d=json.load(open('gpurun_out/exp256/%s.json'%n))
        print(n, 'value %.1f'%d['value'], 'iters %.2f'%d['config']['ipm_iterations_per_sequence'], d['config']['converged_rank0'], 'busy %.2f'%d['roofline']['kernel_busy_fraction'], 'mean seq ms %.0f'%d['config']['mean_sequence_ms'], {k:d['parity'][k] for k in ('sequences_compared','worst_rel_l2','stage_iterations_equal','sequences_above_1e-3')} if 'parity' in d and 'error' not in d['parity'] else d.get('parity'))
        print('   shares', {k: round(v,3) for k,v in d['config']['in_kernel_time_share'].items()})
    except Exception as e: print(n, 'failed', e)
P

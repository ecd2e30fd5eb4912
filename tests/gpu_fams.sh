#!/bin/bash
# usage: gpu_fams.sh MODEL B: every kernel family of the step at batch B (vae_time_step, eager): us per step and GB/s
timeout -k 10 200 python bench.py --model $1 --batch $2 --steps 30 --warmup 5 --no-cpu-baseline --no-strong-ref --roofline-iters 5 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); r=d['roofline']
print('ms/step %.4f launches %d sum %.1f us' % (d['ms_per_step'], r['step_launches'], r['step_sum_of_launches_us']))
for f in r['families']: print('  %-44s x%d %8.1f us %8.1f GB/s' % (f['kernel'], f['launches'], f['us_per_step'], f['GBps']))"

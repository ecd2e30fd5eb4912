#!/bin/bash
# usage: gpu_ab.sh VAR "v1 v2 ..." [bench args]: ms/step of the default bench under each value of the environment variable
VAR=$1; VALS=$2; shift 2
for v in $VALS; do
  for rep in 1 2; do
    env $VAR=$v timeout -k 10 200 python bench.py --no-cpu-baseline --no-strong-ref --steps 300 --warmup 30 --roofline-iters 3 "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); r=d['roofline']
print('$VAR=$v', 'ms/step %.4f' % d['ms_per_step'], 'launches', r['step_launches'], 'sum_us %.1f' % r['step_sum_of_launches_us'])" || exit 1
  done
done

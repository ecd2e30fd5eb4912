#!/bin/bash
# usage: gpu_sweep.sh MODEL "B1 B2 ...": ms/step and samples/s per batch size (whole step as one hipGraph)
M=$1; BS=$2
for b in $BS; do
  timeout -k 10 200 python bench.py --model $M --batch $b --steps 50 --warmup 10 --no-cpu-baseline --no-strong-ref --roofline-iters 3 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline())
print('$M B=$b ms/step %.4f  %.2f M samples/s  dominant %s %.1f us' % (d['ms_per_step'], d['value']/1e6, d['roofline'].get('kernel'), d['roofline'].get('avg_launch_us') or 0))" || echo "$b failed"
done

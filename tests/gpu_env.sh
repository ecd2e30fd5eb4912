#!/bin/bash
# usage: gpu_env.sh VAR "v1 v2 ..." PATTERN [bench args]: ms/step and matching kernel families under each value of VAR
VAR=$1; VALS=$2; PAT=$3; shift 3
for v in $VALS; do
  env $VAR=$v timeout -k 10 200 python bench.py --no-cpu-baseline --no-strong-ref --steps 200 --warmup 20 --roofline-iters 10 "$@" 2>/dev/null | python -c "
import json,sys,re
d=json.loads(sys.stdin.readline()); r=d['roofline']
fam=[f for f in r['families'] if re.search('$PAT', f['kernel'])]
print('$VAR=$v', 'ms/step %.4f' % d['ms_per_step'], ' '.join('%s=%.1f' % (f['kernel'], f['us_per_step']) for f in fam))" || echo "$v failed"
done

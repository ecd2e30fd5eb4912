#!/bin/bash
# usage: gpu_fam.sh "lib1 lib2 ..." PATTERN [bench args]: per library (build_dbg/libvae_<name>.so, "base" = the in-tree
# one): ms/step and the per-step time of the kernel families whose name matches PATTERN (vae_time_step, eager launches)
LIBS=$1; PAT=$2; shift 2
for n in $LIBS; do
  if [ $n = base ]; then unset VAE_AMD_LIB; else export VAE_AMD_LIB=$PWD/build_dbg/libvae_$n.so; fi
  timeout -k 10 200 python bench.py --no-cpu-baseline --no-strong-ref --steps 200 --warmup 20 --roofline-iters 10 "$@" 2>/dev/null | python -c "
import json,sys,re
d=json.loads(sys.stdin.readline()); r=d['roofline']
fam=[f for f in r['families'] if re.search('$PAT', f['kernel'])]
print('$n', 'ms/step %.4f' % d['ms_per_step'], ' '.join('%s=%.1f' % (f['kernel'], f['us_per_step']) for f in fam))" || echo "$n failed"
done

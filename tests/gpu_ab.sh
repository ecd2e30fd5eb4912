#!/bin/bash
# A/B of two builds of the library on ONE box (step times differ by 1-2 % from box to box): alternates the builds.
# usage: tests/gpu_ab.sh <old.so> [rounds] [model]   (the new build is the in-tree library)
OLD=$1; R=${2:-3}; M=${3:-mnist}
for i in $(seq $R); do
  for w in old new; do
    if [ $w = old ]; then export VAE_AMD_LIB=$OLD; else unset VAE_AMD_LIB; fi
    timeout -k 10 120 python bench.py --model $M --no-cpu-baseline --no-strong-ref --roofline-iters 5 > gpurun_out/ab_$w.json || exit 1
    python - $w <<'PY'
import json, sys
d = json.load(open(f"gpurun_out/ab_{sys.argv[1]}.json"))
fam = {f["kernel"]: f["us_per_step"] for f in d["roofline"]["families"]}
top = sorted(fam.items(), key=lambda kv: -kv[1])[:4]
print(sys.argv[1], round(d["ms_per_step"], 4), [(k[:22], v) for k, v in top], flush=True)
PY
  done
done

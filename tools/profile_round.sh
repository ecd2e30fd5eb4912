#!/bin/bash
# usage (on the GPU box, from the repo root): tools/profile_round.sh TAG [models...]
# For every model: bench line -> gpurun_out/TAG_<model>_bench.json, rocprofv3 --kernel-trace --stats of the same command
# -> gpurun_out/TAG_<model>_prof/, and two separate --pmc passes (FETCH_SIZE, WRITE_SIZE; eager launches so that every
# dispatch is attributed) -> gpurun_out/TAG_<model>_pmc_{fetch,write}/.  Copy the summaries into profiles/ afterwards
# (tools/collect_profiles.py).
TAG=${1:-r02}; shift
MODELS=${@:-mnist multiscale tcvae wide}
R=${GRAFT_REPO_ROOT:-$PWD}
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
for M in $MODELS; do
  STEPS=100; [ $M = wide ] && STEPS=40
  timeout -k 10 300 python3 $R/bench.py --model $M --steps $STEPS --warmup 10 > $R/gpurun_out/${TAG}_${M}_bench.json 2> $R/gpurun_out/${TAG}_${M}_bench.err || { tail -5 $R/gpurun_out/${TAG}_${M}_bench.err; exit 1; }
  rm -rf $R/gpurun_out/${TAG}_${M}_prof
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${TAG}_${M}_prof -- python3 $R/bench.py --model $M --steps $STEPS --warmup 10 --no-cpu-baseline --no-strong-ref > $R/gpurun_out/${TAG}_${M}_prof.log 2>&1 || { tail -5 $R/gpurun_out/${TAG}_${M}_prof.log; exit 1; }
  for c in FETCH_SIZE WRITE_SIZE; do
    d=$R/gpurun_out/${TAG}_${M}_pmc_$(echo $c | tr A-Z a-z | cut -d_ -f1)
    rm -rf "$d"
    timeout -s KILL 180 rocprofv3 --pmc $c --kernel-trace -d "$d" -o pmc --output-format csv -- python3 $R/bench.py --model $M --steps 3 --warmup 1 --no-graph --no-cpu-baseline --no-strong-ref --roofline-iters 1 > "$d.log" 2>&1 || { tail -5 "$d.log"; exit 1; }
  done
  if [ $M = mnist ] || [ $M = wide ]; then   # SQ counters (one pass holds at most 8 of them): tools/sq_summary.py reads the result
    d=$R/gpurun_out/${TAG}_${M}_sq
    rm -rf "$d"
    timeout -s KILL 180 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_WAIT_INST_LDS --kernel-trace -d "$d" -o pmc --output-format csv -- python3 $R/bench.py --model $M --steps 3 --warmup 1 --no-graph --no-cpu-baseline --no-strong-ref --roofline-iters 1 > "$d.log" 2>&1 || { tail -5 "$d.log"; exit 1; }
  fi
  echo "$M done"
done

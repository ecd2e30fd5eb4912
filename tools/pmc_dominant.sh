#!/bin/bash
# HBM traffic of the dominant kernel: two separate rocprofv3 --pmc passes (FETCH_SIZE and WRITE_SIZE do not fit
# one pass), eager launches (no hipGraph) so every dispatch is attributed.  Output: gpurun_out/pmc_{fetch,write}.
set -e
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  d=gpurun_out/pmc_$(echo $c | tr A-Z a-z | cut -d_ -f1)
  rm -rf "$d"
  timeout -s KILL 180 rocprofv3 --pmc $c --kernel-trace -d "$d" -o pmc --output-format csv -- \
    python3 bench.py --steps 3 --warmup 1 --no-graph --no-cpu-baseline --roofline-iters 1 > "$d.log" 2>&1
done
ls gpurun_out/pmc_fetch gpurun_out/pmc_write

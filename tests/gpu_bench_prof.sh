#!/bin/bash
# usage: gpu_bench_prof.sh TAG  -> gpurun_out/bench_TAG.json (default bench line, no CPU baseline) and
# gpurun_out/prof_TAG/ (rocprofv3 --kernel-trace --stats of the same bench command)
TAG=${1:-x}
mkdir -p gpurun_out
timeout -k 10 300 python bench.py --steps 100 --warmup 10 --no-cpu-baseline > gpurun_out/bench_$TAG.json 2> gpurun_out/bench_$TAG.err || { tail -5 gpurun_out/bench_$TAG.err; exit 1; }
cat gpurun_out/bench_$TAG.json
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$TAG -- python3 $R/bench.py --steps 100 --warmup 10 --no-cpu-baseline ${PROF_EXTRA} > $R/gpurun_out/prof_$TAG.log 2>&1 || { tail -5 $R/gpurun_out/prof_$TAG.log; exit 1; }

#!/bin/bash
# one GPU call: parity tests (all failures reported), log kept under gpurun_out/
mkdir -p gpurun_out
timeout -k 10 900 python -u -m pytest tests -m gpu -v --timeout 120 --timeout-method thread -q "$@" > gpurun_out/pytest_gpu.log 2>&1
rc=$?
tail -40 gpurun_out/pytest_gpu.log
exit $rc

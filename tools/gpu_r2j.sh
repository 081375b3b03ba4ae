#!/bin/bash
set -u
tools/gpu_quick.sh r2j --no-other-configs --no-end-to-end || exit 1
timeout -k 10 500 python -u tools/soak_gpu.py 120 9000 > gpurun_out/soak_r2j.log 2>&1; rc=$?; tail -3 gpurun_out/soak_r2j.log; exit $rc

#!/bin/bash
set -u
timeout -k 10 400 python -u -m pytest tests -x -v --timeout 180 --timeout-method thread -m gpu -k "cam1 or caller_supplied or main_cpp_shaped" > gpurun_out/pytest_cam1.log 2>&1
rc=$?; tail -8 gpurun_out/pytest_cam1.log
[ $rc -ne 0 ] && { grep -E "^(E|FAILED)" gpurun_out/pytest_cam1.log | head -20; exit $rc; }
tools/gpu_quick.sh r2f --no-other-configs --no-end-to-end

#!/bin/bash
# bench contract tests (one rank, two ranks on gloo), then the profile passes of the round's current build
set -u
timeout -k 10 400 python -u -m pytest tests/test_parity_gpu.py -x -v --timeout 180 --timeout-method thread -k "bench_line or two_ranks" > gpurun_out/pytest_bench.log 2>&1
rc=$?; tail -4 gpurun_out/pytest_bench.log
[ $rc -ne 0 ] && { grep -E "^(E|FAILED)" gpurun_out/pytest_bench.log | head -20; tail -30 gpurun_out/pytest_bench.log; exit $rc; }
bash tools/profile_gpu.sh ${1:-r02a}

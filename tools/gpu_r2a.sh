#!/bin/bash
# round 2, first GPU call: new tests (no-alloc entry point, bench shape vs oracle, queue drains), then the suite + bench
set -u
timeout -k 10 300 python -u -m pytest tests/wip_no_alloc.py -x -v --timeout 120 --timeout-method thread > gpurun_out/pytest_noalloc.log 2>&1
rc=$?; tail -5 gpurun_out/pytest_noalloc.log
[ $rc -ne 0 ] && { grep -E "^(E|FAILED)" gpurun_out/pytest_noalloc.log | head -20; exit $rc; }
timeout -k 10 300 python -u -m pytest tests/test_parity_gpu.py -x -v --timeout 120 --timeout-method thread -k "bench_shape or detect_queue" > gpurun_out/pytest_new.log 2>&1
rc=$?; tail -5 gpurun_out/pytest_new.log
[ $rc -ne 0 ] && { grep -E "^(E|FAILED)" gpurun_out/pytest_new.log | head -20; exit $rc; }
tools/gpu_quick.sh r2a

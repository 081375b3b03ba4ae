#!/bin/bash
set -u
timeout -k 10 400 python -u -m pytest tests -x -v --timeout 180 --timeout-method thread -m gpu -k "match" > gpurun_out/pytest_match.log 2>&1
rc=$?; tail -5 gpurun_out/pytest_match.log
[ $rc -ne 0 ] && { grep -E "^(E|FAILED)" gpurun_out/pytest_match.log | head -20; exit $rc; }
exit 0

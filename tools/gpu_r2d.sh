#!/bin/bash
set -u
timeout -k 10 60 tools/ubench/mfma_i8_rate || exit 1
rocprofv3 -L 2>/dev/null | grep -oE "\b(SQ_[A-Z_0-9]*MFMA[A-Z_0-9]*|SQ_VALU_MFMA_BUSY_CYCLES|SQ_BUSY_CU_CYCLES)\b" | sort -u | tr '\n' ' '; echo
BENCH_ARGS="--no-other-configs --no-end-to-end --no-overlap" bash tools/pmc.sh mfma1 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAVE_CYCLES GRBM_GUI_ACTIVE

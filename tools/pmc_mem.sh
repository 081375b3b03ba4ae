#!/bin/bash
# GPU box: memory-path counters (vector L1, L2, texture addresser) of the bench kernels, one pass each.
# usage: tools/pmc_mem.sh <tag>
TAG=${1:-m}
export BENCH_ARGS="--no-other-configs --no-end-to-end"
bash tools/pmc.sh ${TAG}_tcp TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_WRITE_REQ_sum GRBM_GUI_ACTIVE &&
bash tools/pmc.sh ${TAG}_tcc TCC_HIT_sum TCC_MISS_sum TCC_EA0_WRREQ_STALL_sum TCC_TAG_STALL_sum &&
bash tools/pmc.sh ${TAG}_tcc2 TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_REQ_sum &&
bash tools/pmc.sh ${TAG}_ta TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum GRBM_GUI_ACTIVE

#!/bin/bash
# Device ISA of one kernel: tools/isa.sh <mangled-name-substring> [out.s]
# also prints the resource-usage remarks (VGPRs, occupancy) of every kernel with -v
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p /tmp/isa
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++20 -fconstexpr-steps=200000000 $ISA_FLAGS --cuda-device-only -S -o /tmp/isa/all.s "$ROOT/sift_amd/csrc/sift_hip.hip" -Rpass-analysis=kernel-resource-usage 2> /tmp/isa/remarks.txt || { cat /tmp/isa/remarks.txt | grep error; exit 1; }
awk -v pat="$1" '$0 ~ "^_Z.*" pat ".*:" {p=1} p {print} p && /s_endpgm/ {exit}' /tmp/isa/all.s | grep -v "^\s*;" > "${2:-/tmp/isa/k.s}"
grep -A10 "Function Name: .*$1" /tmp/isa/remarks.txt | grep -E "Function Name|VGPRs:|SGPRs:|Occupancy|LDS Size|ScratchSize" | sed 's/.*remark: [^ ]* *//; s/\[-Rpass.*//' | head -12
wc -l "${2:-/tmp/isa/k.s}"

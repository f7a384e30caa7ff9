#!/bin/bash
# register / scratch use of every k_multi instantiation (cross-compile, no GPU needed)
#   usage: scripts/res_usage.sh [R digits, default 56]
cd "$(dirname "$0")/../qcmrf_amd/csrc" || exit 1
: > /tmp/res_usage.txt
for f in qsv_kmulti_m0_low qsv_kmulti_m0_r3 qsv_kmulti_m0_r4 qsv_kmulti_m0_r5 qsv_kmulti_m1 qsv_kmulti_m2; do
  hipcc -O3 --offload-arch=gfx950 -std=c++17 -munsafe-fp-atomics -Wno-unused-value -Wno-unused-result --cuda-device-only -c \
    -Rpass-analysis=kernel-resource-usage -o /dev/null $f.hip 2>> /tmp/res_usage.txt &
done; wait
grep -A12 "Function Name: _Z7k_multiILi[${1:-56}]E" /tmp/res_usage.txt | grep -E "Function Name|VGPRs:|Scratch|Occupancy" \
  | sed 's/.*remark: *//; s/\[-Rpass.*//' | paste - - - - \
  | sed 's/Function Name: _Z7k_multiILi\([0-9]\)ELb\([01]\)ELi\([0-9]\)ELb\([01]\).*VGPRs/R=\1 INIT=\2 MODE=\3 NT=\4 VGPRs/' | grep "^R=" | sort -u

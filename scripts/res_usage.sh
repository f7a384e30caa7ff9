#!/bin/bash
# register / scratch use of every k_multi instantiation (cross-compile, no GPU needed)
cd "$(dirname "$0")/../qcmrf_amd/csrc" || exit 1
hipcc -O3 --offload-arch=gfx950 -std=c++17 -fPIC -shared -munsafe-fp-atomics -Wno-unused-value -Wno-unused-result \
  -Rpass-analysis=kernel-resource-usage -o /tmp/libqsv_probe.so qsv.hip -ldl 2> /tmp/res_usage.txt
grep -A12 "Function Name: _Z7k_multiILi[${1:-56}]E" /tmp/res_usage.txt | grep -E "Function Name|VGPRs:|Scratch|Occupancy" \
  | sed 's/.*remark: *//; s/\[-Rpass.*//' | paste - - - - \
  | sed 's/Function Name: _Z7k_multiILi\([0-9]\)ELb\([01]\)ELi\([0-9]\).*VGPRs/R=\1 INIT=\2 MODE=\3 VGPRs/' | grep "^R="

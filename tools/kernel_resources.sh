#!/bin/bash
# registers, LDS, scratch and occupancy of the kernels whose name matches $1 (default: the gather kernels), from the compiler's remarks
# usage: tools/kernel_resources.sh [regex] [extra hipcc flags]
cd "$(dirname "$0")/../raytracer_amd/csrc"
pat=${1:-k_gather}; shift
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Wno-unused-function "$@" -Rpass-analysis=kernel-resource-usage --cuda-device-only -c pm_api.hip -o /dev/null 2>&1 |
  grep -E "Function Name|VGPRs:|ScratchSize|Occupancy|LDS Size|SGPRs Spill" | sed -E 's/.*remark: +//; s/ \[-Rpass.*//; s/Function Name: _ZN2pm[0-9]+//' | paste - - - - - - |
  awk -F'\t' '{printf "%-30s %s | %s | %s | %s | %s\n", substr($1,1,30), $2,$3,$4,$5,$6}' | grep -E "$pat" | sort | uniq

#!/bin/bash
# Register / LDS / spill report of the kernels of one source file (hipcc -Rpass-analysis=kernel-resource-usage):
#   scripts/kernel_resources.sh conv3d_k3.hip [name filter]
set -eo pipefail
SRC=${1:-conv3d_k3.hip}
FILT=${2:-.}
cd "$(dirname "$0")/../bodyct-dram_amd/csrc"
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -Rpass-analysis=kernel-resource-usage \
    -c "$SRC" -o /tmp/kr_$$.o 2>&1 | grep -E "Function Name|VGPRs:|AGPRs|Spill|Occupancy|LDS Size|SGPRs:" | \
    awk -v f="$FILT" '/Function Name/ {show = ($0 ~ f)} show {sub(/^.*remark: [^ ]* /, ""); print}'
rm -f /tmp/kr_$$.o

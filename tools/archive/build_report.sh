#!/bin/bash
# Builds csrc/libssp_hip.so with -Rpass-analysis=kernel-resource-usage and prints the register / spill lines of the kernels
# whose (mangled) name matches $1.
cd "$(dirname "$0")/../semantic-superpoint_amd/csrc" || exit 1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -munsafe-fp-atomics -Rpass-analysis=kernel-resource-usage ${EXTRA_FLAGS} ssp.hip -o ${OUT:-libssp_hip.so} 2> /tmp/build.log
grep -E " error" /tmp/build.log | head
grep -A12 "Function Name: .*${1:-conv_wino_p2}" /tmp/build.log | grep -E "Function Name|VGPRs:|Spill|TotalSGPRs|ScratchSize" | sed 's/.*remark: *//; s/ \[-Rpass.*//'

#!/bin/bash
# perf-debug: device assembly of conv_bf16_ws_kernel alone (2 s) -> /tmp/wsx/k1.s (forward form), memory operations and waits listed
mkdir -p /tmp/wsx && cd /tmp/wsx
printf '#include "pk_math.hip.h"\n#include "conv_bf16_ws.hip.h"\ntemplate __global__ void sspk::conv_bf16_ws_kernel<1, true>(const sspk::ConvBArgs);\ntemplate __global__ void sspk::conv_bf16_ws_kernel<0, true>(const sspk::ConvBArgs);\n' > k.hip
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -S --cuda-device-only -munsafe-fp-atomics -I/root/repo/semantic-superpoint_amd/csrc k.hip -o k.s 2>&1 | grep -E "error" | head
a=$(grep -n "^_ZN4sspk19conv_bf16_ws_kernelILi1ELb1EEEvNS_9ConvBArgsE:" k.s | cut -d: -f1); b=$(grep -n "^_ZN4sspk19conv_bf16_ws_kernelILi0ELb1EEEvNS_9ConvBArgsE:" k.s | cut -d: -f1)
awk -v a=$a -v b=$b 'NR>=a && NR<b' k.s > k1.s
grep -n "NumVgprs\|ScratchSize" k1.s
grep -n "s_waitcnt vmcnt\|s_barrier\|buffer_load\|buffer_store\|global_load\|global_store" k1.s

#!/bin/bash
# Device assembly of csrc/ssp.hip -> /tmp/ssp.s, then the instruction mix of the MFMA loop of the kernel whose mangled
# name is $1 (tools/isa_mix.py).
cd "$(dirname "$0")/../semantic-superpoint_amd/csrc" || exit 1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -S --cuda-device-only -munsafe-fp-atomics ${EXTRA_FLAGS} ssp.hip -o /tmp/ssp.s 2>&1 | grep -v hip-link | head -30
[ -n "$1" ] && python3 /root/repo/tools/isa_mix.py "$1"

#!/bin/bash
# Builds the HIP library of a git revision into ab/libssp_base.so for same-box A/B runs (tools/ab_bench.sh, tools/ab_kernels.sh:
# SSP_HIP_LIB selects it).  usage: tools/build_base.sh [rev]   (default HEAD)
set -eu
R=$(cd "$(dirname "$0")/.." && pwd)
REV=${1:-HEAD}
T=$(mktemp -d /tmp/ssp_base.XXXXXX)
git -C "$R" archive "$REV" semantic-superpoint_amd/csrc include | tar -x -C "$T"
mkdir -p "$R/ab"
(cd "$T/semantic-superpoint_amd/csrc" && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -munsafe-fp-atomics ssp.hip -o "$R/ab/libssp_base.so")
rm -rf "$T"
rm -f "$R/ab/libssp_base.so.isa_ok"
echo "ab/libssp_base.so <- $REV"

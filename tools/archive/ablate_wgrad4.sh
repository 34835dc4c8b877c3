#!/bin/bash
# Compile-time ablation of wgrad_wino4_kernel (WG4_ABL bits, wgrad_wino4.hip.h).
#   build (container, no GPU):  tools/ablate_wgrad4.sh build "0 1 2 4 6 8 16 24"   -> ab/libssp_wg4_<bits>.so
#   run   (GPU box):            tools/ablate_wgrad4.sh run   "0 1 2 4 6 8 16 24"   -> one line per variant (64->64 @240x320, 32 images)
# Ablated libraries break the accumulation-register contract on purpose (no MFMAs ...): the timing script loads them with
# SSP_SKIP_ISA_VERIFY=1; their results are garbage, only their time is read.
set -u
R=$(cd "$(dirname "$0")/.." && pwd)
MODE=$1; LIST=${2:-"0 1 2 4 6 8 16"}
if [ "$MODE" = build ]; then
  mkdir -p $R/ab
  for b in $LIST; do
    (cd $R/semantic-superpoint_amd/csrc && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -munsafe-fp-atomics -DWG4_ABL=$b ssp.hip -o $R/ab/libssp_wg4_$b.so) &
    if (( $(jobs -r | wc -l) >= 4 )); then wait -n; fi
  done
  wait
  ls -la $R/ab/libssp_wg4_*.so
else
  for b in $LIST; do
    SSP_SKIP_ISA_VERIFY=1 SSP_HIP_LIB=$R/ab/libssp_wg4_$b.so python3 $R/tools/wgrad_probe.py "WG4_ABL=$b" ${3:-}
  done
fi

#!/bin/bash
# conv_wino4_kernel with constants instead of its fused BatchNorm-backward tensor loads (W4_ABL=2048) beside the shipped kernel
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/w4bnr; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
run() {
  rocprofv3 --kernel-trace --stats -d $O/p_$1 -o n -- python3 $R/bench.py --no-cpu-baseline --traffic none --no-roofline --no-export --steps 4 --warmup 1 > /dev/null 2>&1
  find $O/p_$1 -name "*results.db" | head -1 | xargs -I{} python3 $R/tools/rocpd_stats.py {} 80 | grep -E "conv_wino4" | cut -c1-60,95-150 | sed "s/^/$1  /"
  rm -rf $O/p_$1
}
run full
export SSP_SKIP_ISA_VERIFY=1 SSP_HIP_LIB=$R/ab/libssp_w4abl_2048.so
run nobnrloads

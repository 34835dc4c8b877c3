#!/bin/bash
# Compile-time ablation of conv1x1_group_kernel (G1_ABL, conv1x1_group.hip.h): kernel times of the variant libraries
# ab/libssp_g1abl_<n>.so beside the shipped one.  build: for v in 1 2 4 3; do hipcc ... -DG1_ABL=$v ssp.hip -o ab/libssp_g1abl_$v.so; usage (on a GPU box): tools/ablate_g1.sh "1 2 4 3"
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/g1abl; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
run() { # tag
  rocprofv3 --kernel-trace --stats -d $O/p_$1 -o n -- python3 $R/bench.py --no-cpu-baseline --traffic none --no-roofline --no-export --steps 4 --warmup 1 > /dev/null 2>&1
  find $O/p_$1 -name "*results.db" | head -1 | xargs -I{} python3 $R/tools/rocpd_stats.py {} 80 | grep -E "conv1x1" | cut -c1-60,95-140 | sed "s/^/$1  /"
  rm -rf $O/p_$1
}
run full
export SSP_SKIP_ISA_VERIFY=1
for v in $1; do export SSP_HIP_LIB=$R/ab/libssp_g1abl_$v.so; run abl$v; done

#!/bin/bash
# Sweep of the epilogue term of launch_g1's cost model (CU shares of the grouped pointwise problems): kernel times per setting.
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/g1cost; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
for e in "500 1500" "1000 2500" "2000 4000" "3000 6000" "4000 8000"; do
  set -- $e; export SSP_G1_ECOST=$1 SSP_G1_ECOST_BNR=$2
  rocprofv3 --kernel-trace --stats -d $O/p -o n -- python3 $R/bench.py --no-cpu-baseline --traffic none --no-roofline --no-export --steps 4 --warmup 1 > /dev/null 2>&1
  find $O/p -name "*results.db" | head -1 | xargs -I{} python3 $R/tools/rocpd_stats.py {} 80 | grep -E "conv1x1" | cut -c1-60,95-140 | sed "s/^/E=$1,$2  /"
  rm -rf $O/p
done

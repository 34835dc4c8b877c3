#!/bin/bash
# kernel-trace statistics of the bf16 path's pair step (run through gpurun from the repo root) -> gpurun_out/pbf16/kernel_stats.txt
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/pbf16
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
Q="--conv-algo 12 --no-cpu-baseline --traffic none --no-export --no-roofline"
rocprofv3 --kernel-trace --stats -d $O/kt -o k -- python3 $R/bench.py $Q --steps 6 --warmup 1 > $O/bench_line.txt 2>/dev/null
cd $R
find $O/kt -name "*results.db" | head -1 | xargs -I{} python3 tools/rocpd_stats.py {} 60 > $O/kernel_stats.txt
rm -rf $O/kt

#!/bin/bash
# Dispatch timeline of the last pair step (every launch with its duration and the gap before it): fp32 and bf16 paths
# (run through gpurun from the repo root) -> gpurun_out/timeline/{f32,bf16}.txt       usage: tools/timeline_step.sh [bench args]
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/timeline
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
Q="--no-cpu-baseline --traffic none --no-export --no-roofline --no-bf16 --no-sp $*"
for dt in f32 bf16; do
  rocprofv3 --kernel-trace -d $O/kt_$dt -o k -- python3 $R/bench.py --dtype $dt $Q --steps 3 --warmup 1 > /dev/null 2>&1
  find $O/kt_$dt -name "*results.db" | head -1 | xargs -I{} python3 $R/tools/rocpd_timeline.py {} > $O/$dt.txt
  rm -rf $O/kt_$dt
done

#!/bin/bash
# Same-box per-kernel A/B: rocprofv3 kernel stats of a short bench run with the in-tree library and with ab/libssp_base.so.
# usage: tools/ab_kernels.sh <tag> [grep pattern]   -> gpurun_out/<tag>/{new,base}_kernels.txt
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
T=$1; PAT=${2:-.}
O=$R/gpurun_out/$T
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O/p_new -o n -- python3 $R/bench.py --no-cpu-baseline --traffic none --no-roofline --no-export --steps 6 --warmup 1 > $O/p_new.json 2>/dev/null
export SSP_HIP_LIB=$R/ab/libssp_base.so SSP_SKIP_ISA_VERIFY=1  # (verified when it was HEAD; the contract may have moved on)
rocprofv3 --kernel-trace --stats -d $O/p_base -o b -- python3 $R/bench.py --no-cpu-baseline --traffic none --no-roofline --no-export --steps 6 --warmup 1 > $O/p_base.json 2>/dev/null
unset SSP_HIP_LIB SSP_SKIP_ISA_VERIFY
cd $R
find $O/p_new -name "*results.db" | head -1 | xargs -I{} python3 tools/rocpd_stats.py {} 60 > $O/new_kernels.txt
find $O/p_base -name "*results.db" | head -1 | xargs -I{} python3 tools/rocpd_stats.py {} 60 > $O/base_kernels.txt
rm -rf $O/p_new $O/p_base
echo "== new"; grep -E "$PAT" $O/new_kernels.txt | cut -c1-60,95-160
echo "== base"; grep -E "$PAT" $O/base_kernels.txt | cut -c1-60,95-160

#!/bin/bash
# Same-box per-kernel comparison of several library builds: rocprofv3 kernel statistics of a short fp32 bench run per library.
# usage: tools/ab_kernels_multi.sh <tag> <grep pattern> <name>=<lib path | tree> ...   -> gpurun_out/<tag>/<name>_<pass>_kernels.txt
# ("tree" = the in-tree library).  Runs the libraries in the given order, then once more in reverse (clock drift shows as a spread).
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
T=$1; PAT=$2; shift; shift
O=$R/gpurun_out/$T
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
run() {  # name lib pass
  if [ "$2" = tree ]; then unset SSP_HIP_LIB SSP_SKIP_ISA_VERIFY; else export SSP_HIP_LIB=$R/$2 SSP_SKIP_ISA_VERIFY=1; fi
  rocprofv3 --kernel-trace --stats -d $O/p_$1_$3 -o x -- python3 $R/bench.py --no-cpu-baseline --traffic none --no-roofline --no-export --no-bf16 --no-sp --steps 6 --warmup 1 > $O/$1_$3.json 2>/dev/null
  find $O/p_$1_$3 -name "*results.db" | head -1 | xargs -I{} python3 $R/tools/rocpd_stats.py {} 60 > $O/$1_$3_kernels.txt
  rm -rf $O/p_$1_$3
}
ARGS=("$@")
for a in "${ARGS[@]}"; do run ${a%%=*} ${a##*=} 1; done
for ((i=${#ARGS[@]}-1; i>=0; i--)); do a=${ARGS[$i]}; run ${a%%=*} ${a##*=} 2; done
cd $R
for a in "${ARGS[@]}"; do for p in 1 2; do echo "== ${a%%=*} pass $p"; grep -E "$PAT" $O/${a%%=*}_${p}_kernels.txt | cut -c1-70,95-150; done; done

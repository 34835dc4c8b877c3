#!/bin/bash
# kernel-trace statistics (+ optionally the SQ PMC pass: PMC=1) of the fp32 pair step alone (run through gpurun from the repo root)
# -> gpurun_out/pf32/{kernel_stats,pmc_sq_summary}.txt      usage: tools/prof_f32_quick.sh [bench args, e.g. --arch sp]
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/pf32
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
Q="--no-cpu-baseline --traffic none --no-export --no-roofline --no-bf16 --no-sp $*"
rocprofv3 --kernel-trace --stats -d $O/kt -o k -- python3 $R/bench.py $Q --steps 6 --warmup 1 > $O/bench_line.txt 2>/dev/null
if [ "${PMC:-0}" = "1" ]; then
  rocprofv3 --pmc SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_ANY SQ_WAVE_CYCLES --kernel-trace --output-format csv -d $O/pmc_sq -o s -- python3 $R/bench.py $Q --steps 1 --warmup 1 > /dev/null 2>&1
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -o f -- python3 $R/bench.py $Q --steps 1 --warmup 1 > /dev/null 2>&1
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -o w -- python3 $R/bench.py $Q --steps 1 --warmup 1 > /dev/null 2>&1
fi
cd $R
find $O/kt -name "*results.db" | head -1 | xargs -I{} python3 tools/rocpd_stats.py {} 60 > $O/kernel_stats.txt
if [ "${PMC:-0}" = "1" ]; then
  for k in fetch:f write:w sq:s; do n=${k%%:*}; cc=$(find $O/pmc_$n -name "*counter_collection.csv" | head -1); kt=$(find $O/pmc_$n -name "*kernel_trace.csv" | head -1); python3 tools/pmc_summary.py $cc $kt 40 > $O/pmc_${n}_summary.txt; done
  python3 tools/hbm_table.py $O/pmc_fetch_summary.txt $O/pmc_write_summary.txt 50 > $O/hbm_kernel_table.txt
fi
rm -rf $O/kt $O/pmc_sq $O/pmc_fetch $O/pmc_write

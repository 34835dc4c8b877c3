#!/bin/bash
# PMC + kernel-trace profile of the bf16 path's pair step (run through gpurun from the repo root); outputs in gpurun_out/pbf16/
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/pbf16
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
Q="--conv-algo 12 --no-cpu-baseline --traffic none --no-export --no-roofline"
rocprofv3 --kernel-trace --stats -d $O/kt -o k -- python3 $R/bench.py $Q --steps 6 --warmup 1 > /dev/null 2>&1
rocprofv3 --pmc SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_ANY SQ_WAVE_CYCLES --kernel-trace --output-format csv -d $O/pmc_sq -o s -- python3 $R/bench.py $Q --steps 1 --warmup 1 > /dev/null 2>&1
rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_LDS_IDX_ACTIVE SQ_WAVES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS --kernel-trace --output-format csv -d $O/pmc_sq2 -o s -- python3 $R/bench.py $Q --steps 1 --warmup 1 > /dev/null 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -o f -- python3 $R/bench.py $Q --steps 1 --warmup 1 > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -o w -- python3 $R/bench.py $Q --steps 1 --warmup 1 > /dev/null 2>&1
cd $R
find $O/kt -name "*results.db" | head -1 | xargs -I{} python3 tools/rocpd_stats.py {} 60 > $O/kernel_stats.txt
for k in fetch:f write:w sq:s sq2:s; do n=${k%%:*}; cc=$(find $O/pmc_$n -name "*counter_collection.csv" | head -1); kt=$(find $O/pmc_$n -name "*kernel_trace.csv" | head -1); python3 tools/pmc_summary.py $cc $kt 40 > $O/pmc_${n}_summary.txt; done
python3 tools/hbm_table.py $O/pmc_fetch_summary.txt $O/pmc_write_summary.txt 40 > $O/hbm_kernel_table.txt
rm -rf $O/kt $O/pmc_fetch $O/pmc_write $O/pmc_sq $O/pmc_sq2
ls -la $O

#!/bin/bash
# Regenerates the round's measurement artefacts on a GPU box (run through gpurun from the repo root); the outputs land in
# gpurun_out/refresh/ and are copied into profiles/ by hand after inspection.
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/refresh
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
python3 $R/bench.py > $O/bench_sp.json 2> $O/bench_sp.err
python3 $R/bench.py --arch ssp --no-cpu-baseline > $O/bench_ssp.json 2>/dev/null
python3 $R/bench.py --desc-loss dense --no-cpu-baseline > $O/bench_sp_dense_loss.json 2>/dev/null
python3 $R/bench.py --conv-algo 3 --no-cpu-baseline > $O/bench_sp_bf16.json 2>/dev/null
python3 $R/bench_export.py > $O/bench_export_240x320.json 2>/dev/null
python3 $R/bench_export.py --height 480 --width 640 > $O/bench_export_480x640.json 2>/dev/null
rocprofv3 --kernel-trace --stats -d $O/prof_sp -o sp -- python3 $R/bench.py --no-cpu-baseline --steps 6 --warmup 1 > $O/prof_sp.json 2>/dev/null
rocprofv3 --kernel-trace --stats -d $O/prof_ssp -o ssp -- python3 $R/bench.py --arch ssp --no-cpu-baseline --no-roofline --steps 6 --warmup 1 > /dev/null 2>&1
rocprofv3 --kernel-trace --stats -d $O/prof_export -o ex -- python3 $R/bench_export.py --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -o f -- python3 $R/bench.py --no-cpu-baseline --no-roofline --steps 1 --warmup 1 > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -o w -- python3 $R/bench.py --no-cpu-baseline --no-roofline --steps 1 --warmup 1 > /dev/null 2>&1
rocprofv3 --pmc SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/pmc_sq -o s -- python3 $R/bench.py --no-cpu-baseline --no-roofline --steps 1 --warmup 1 > /dev/null 2>&1
cd $R
python3 tools/rocpd_stats.py $O/prof_sp/sp_results.db 60 > $O/sp_kernel_stats.txt
python3 tools/rocpd_stats.py $O/prof_ssp/ssp_results.db 60 > $O/ssp_kernel_stats.txt
python3 tools/rocpd_stats.py $O/prof_export/ex_results.db 40 > $O/export_kernel_stats.txt
python3 tools/pmc_summary.py $O/pmc_fetch/f_counter_collection.csv $O/pmc_fetch/f_kernel_trace.csv 30 > $O/pmc_fetch_summary.txt
python3 tools/pmc_summary.py $O/pmc_write/w_counter_collection.csv $O/pmc_write/w_kernel_trace.csv 30 > $O/pmc_write_summary.txt
python3 tools/pmc_summary.py $O/pmc_sq/s_counter_collection.csv $O/pmc_sq/s_kernel_trace.csv 30 > $O/pmc_sq_summary.txt
rm -rf $O/prof_sp $O/prof_ssp $O/prof_export $O/pmc_fetch/*.csv $O/pmc_write/*.csv $O/pmc_sq/*.csv
ls -la $O

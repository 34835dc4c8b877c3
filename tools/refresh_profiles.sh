#!/bin/bash
# Regenerates the round's measurement artefacts on a GPU box (run through gpurun from the repo root); the outputs land in
# gpurun_out/refresh/ and are copied into profiles/ (named per round) after inspection.
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/refresh
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
Q="--no-cpu-baseline --traffic none --no-export --no-bf16 --no-sp"   # (--no-bf16 --no-sp: the fp32 passes must not trace the in-process bf16 block)
python3 $R/bench.py --steps 20 --warmup 5 > $O/bench_ssp.json 2> $O/bench_ssp.err
python3 $R/bench.py --arch sp --steps 20 --warmup 5 --no-cpu-baseline --no-export > $O/bench_sp.json 2>/dev/null
python3 $R/bench.py --steps 20 --warmup 5 $Q --no-roofline --graph > $O/bench_ssp_graph.json 2>/dev/null
python3 $R/bench.py --steps 20 --warmup 5 $Q --no-roofline > $O/bench_ssp_eager.json 2>/dev/null
python3 $R/bench.py --conv-algo 0 --steps 10 --warmup 3 $Q > $O/bench_ssp_direct.json 2>/dev/null
python3 $R/bench.py --conv-algo 9 --steps 20 --warmup 5 $Q > $O/bench_ssp_f2x2_only.json 2>/dev/null
python3 $R/bench.py --conv-algo 11 --steps 20 --warmup 5 $Q > $O/bench_ssp_wgrad_f3x3_4x4.json 2>/dev/null
SSP_FUSE_APPLY=0 python3 $R/bench.py --steps 20 --warmup 5 $Q > $O/bench_ssp_no_fused_apply.json 2>/dev/null
SSP_LOSS_STREAM=0 python3 $R/bench.py --steps 20 --warmup 5 $Q > $O/bench_ssp_one_stream.json 2>/dev/null
SSP_G1=0 python3 $R/bench.py --steps 20 --warmup 5 $Q > $O/bench_ssp_no_grouped_pointwise.json 2>/dev/null
python3 $R/bench.py --desc-loss dense --steps 10 --warmup 3 $Q > $O/bench_ssp_dense_loss.json 2>/dev/null
python3 $R/bench_export.py --steps 5 --warmup 2 > $O/bench_export_480x640.json 2>/dev/null
python3 $R/bench_export.py --height 240 --width 320 --steps 10 --warmup 2 --no-cpu-baseline > $O/bench_export_240x320.json 2>/dev/null
rocprofv3 --kernel-trace --stats -d $O/prof_ssp -o ssp -- python3 $R/bench.py $Q --steps 6 --warmup 1 > $O/prof_ssp.json 2>/dev/null
rocprofv3 --kernel-trace --stats -d $O/prof_sp -o sp -- python3 $R/bench.py --arch sp $Q --steps 6 --warmup 1 > $O/prof_sp.json 2>/dev/null
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -o f -- python3 $R/bench.py $Q --no-roofline --steps 1 --warmup 1 > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -o w -- python3 $R/bench.py $Q --no-roofline --steps 1 --warmup 1 > /dev/null 2>&1
rocprofv3 --pmc SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_ANY SQ_WAVE_CYCLES --kernel-trace --output-format csv -d $O/pmc_sq -o s -- python3 $R/bench.py $Q --no-roofline --steps 1 --warmup 1 > /dev/null 2>&1
# ---- round 4: the bf16 path (conv algorithm 12), deterministic mode, power / clock evidence ----
python3 $R/bench.py --dtype bf16 --steps 20 --warmup 5 --no-cpu-baseline --no-export > $O/bench_ssp_bf16.json 2>/dev/null
python3 $R/bench.py --arch sp --dtype bf16 --steps 20 --warmup 5 --no-cpu-baseline --no-export > $O/bench_sp_bf16.json 2>/dev/null
SSP_CONVB_WS=0 python3 $R/bench.py --dtype bf16 --steps 20 --warmup 5 $Q > $O/bench_ssp_bf16_generic_3x3_kernels.json 2>/dev/null
SSP_DETERMINISTIC=1 python3 $R/bench.py --steps 20 --warmup 5 $Q --no-roofline --no-bf16 --no-sp > $O/bench_ssp_deterministic.json 2>/dev/null
SSP_DETERMINISTIC=1 python3 $R/bench.py --dtype bf16 --steps 20 --warmup 5 $Q --no-roofline > $O/bench_ssp_bf16_deterministic.json 2>/dev/null
cd $R
bash tools/prof_bf16.sh > /dev/null 2>&1
for f in kernel_stats pmc_fetch_summary pmc_write_summary pmc_sq_summary pmc_sq2_summary hbm_kernel_table; do cp gpurun_out/pbf16/$f.txt $O/bf16_$f.txt; done
bash tools/dbg/instep_trace.sh; cp gpurun_out/instep_trace.txt $O/bf16_instep_phase_trace.txt
bash tools/dbg/sustained_ablate.sh; cp gpurun_out/sustained_ablate.txt $O/bf16_conv_sustained_ablation_power.txt
bash tools/dbg/power_step.sh bf16 > $O/power_clock_pair_step_bf16.txt 2>&1
bash tools/dbg/power_step.sh f32 > $O/power_clock_pair_step_f32.txt 2>&1
$R/ab/mfma_valu_share > $O/ubench_mfma_valu_share.txt 2>&1
$R/ab/mfma_lds_loop > $O/ubench_mfma_lds_loop.txt 2>&1
find $O/prof_ssp -name "*results.db" | head -1 | xargs -I{} python3 tools/rocpd_stats.py {} 60 > $O/ssp_kernel_stats.txt
find $O/prof_sp -name "*results.db" | head -1 | xargs -I{} python3 tools/rocpd_stats.py {} 60 > $O/sp_kernel_stats.txt
for k in fetch:f write:w sq:s; do n=${k%%:*}; p=${k##*:}; cc=$(find $O/pmc_$n -name "*counter_collection.csv" | head -1); kt=$(find $O/pmc_$n -name "*kernel_trace.csv" | head -1); python3 tools/pmc_summary.py $cc $kt 40 > $O/pmc_${n}_summary.txt; done
python3 tools/hbm_table.py $O/pmc_fetch_summary.txt $O/pmc_write_summary.txt 50 > $O/hbm_kernel_table.txt
rm -rf $O/prof_ssp $O/prof_sp $O/pmc_fetch $O/pmc_write $O/pmc_sq
ls -la $O

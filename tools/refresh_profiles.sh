#!/bin/bash
# Regenerates the round's measurement artefacts on a GPU box (run through gpurun from the repo root); the outputs land in
# gpurun_out/refresh/ and are copied into profiles/ (named per round) after inspection.
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/refresh
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
Q="--no-cpu-baseline --traffic none --no-export --no-bf16 --no-sp"   # (the fp32 passes must not run the in-process side blocks)
# ---- the driver's command, timed; then the single-configuration lines ----
( time python3 $R/bench.py > $O/bench_default_run.json 2> $O/bench_default_run.err ) 2> $O/bench_default_run.time
python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-export --no-bf16 --no-sp > $O/bench_ssp.json 2>/dev/null
python3 $R/bench.py --arch sp --steps 20 --warmup 5 --no-cpu-baseline --no-export --no-bf16 > $O/bench_sp.json 2>/dev/null
python3 $R/bench.py --dtype bf16 --steps 20 --warmup 5 --no-cpu-baseline --no-export > $O/bench_ssp_bf16.json 2>/dev/null
python3 $R/bench.py --arch sp --dtype bf16 --steps 20 --warmup 5 --no-cpu-baseline --no-export > $O/bench_sp_bf16.json 2>/dev/null
python3 $R/bench.py --steps 20 --warmup 5 $Q --no-roofline --graph > $O/bench_ssp_graph.json 2>/dev/null
python3 $R/bench.py --steps 20 --warmup 5 $Q --no-roofline > $O/bench_ssp_eager.json 2>/dev/null
python3 $R/bench.py --conv-algo 0 --steps 10 --warmup 3 $Q > $O/bench_ssp_direct.json 2>/dev/null
python3 $R/bench.py --conv-algo 9 --steps 20 --warmup 5 $Q > $O/bench_ssp_f2x2_only.json 2>/dev/null
python3 $R/bench.py --desc-loss dense --steps 10 --warmup 3 $Q > $O/bench_ssp_dense_loss.json 2>/dev/null
SSP_DETERMINISTIC=1 python3 $R/bench.py --steps 20 --warmup 5 $Q --no-roofline > $O/bench_ssp_deterministic.json 2>/dev/null
SSP_DETERMINISTIC=1 python3 $R/bench.py --dtype bf16 --steps 20 --warmup 5 $Q --no-roofline > $O/bench_ssp_bf16_deterministic.json 2>/dev/null
SSP_BF16_BNR=0 python3 $R/bench.py --dtype bf16 --steps 20 --warmup 5 $Q --no-roofline > $O/bench_ssp_bf16_separate_bn_sums.json 2>/dev/null
python3 $R/bench_export.py --steps 5 --warmup 2 > $O/bench_export_480x640.json 2>/dev/null
python3 $R/bench_export.py --height 240 --width 320 --steps 10 --warmup 2 --no-cpu-baseline > $O/bench_export_240x320.json 2>/dev/null
cd $R
# ---- kernel statistics + PMC passes: fp32 SSp / SP (the step alone), bf16 ----
PMC=1 bash tools/prof_f32_quick.sh
for f in kernel_stats pmc_fetch_summary pmc_write_summary pmc_sq_summary hbm_kernel_table; do cp gpurun_out/pf32/$f.txt $O/ssp_$f.txt; done
bash tools/prof_f32_quick.sh --arch sp
cp gpurun_out/pf32/kernel_stats.txt $O/sp_kernel_stats.txt
bash tools/prof_bf16.sh > /dev/null 2>&1
for f in kernel_stats pmc_fetch_summary pmc_write_summary pmc_sq_summary pmc_sq2_summary hbm_kernel_table; do cp gpurun_out/pbf16/$f.txt $O/bf16_$f.txt || echo "refresh_profiles: MISSING gpurun_out/pbf16/$f.txt (prof_bf16.sh failed?)" >&2; done
# ---- dispatch timelines of one step, in-step phase trace of the bf16 3x3 kernel ----
bash tools/timeline_step.sh
cp gpurun_out/timeline/f32.txt $O/step_timeline_f32.txt; cp gpurun_out/timeline/bf16.txt $O/step_timeline_bf16.txt
bash tools/dbg/instep_trace.sh; cp gpurun_out/instep_trace.txt $O/bf16_instep_phase_trace.txt
# ---- every artefact the round copies into profiles/ must exist and be non-empty: a failed pass must not leave stale files unnoticed ----
rc=0
for f in bench_default_run.json bench_ssp.json bench_sp.json bench_ssp_bf16.json bench_sp_bf16.json bench_ssp_graph.json bench_ssp_eager.json \
         bench_ssp_direct.json bench_ssp_f2x2_only.json bench_ssp_dense_loss.json bench_ssp_deterministic.json bench_ssp_bf16_deterministic.json \
         bench_ssp_bf16_separate_bn_sums.json bench_export_480x640.json bench_export_240x320.json ssp_kernel_stats.txt ssp_pmc_fetch_summary.txt \
         ssp_pmc_write_summary.txt ssp_pmc_sq_summary.txt ssp_hbm_kernel_table.txt sp_kernel_stats.txt bf16_kernel_stats.txt bf16_pmc_fetch_summary.txt \
         bf16_pmc_write_summary.txt bf16_pmc_sq_summary.txt bf16_pmc_sq2_summary.txt bf16_hbm_kernel_table.txt step_timeline_f32.txt \
         step_timeline_bf16.txt bf16_instep_phase_trace.txt; do
  if [ ! -s $O/$f ]; then echo "refresh_profiles: MISSING or empty: $O/$f" >&2; rc=1; fi
done
ls -la $O
exit $rc

# perf-debug: conv_bf16_ws_kernel phase ablations, kernel durations from rocprofv3 (run through gpurun)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/wsab; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for a in ${ABLATES:-0 7 8 15}; do
  SSP_CONVB_ABLATE=$a rocprofv3 --kernel-trace --stats -d $O/p$a -o x -- python3 $R/tools/dbg/convb_time.py conv > $O/run$a.txt 2>&1
  db=$(find $O/p$a -name "*results.db" | head -1)
  echo "== ablate $a"; python3 $R/tools/rocpd_stats.py $db 6
  rm -rf $O/p$a
done > $O/summary.txt 2>&1

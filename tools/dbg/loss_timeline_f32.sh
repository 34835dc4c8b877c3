# perf-debug: timeline of the loss phase inside the fp32 pair step under values of an environment variable (run through gpurun)
# usage: tools/dbg/loss_timeline_f32.sh <VAR> "<v1> <v2> ..."      -> gpurun_out/losstl/timeline_f32_<VAR>_<v>.txt
R=${GRAFT_REPO_ROOT:-/root/repo}
V=${1:-SSP_SEM_XC}; VALS=${2:-"1 0"}
O=$R/gpurun_out/losstl; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for x in $VALS; do
  export $V=$x
  rocprofv3 --kernel-trace --output-format csv -d $O/kt$x -o k -- python3 $R/bench.py --no-cpu-baseline --traffic none --no-export --no-roofline --no-bf16 --no-sp --steps 10 --warmup 2 > /dev/null 2>&1
  python3 $R/tools/loss_phase_timeline.py $O/kt$x > $O/timeline_f32_${V}_$x.txt 2>&1
  rm -rf $O/kt$x
done

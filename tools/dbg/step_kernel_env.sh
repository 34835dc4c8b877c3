# perf-debug: kernel trace of the fp32 step under two values of an environment variable (A = first value, B = second), per-kernel
# differences (run through gpurun).  usage: tools/dbg/step_kernel_env.sh <VAR> <a> <b> [bench args]
R=${GRAFT_REPO_ROOT:-/root/repo}
V=$1; A=$2; B=$3; shift; shift; shift
O=$R/gpurun_out/stepenv; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
Q="--no-cpu-baseline --traffic none --no-export --no-roofline --no-bf16 --no-sp --steps 12 --warmup 2 $*"
for r in 1 2; do
export $V=$A; rocprofv3 --kernel-trace --output-format csv -d $O/a -o k -- python3 $R/bench.py $Q > /dev/null 2>&1
export $V=$B; rocprofv3 --kernel-trace --output-format csv -d $O/b -o k -- python3 $R/bench.py $Q > /dev/null 2>&1
python3 $R/tools/dbg/step_kernel_diff.py $O/a $O/b > $O/diff_${V}_$r.txt 2>&1
rm -rf $O/a $O/b
done

# perf-debug: sustained (4 s) operator loops under the phase ablations: ms per call, package power, shader clock  (run through gpurun)
for w in dgrad conv; do for cfg in "0 1" "8 1" "4 1" "1 1" "12 1" "0 0"; do set -- $cfg
  echo "== $w ablate $1 ws $2"; SSP_CONVB_ABLATE=$1 SSP_CONVB_WS=$2 bash tools/dbg/power_watch.sh $w 2>&1 | grep -v amdgpu.ids
done; done > gpurun_out/sustained_ablate.txt 2>&1

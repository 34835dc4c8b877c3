# perf-debug: package power / shader clock sampled during a long run of the pair step  (run through gpurun; $1 = f32 | bf16)
D=${1:-bf16}
python bench.py --dtype $D --no-cpu-baseline --traffic none --no-export --no-roofline --no-bf16 --no-sp --steps 2500 --warmup 10 > gpurun_out/power_step_$D.json 2>/dev/null &
PID=$!
for i in $(seq 1 90); do rocm-smi --showpower --showclocks 2>&1 | grep -i "Power (W)\|sclk" | sed 's/GPU\[0\]\t\t: //' | tr '\n' ' '; echo; sleep 0.4; kill -0 $PID 2>/dev/null || break; done | awk '{ if ($NF + 0 > 300) print }' | tail -25
wait $PID
python -c "import json; d = json.loads(open('gpurun_out/power_step_$D.json').read().strip().splitlines()[-1]); print('$D: %.1f pairs/s %.3f ms/step' % (d['value'], d['ms_per_step']))"

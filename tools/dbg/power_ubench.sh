# perf-debug: socket power / shader clock of the MFMA loop with register operands (0) and with its LDS operand reads (2), sustained
for v in 0 2; do
  ab/mfma_lds_loop $v 4 &
  PID=$!
  sleep 1.5
  for i in 1 2 3 4 5; do rocm-smi --showpower --showclocks 2>&1 | grep -i "Power (W)\|sclk" | sed 's/GPU\[0\]\t\t: //' | tr '\n' ' '; echo; sleep 0.4; done
  wait $PID
done

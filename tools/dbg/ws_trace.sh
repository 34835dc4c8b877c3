# perf-debug: per-phase cycle sums of conv_bf16_ws_kernel (workgroup 0); SSP_CONVB_TRACE=N traces every N-th launch (the others run
# back to back: the shader clock of a sustained run differs from that of an isolated launch)  (run through gpurun)
for a in ${ABLATES:-0}; do echo "== ablate $a"; SSP_CONVB_TRACE=${TRACE_EVERY:-13} SSP_CONVB_ABLATE=$a timeout 120 python tools/dbg/convb_time.py conv 2>&1 | grep -A20 "convb trace" | head -${LINES_:-22}; done > gpurun_out/ws_trace.txt 2>&1

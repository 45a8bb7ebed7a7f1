#!/bin/bash
# tools/power_trace.sh -- sample rocm-smi (socket power, sclk) while the headline demodulator runs back to back, to tell whether
# the kernel is clock-throttled (power-limited) on the box it is measured on.  Output: gpurun_out/power_trace.txt
mkdir -p gpurun_out
OUT=gpurun_out/power_trace.txt
{ echo "# rocm-smi idle"; rocm-smi --showpower --showclocks --showmaxpower 2>&1 | grep -v "^=\|^$" | head -30; } > $OUT
python tools/power_loop.py 20 > gpurun_out/power_loop.txt 2>&1 &
BPID=$!
for i in $(seq 1 120); do grep -q "loop start" gpurun_out/power_loop.txt 2>/dev/null && break; sleep 1; done
for i in $(seq 1 12); do
  kill -0 $BPID 2>/dev/null || break
  echo "# t=$i" >> $OUT
  rocm-smi --showpower --showclocks 2>&1 | grep -i "power\|sclk\|mclk" >> $OUT
  sleep 1
done
wait $BPID
cat gpurun_out/power_loop.txt >> $OUT

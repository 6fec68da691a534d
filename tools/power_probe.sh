#!/bin/bash
# Board power / sclk (rocm-smi, once a second) while tools/wave_clock loops ONE configuration of the
# N = 2048 kernel for 10 s: random data from HBM, all-zero data, random data L2-resident.
# Is the chip at its power cap, and which of the three is?    bash tools/power_probe.sh > gpurun_out/power.txt
cd "$(dirname "$0")/.."
rocm-smi --showmaxpower 2>/dev/null | grep -i "power" | head -2
for cfg in 0 1 3; do
  echo "=== wave_clock config $cfg"
  tools/wave_clock 2048 3353 $cfg 10 &
  PID=$!
  sleep 3
  for k in 1 2 3 4 5; do
    rocm-smi --showpower --showclocks 2>/dev/null | grep -E "Power|sclk" | tr -s ' ' | tr '\n' ' '
    echo
    sleep 1
  done
  wait $PID
done

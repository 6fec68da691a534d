#!/bin/bash
# Sample GPU power / clocks while bench.py loops (run inside gpurun):
#   bash tools/power_watch.sh [bench args]   -> gpurun_out/power_watch.txt
# Shows whether the feature kernel runs at the power cap (sclk below its maximum).
OUT=gpurun_out/power_watch.txt
mkdir -p gpurun_out
rocm-smi --showmaxpower --showclocks --showpower > $OUT 2>&1
python3 bench.py --no-cpu-baseline --steps 3000 --warmup 3 "$@" > gpurun_out/power_watch_bench.json 2> gpurun_out/power_watch_bench.err &
BPID=$!
i=0
while kill -0 $BPID 2> /dev/null && [ $i -lt 60 ]; do      # one sample a second while the bench lives
  i=$((i + 1))
  echo "--- sample $i: $(rocm-smi --showpower --showclocks --showuse 2>&1 | grep -E "Package Power|sclk|busy" | sed 's/.*: //' | tr '\n' ' ')" >> $OUT
  sleep 1
done
wait $BPID
echo "bench rc=$?" >> $OUT
cat gpurun_out/power_watch_bench.json >> $OUT

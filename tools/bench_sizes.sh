#!/bin/bash
# bench.py at every wave size (kernel-only legs), one JSON line each: bash tools/bench_sizes.sh OUT.jsonl [extra bench args]
OUT=${1:-gpurun_out/bench_all_sizes.jsonl}; shift
: > $OUT
for N in 128 256 512 1024 2048 4096 8192 16384 32768; do
  FR=4096; [ $N -le 512 ] && FR=16384; [ $N -eq 16384 ] && FR=512; [ $N -eq 32768 ] && FR=256
  timeout -k 10 200 python3 bench.py --no-cpu-baseline --no-h2d --frame-size $N --frames $FR --warmup 300 --steps 50 "$@" >> $OUT 2>> ${OUT%.jsonl}.err || exit 1
done
python3 - $OUT <<'PY'
import json, sys
for line in open(sys.argv[1]):
    d = json.loads(line); r = d["roofline"]
    print(f'N={d["config"]["frame_size"]:5d} {d["value"]/1e6:8.1f} M frames/s  step {d["ms_per_step"]:.4f} ms  launch min/med/max {r["launch_ms_min"]:.4f}/{r["launch_ms_median"]:.4f}/{r["launch_ms_max"]:.4f}  frac {r["frac"]:.3f}  with D2H {d["wall_incl_d2h_ms"]:.4f} ms ({d["wall_incl_d2h_ms"]/d["ms_per_step"]:.3f}x)')
PY

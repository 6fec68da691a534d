#!/bin/bash
# one kernel-only bench line at frame size $1: bash tools/bench_n.sh 8192 [extra args]
N=$1; shift
timeout -k 10 150 python3 bench.py --no-cpu-baseline --no-h2d --frame-size $N --warmup 100 --steps 30 "$@" 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print(f\"N={d['config']['frame_size']} {d['value']/1e6:.2f} M frames/s step {d['ms_per_step']:.3f} ms min/med/max {r['launch_ms_min']:.3f}/{r['launch_ms_median']:.3f}/{r['launch_ms_max']:.3f} frac {r['frac']:.3f} {d['config']['kernel']}\")"

#!/usr/bin/env python3
"""Do bench.py's HIP-event durations agree with rocprofv3's kernel trace of the SAME run?
    rocprofv3 --kernel-trace --output-format csv -d DIR -- python3 bench.py --no-cpu-baseline --no-h2d --steps K --warmup W > line.json
    python tools/events_vs_rocprof.py DIR line.json
Takes the K timed steps (dispatches W .. W+K-1 of each amcx kernel; the later launches belong to the
D2H-inclusive loop and the read probe) and compares the sum of the per-step kernel means with
`roofline.mean_launch_ms` of the line."""
import csv
import glob
import json
import os
import sys

d, line = sys.argv[1], json.load(open(sys.argv[2]))
W, K = line["warmup"], line["steps"]
rows = []
for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
    rows += list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
per = {}
for r in rows:
    n = r["Kernel_Name"]
    if "amcx" in n and "probe" not in n:
        per.setdefault(n, []).append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
out = {"steps": K, "warmup": W, "kernels": {}}
total = 0.0
for n, v in per.items():
    timed = v[W:W + K]
    m = sum(timed) / len(timed) / 1e6
    out["kernels"][n.split("(")[0]] = {"timed_mean_ms": m, "timed_min_ms": min(timed) / 1e6, "timed_max_ms": max(timed) / 1e6,
                                        "all_dispatches": len(v), "all_mean_ms": sum(v) / len(v) / 1e6}
    total += m
out["rocprof_sum_of_kernel_means_ms"] = total
out["bench_mean_launch_ms_from_hip_events"] = line["roofline"]["mean_launch_ms"]
out["bench_ms_per_step_wall"] = line["ms_per_step"]
out["difference_ms"] = line["roofline"]["mean_launch_ms"] - total
print(json.dumps(out, indent=1))

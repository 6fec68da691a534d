#!/bin/bash
# rocprofv3 kernel trace of the consumer kernels (tools/bench_post.py); per-kernel, per-grid durations to stdout
OUT=gpurun_out/prof_post
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 tools/bench_post.py > $OUT/bench_post.json 2> $OUT/trace.err || exit 1
cat $OUT/bench_post.json
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
f = sorted(glob.glob(sys.argv[1] + "/trace/**/*kernel_trace.csv", recursive=True))[-1]
d = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    if r["Kernel_Name"].startswith("amcx"):
        d[(r["Kernel_Name"].split("(")[0], int(r["Grid_Size_X"]) // int(r["Workgroup_Size_X"]))].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
for (k, g), v in sorted(d.items()):
    v.sort()
    print(f"{k:42s} workgroups {g:6d}  launches {len(v):3d}  min {v[0]/1e3:7.2f} us  median {v[len(v)//2]/1e3:7.2f} us")
PY
find $OUT -name "*.db" -delete; find $OUT -name "*agent_info.csv" -delete

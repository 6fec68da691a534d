#!/bin/bash
# Instruction-cache counters of the throughput kernels (rocprofv3 --pmc SQC_ICACHE_*), one pass per frame size.
# usage (inside gpurun): bash tools/icache_probe.sh TAG [sizes...]      -> gpurun_out/icache_$TAG/summary.txt
TAG=${1:-x}; shift
SIZES=${@:-"1024 2048 4096 8192"}
OUT=gpurun_out/icache_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
for N in $SIZES; do
  BENCH="python3 bench.py --no-cpu-baseline --no-h2d --no-d2h --no-fma-probe --steps 10 --warmup 3 --frame-size $N"
  rocprofv3 --kernel-trace --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_VALU GRBM_GUI_ACTIVE \
      --output-format csv -d $OUT/n$N -- $BENCH > /dev/null 2> $OUT/n$N.err || { tail -5 $OUT/n$N.err; exit 1; }
done
python3 - $OUT <<'PY' | tee $OUT/summary.txt
import csv, glob, sys, collections
out = sys.argv[1]
for d in sorted(glob.glob(out + "/n*/")):
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
    seen = set()
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0]
            if "features18" not in k: continue
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
            key = (k, r["Dispatch_Id"])
            if key not in seen: seen.add(key); cnt[k] += 1
    for k, v in acc.items():
        n = cnt[k]
        req, hit, miss = v["SQC_ICACHE_REQ"] / n, v["SQC_ICACHE_HITS"] / n, v["SQC_ICACHE_MISSES"] / n
        print(f"{d.rstrip('/').split('/')[-1]:>7} {k[-45:]:45} dispatches {n:3d}  ICACHE req {req:.4g} hits {hit:.4g} misses {miss:.4g} "
              f"(dup {v['SQC_ICACHE_MISSES_DUPLICATE'] / n:.4g})  miss/req {miss / max(req, 1):.4f}  "
              f"wait_inst/wave_cycles {v['SQ_WAIT_INST_ANY'] / max(v['SQ_WAVE_CYCLES'], 1):.3f}  VALU {v['SQ_INSTS_VALU'] / n:.4g}")
PY
find $OUT -name "*.db" -delete; find $OUT -name "*counter_collection.csv" -delete; find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*agent_info.csv" -delete

#!/bin/bash
# counters of a frame size's throughput kernel (NN=128 by default): VALU / LDS / VMEM instructions per frame, wait shares per wave-cycle
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
for tag in ${TAGS:-product}; do
  : # (while both N = 128 kernels existed, AMCX_SHORT=1 selected the new one: profiles/r5_short_kernel_ab.txt)
  OUT=gpurun_out/pmc_n${NN:-128}_$tag
  mkdir -p $OUT
  rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE --output-format csv -d $OUT/a -- python3 bench.py --no-cpu-baseline --no-h2d --no-d2h --no-fma-probe --steps 20 --warmup 5 --frame-size ${NN:-128} > $OUT/a.json 2> $OUT/a.err
  rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY --output-format csv -d $OUT/b -- python3 bench.py --no-cpu-baseline --no-h2d --no-d2h --no-fma-probe --steps 20 --warmup 5 --frame-size ${NN:-128} > /dev/null 2> $OUT/b.err
  python3 - $OUT $tag <<'PY'
import csv, glob, sys, collections
acc = collections.defaultdict(float); n=collections.defaultdict(set)
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "features18" not in r["Kernel_Name"]: continue
        acc[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]].add(r["Dispatch_Id"]); name=r["Kernel_Name"][:70]
m={k: acc[k]/len(n[k]) for k in acc}
gui=m["GRBM_GUI_ACTIVE"]/8
wc=m["SQ_WAVE_CYCLES"]
frames=6*26*4096
print(sys.argv[2], name)
print("  cycles(GUI)/dispatch %.3g  waves %d  waves/SIMD %.2f  VALU/frame %.1f  cyc/instr/SIMD %.2f" % (gui, m["SQ_WAVES"], wc*4/(1024*gui), m["SQ_INSTS_VALU"]/frames, gui*1024/m["SQ_INSTS_VALU"]))
print("  per wave-cycle: active_valu %.3f wait_inst %.3f wait_any %.3f active_lds %.3f wait_lds %.3f  bank_conflict/idx_active %.3f  LDS instr/frame %.1f  VMEM_RD/frame %.2f SALU/frame %.1f" % (
   m["SQ_ACTIVE_INST_VALU"]/wc, m["SQ_WAIT_INST_ANY"]/wc, m["SQ_WAIT_ANY"]/wc, m["SQ_ACTIVE_INST_LDS"]/wc, m["SQ_WAIT_INST_LDS"]/wc, m["SQ_LDS_BANK_CONFLICT"]/max(m["SQ_LDS_IDX_ACTIVE"],1), m["SQ_INSTS_LDS"]/frames, m["SQ_INSTS_VMEM_RD"]/frames, m["SQ_INSTS_SALU"]/frames))
PY
  find $OUT -name "*.db" -delete; find $OUT -name "*.csv" -delete
done

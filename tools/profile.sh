#!/bin/bash
# rocprofv3 passes over bench.py on the GPU box; summaries land in gpurun_out/prof_$TAG/.
# usage (inside gpurun): bash tools/profile.sh TAG [bench args...]
# Separate passes: kernel-trace+stats, then PMC groups (never combined with trace domains
# other than kernel-trace).
TAG=${1:-r1}; shift
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
# 100 timed launches behind 5 warm-up ones: rocprofv3's plain --stats average (which cannot tell them apart) is then within
# 0.2 % of the mean over the timed launches that bench.py's events bracket (prof_summary.py reports that mean as well)
export AMCX_PROFILE_TIMED=100
BENCH="python3 bench.py --no-cpu-baseline --no-h2d --no-d2h --no-fma-probe --no-other-configs --steps 100 --warmup 5 $@"
set -x
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- $BENCH > $OUT/bench_trace.json 2> $OUT/trace.err || exit 1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- $BENCH > /dev/null 2> $OUT/pmc_fetch.err || exit 1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- $BENCH > /dev/null 2> $OUT/pmc_write.err || exit 1
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $OUT/pmc_sq1 -- $BENCH > /dev/null 2> $OUT/pmc_sq1.err || exit 1
rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_sq2 -- $BENCH > /dev/null 2> $OUT/pmc_sq2.err || exit 1
set +x
python3 tools/prof_summary.py $OUT > $OUT/summary.txt 2>&1
cat $OUT/summary.txt
# keep only the condensed results: summary.txt / pmc_traffic.json / the --stats table (gpurun merges <= 64 MiB back)
find $OUT -name "*.db" -delete
find $OUT -name "*counter_collection.csv" -delete
find $OUT -name "*kernel_trace.csv" -delete
find $OUT -name "*agent_info.csv" -delete
du -sh $OUT

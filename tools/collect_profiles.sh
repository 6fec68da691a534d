#!/bin/bash
# Copy the condensed results of `bash tools/profile_all.sh TAG` (gpurun_out/, scratch) into profiles/ (tracked).
#   bash tools/collect_profiles.sh r3
TAG=${1:-r4}
cd "$(dirname "$0")/.."
for N in 2048 4096 1024 8192 16384 32768; do
  D=gpurun_out/prof_${TAG}_n$N
  [ -d $D ] || { echo "no $D"; continue; }
  cp $D/summary.txt profiles/${TAG}_n${N}_summary.json
  cp $D/pmc_traffic.json profiles/${TAG}_n${N}_pmc_traffic.json
  cp $D/bench_trace.json profiles/${TAG}_bench_under_rocprof_n$N.json
  cp "$(ls -t $(find $D/trace -name '*kernel_stats.csv') | head -1)" profiles/${TAG}_n${N}_kernel_stats.csv
done
for f in bench_all_sizes.jsonl bench_default.json wave_clock.txt; do
  [ -f gpurun_out/${TAG}_$f ] && cp gpurun_out/${TAG}_$f profiles/${TAG}_$f
done
for f in bench_driver_flags.json bench_all_sizes.txt bench_block_sizes.txt bench_block_sizes.jsonl; do
  [ -f gpurun_out/${TAG}_$f ] && cp gpurun_out/${TAG}_$f profiles/${TAG}_$f
done
# the N = 2048 kernel's instruction budget, bound to the digests of the binary the counters were taken on (bench.py replays
# it into roofline.secondary only while the running kernel's machine code is the same)
[ -f profiles/${TAG}_n2048_summary.json ] && [ -f profiles/${TAG}_wave_clock.txt ] && \
  python3 tools/make_wave_budget.py profiles/${TAG}_n2048_summary.json profiles/${TAG}_wave_clock.txt > profiles/${TAG}_wave_budget.json
ls -la profiles/${TAG}_n2048_* profiles/${TAG}_bench_default.json

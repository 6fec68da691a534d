#!/bin/bash
# The round's committed evidence in one gpurun call: rocprofv3 passes at the BASELINE frame sizes (+ 8192), the
# default bench line with both CPU baselines, every size un-profiled, the in-kernel clock.
#   bash tools/profile_all.sh r3     -> gpurun_out/prof_r3_n*/, gpurun_out/r3_*.json*
TAG=${1:-r4}
for N in 2048 4096 1024 8192 16384 32768; do
  FR=4096; [ $N -eq 16384 ] && FR=512; [ $N -eq 32768 ] && FR=256
  export AMCX_PROFILE_FRAMES=$((6 * 26 * FR))
  bash tools/profile.sh ${TAG}_n$N --frame-size $N --frames $FR > gpurun_out/${TAG}_n${N}_profile.log 2>&1 || { echo "profile N=$N failed"; tail -5 gpurun_out/${TAG}_n${N}_profile.log; exit 1; }
  echo "profiled N=$N: $(python3 -c "
import json; d=json.load(open('gpurun_out/prof_${TAG}_n$N/summary.txt')); b=json.load(open('gpurun_out/prof_${TAG}_n$N/bench_trace.json'))
k=[v for n,v in d['dispatch_ns'].items() if 'features18' in n][0]; r=([v for n,v in d['dispatch_ns'].items() if 'range' in n] or [{'mean_of_the_timed_launches': 0.0}])[0]
print(f\"trace: feature kernel {k['mean_of_the_timed_launches']/1e3:.1f} us + range fix-up {r['mean_of_the_timed_launches']/1e3:.1f} us over the timed launches; events {b['roofline']['mean_launch_ms']*1e3:.1f} us; step {b['ms_per_step']*1e3:.1f} us\")")"
done
bash tools/bench_sizes.sh gpurun_out/${TAG}_bench_all_sizes.jsonl > gpurun_out/${TAG}_bench_all_sizes.txt 2>&1 || exit 1
cat gpurun_out/${TAG}_bench_all_sizes.txt
# the accuracy-first block kernel at two sizes that are not powers of two (Bluestein in LDS; in registers): its cost, for the record
: > gpurun_out/${TAG}_bench_block_sizes.jsonl
for N in 1000 5000; do
  timeout -k 10 200 python3 bench.py --no-cpu-baseline --no-h2d --no-d2h --no-fma-probe --frame-size $N --frames 1024 --warmup 20 --steps 20 >> gpurun_out/${TAG}_bench_block_sizes.jsonl 2>> gpurun_out/${TAG}_bench_block_sizes.err || exit 1
done
python3 -c "
import json
for line in open('gpurun_out/${TAG}_bench_block_sizes.jsonl'):
    d = json.loads(line); r = d['roofline']
    print(f'N={d[\"config\"][\"frame_size\"]:5d} {d[\"config\"][\"kernel\"]:34s} {d[\"value\"]/1e6:8.2f} M frames/s  frac {r[\"frac\"]:.4f}')" | tee gpurun_out/${TAG}_bench_block_sizes.txt
timeout -k 10 300 python3 bench.py --steps 20 --warmup 5 > gpurun_out/${TAG}_bench_driver_flags.json 2> gpurun_out/${TAG}_bench_driver_flags.err || { tail -5 gpurun_out/${TAG}_bench_driver_flags.err; exit 1; }
timeout -k 10 400 python3 bench.py > gpurun_out/${TAG}_bench_default.json 2> gpurun_out/${TAG}_bench_default.err || { tail -5 gpurun_out/${TAG}_bench_default.err; exit 1; }
python3 -c "
import json; d=json.load(open('gpurun_out/${TAG}_bench_default.json')); r=d['roofline']
print('default bench:', d['value'], d['ms_per_step'], r['frac'], r['launch_ms_min'], r['launch_ms_median'], r['launch_ms_max'], 'd2h', d['wall_incl_d2h_ms'])
print('h2d', {k: (v if not isinstance(v, dict) else {kk: vv for kk, vv in v.items() if kk != 'what'}) for k, v in d['h2d'].items() if k != 'what'})
print('cpu', d['cpu_baseline']['value'], d['cpu_baseline']['cores'], d['cpu_baseline_reference_shaped']['value'])"
for N in 2048 4096 1024; do timeout -k 5 60 ./tools/wave_clock $N 0 0 2.5; timeout -k 5 60 ./tools/wave_clock $N 0 1 2.5; done > gpurun_out/${TAG}_wave_clock.txt 2>&1
cat gpurun_out/${TAG}_wave_clock.txt

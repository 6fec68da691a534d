#!/bin/bash
# Same-box A/B of the whole library through bench.py: the product build against one built with AMCX_EXTRA_FLAGS
# (default -DAMCX_EXP_WAVES12), alternating; prints frames/s, step, and the step with the result copied to the host.
#   bash tools/ab_bench_d2h.sh ["-DAMCX_EXP_WAVES12"] [rounds=2]
# The experiment lives in its own file (amcpy_amd/lib/libamcx_exp.so, chosen with AMCX_LIB): libamcx.so stays the product.
FLAGS=${1:--DAMCX_EXP_WAVES12}; ROUNDS=${2:-2}
cd "$(dirname "$0")/.."
EXP=$PWD/amcpy_amd/lib/libamcx_exp.so
python3 amcpy_amd/csrc/build.py > /dev/null 2>&1 || exit 1
AMCX_EXTRA_FLAGS="$FLAGS" python3 amcpy_amd/csrc/build.py --output "$EXP" > /dev/null 2>&1 || exit 1
one() {
  python3 bench.py --no-cpu-baseline --no-h2d --steps 100 --warmup 300 2>/dev/null | python3 -c "
import json, sys
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); r = d['roofline']
print('  %.1f M frames/s  step %.4f ms  launch min/med/max %.4f/%.4f/%.4f  with D2H %.4f ms (%.3fx)' % (d['value'] / 1e6, d['ms_per_step'], r['launch_ms_min'], r['launch_ms_median'], r['launch_ms_max'], d['wall_incl_d2h_ms'], d['wall_incl_d2h_ms'] / d['ms_per_step']))"
}
for r in $(seq 1 $ROUNDS); do
  echo "## product build"; one
  echo "## built with $FLAGS"; AMCX_LIB="$EXP" one
done

#!/bin/bash
# Same-box A/B of library builds with tools/ab_lib_timing.py: the product against AMCX_EXTRA_FLAGS builds.
#   bash tools/ab_lib.sh "-DAMCX_EXP_WAVES12" [rounds=2]
FLAGS=${1:--DAMCX_EXP_WAVES12}; ROUNDS=${2:-2}
cd "$(dirname "$0")/.."
for r in $(seq 1 $ROUNDS); do
  echo "## product build"; python3 amcpy_amd/csrc/build.py --force > /dev/null 2>&1 || exit 1; python3 tools/ab_lib_timing.py 2>/dev/null
  echo "## built with $FLAGS"; AMCX_EXTRA_FLAGS="$FLAGS" python3 amcpy_amd/csrc/build.py --force > /dev/null 2>&1 || exit 1; python3 tools/ab_lib_timing.py 2>/dev/null
done
python3 amcpy_amd/csrc/build.py --force > /dev/null 2>&1

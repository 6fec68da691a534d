#!/bin/bash
# Same-box A/B of library builds with tools/ab_lib_timing.py: the product against an AMCX_EXTRA_FLAGS build.
#   bash tools/ab_lib.sh "-DAMCX_EXP_WAVES12" [rounds=2] [frame size=2048]
# The experiment is built ONCE into its own file (amcpy_amd/lib/libamcx_exp.so) and selected with AMCX_LIB: the
# product library is never replaced, whatever interrupts this script.
FLAGS=${1:--DAMCX_EXP_WAVES12}; ROUNDS=${2:-2}; FS=${3:-2048}
cd "$(dirname "$0")/.."
EXP=$PWD/amcpy_amd/lib/libamcx_exp.so
python3 amcpy_amd/csrc/build.py > /dev/null 2>&1 || exit 1
AMCX_EXTRA_FLAGS="$FLAGS" python3 amcpy_amd/csrc/build.py --output "$EXP" > /dev/null 2>&1 || exit 1
for r in $(seq 1 $ROUNDS); do
  echo "## product build"; python3 tools/ab_lib_timing.py $FS 2>/dev/null
  echo "## built with $FLAGS"; AMCX_LIB="$EXP" python3 tools/ab_lib_timing.py $FS 2>/dev/null
done

#!/bin/bash
# Same-box A/B of two or more prebuilt libraries through tools/ab_lib_timing.py (features18 on a resident arena):
#   bash tools/ab_libs.sh [-n frame_size] [-r rounds] libA.so libB.so ...
# (e.g. the previous commit's library built aside against the tree's: profiles/r4_pool_ab.txt)
N=2048; ROUNDS=2
while getopts "n:r:" o; do case $o in n) N=$OPTARG;; r) ROUNDS=$OPTARG;; esac; done
shift $((OPTIND - 1))
cd "$(dirname "$0")/.."
for r in $(seq 1 $ROUNDS); do
  for lib in "$@"; do
    echo "## $lib"; AMCX_LIB="$(realpath $lib)" python3 tools/ab_lib_timing.py $N 2>/dev/null
  done
done

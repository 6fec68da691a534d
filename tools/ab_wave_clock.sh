#!/bin/bash
# Same-box A/B of wave_clock builds: alternate them ROUNDS times on one configuration.
#   bash tools/ab_wave_clock.sh [-n frame_size] [-c config] [-r rounds] [-s seconds] build1 build2 ...
N=2048; CFG=0; ROUNDS=2; SEC=4
while getopts "n:c:r:s:" o; do case $o in n) N=$OPTARG;; c) CFG=$OPTARG;; r) ROUNDS=$OPTARG;; s) SEC=$OPTARG;; esac; done
shift $((OPTIND - 1))
for r in $(seq 1 $ROUNDS); do
  for bin in "$@"; do
    echo "## $bin"
    $bin $N 0 $CFG $SEC || exit 1
  done
done

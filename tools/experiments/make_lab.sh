#!/bin/bash
# The laboratory: the product sources of commit LAB_BASE with every experiment / ablation branch of rounds 1-4 put back
# (tools/experiments/r5_lab_branches.patch: -DAMCX_EXP_WAVES12, _PK_FFT / _PK_PASS1_ONLY / _PK_TAIL_ONLY, _INTERLEAVE,
# _LOAD_POLICY, _PLAIN_LOADS, _A_IN_REGS, _PAIR4096, -DAMCX_ABL_NOFFT / _NOSTATS / _NOSQRT / _NORCP / _NOTIEFIX /
# _FFT_TAIL / _FFT_TAIL_MFMA, AMCX_PRIO_LEVELS) and the two experiment kernels beside them, as a tree of its own:
#
#     bash tools/experiments/make_lab.sh [DIR=tools/experiments/lab]
#     cd DIR && hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-math-errno -fno-slp-vectorize \
#               -DAMCX_ABL_NOFFT tools/wave_clock.hip -o tools/wave_clock_nofft
#     cd DIR && AMCX_EXTRA_FLAGS=-DAMCX_EXP_WAVES12 python3 amcpy_amd/csrc/build.py --output amcpy_amd/lib/libamcx_exp.so
#
# The product tree never sees these branches.  With no -D flag the lab's library has the same gfx950 code object as the
# product library of LAB_BASE (make_lab.sh --check builds both and compares their SHA-256).
set -e
LAB_BASE=b5f15f3          # "Product kernels without the laboratory"
cd "$(dirname "$0")/../.."
CHECK=0; DIR=tools/experiments/lab
for a in "$@"; do if [ "$a" = "--check" ]; then CHECK=1; else DIR=$a; fi; done
# DIR is deleted and rebuilt: only ever a directory this script made before (it leaves a marker file in it), never an
# existing path that happens to be named on the command line (`make_lab.sh amcpy_amd` would have removed the package)
MARK=.amcx_lab_tree
if [ -e "$DIR" ] && [ ! -f "$DIR/$MARK" ]; then
  echo "make_lab.sh: $DIR exists and was not made by this script (no $MARK in it): refusing to delete it" >&2; exit 2
fi
rm -rf "$DIR"; mkdir -p "$DIR"; : > "$DIR/$MARK"
git archive "$LAB_BASE" amcpy_amd include tools/wave_clock.hip tools/wave_stamps.hip tools/ab_lib.sh tools/ab_libs.sh \
    tools/ab_lib_timing.py tools/ab_wave_clock.sh tools/ab_summary.py tools/ab_bench_d2h.sh tools/resource_usage.py bench.py | tar -x -C "$DIR"
cp tools/experiments/amcx_pair_kernel.h tools/experiments/amcx_fixup_kernel.h "$DIR/amcpy_amd/csrc/"
patch -s -p1 -d "$DIR" < tools/experiments/r5_lab_branches.patch
echo "lab tree at $DIR (base $LAB_BASE + r5_lab_branches.patch)"
if [ $CHECK = 1 ]; then
  T=$(mktemp -d)
  git archive "$LAB_BASE" amcpy_amd include | tar -x -C "$T"
  python3 "$T/amcpy_amd/csrc/build.py" --output "$T/base.so" > /dev/null 2>&1
  python3 "$DIR/amcpy_amd/csrc/build.py" --output "$DIR/lab_noflags.so" > /dev/null 2>&1
  python3 tools/codeobj_gate.py "$T/base.so" "$DIR/lab_noflags.so"
  rm -rf "$T"
fi

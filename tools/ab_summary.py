#!/usr/bin/env python3
"""Condense the output of tools/ab_libs.sh / ab_lib.sh (one block per library and round) into one line per
(frame size, library): the 'bench modulations' rates of every round and their mean.
    python tools/ab_summary.py gpurun_out/.../ab.txt"""
import collections
import sys

cur, N, res = None, "?", collections.OrderedDict()
for line in open(sys.argv[1]):
    if line.startswith("### N="):
        N = line.split("=")[1].strip()
    elif line.startswith("## "):
        cur = line[3:].strip().split("/")[-1]
    elif "bench modulations" in line and "->" in line:
        res.setdefault((N, cur), []).append(float(line.split("->")[1].split()[0]))
base = {}
for (n, lib), v in res.items():
    mean = sum(v) / len(v)
    base.setdefault(n, mean)
    print(f"N={n:>5} {lib:34s} {' '.join('%6.1f' % x for x in v)}   mean {mean:6.1f}  ({100 * (mean / base[n] - 1):+.1f} % vs the first)")

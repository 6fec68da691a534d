#!/usr/bin/env python3
"""Condense rocprofv3 CSV output (tools/profile.sh) into a short text + JSON summary."""
import csv, glob, json, os, re, sys
from collections import defaultdict

out = sys.argv[1]
summary = {}

# Which GPU program these numbers belong to: the digests of the library the profiled command loaded (AMCX_LIB, else the
# tree's) -- of its whole gfx950 code object and of each kernel's machine code (tools/codeobj_gate.py) -- and each
# kernel's registers as its code object states them (tools/resource_usage.py).  bench.py replays a committed summary
# into its line only while the running library's digests agree.
import importlib.util
from pathlib import Path
_TOOLS = Path(__file__).resolve().parent
def _tool(name):
    spec = importlib.util.spec_from_file_location(name, _TOOLS / f"{name}.py")
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod
_lib_path = os.environ.get("AMCX_LIB") or str(_TOOLS.parent / "amcpy_amd" / "lib" / "libamcx.so")
try:
    _gate, _res = _tool("codeobj_gate"), _tool("resource_usage")
    _whole = _gate.digests(_lib_path)["code_object_sha256"]
    _per_kernel = _gate.kernel_digests(_lib_path)
    _regs = _res.read(_lib_path)
except Exception as exc:                                    # summaries without an identity are never replayed
    _whole, _per_kernel, _regs = None, {}, {}
    summary["binary_error"] = repr(exc)
summary["binary"] = {"library": os.path.basename(_lib_path), "code_object_sha256": _whole}

def _kernel_identity(profiler_name):
    """(SHA-256 of the kernel's machine code, its code object's register metadata) for a rocprofv3 kernel name"""
    sha = _gate.kernel_digest(_per_kernel, profiler_name) if _per_kernel else None
    leaf = _gate._leaf(profiler_name) if _per_kernel else profiler_name
    regs = next((v for k, v in _regs.items() if _gate._leaf(k) == leaf), None) if _regs else None
    return sha, regs

def rows(pattern):
    for f in glob.glob(os.path.join(out, pattern), recursive=True):
        with open(f, newline="") as fh:
            for r in csv.DictReader(fh):
                yield r

# kernel stats
for r in rows("trace/**/*kernel_stats.csv"):
    name = r.get("Name", "")
    if "amcx" in name:
        summary.setdefault("kernel_stats", []).append(
            {k: r[k] for k in r if k in ("Name", "Calls", "TotalDurationNs", "AverageNs", "MinNs", "MaxNs", "Percentage", "StdDev")})
# per-dispatch durations from the kernel trace
dur = defaultdict(list)
start = defaultdict(list)
meta = {}
for r in rows("trace/**/*kernel_trace.csv"):
    n = r.get("Kernel_Name", "")
    if "amcx" in n:
        dur[n].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
        start[n].append(int(r["Start_Timestamp"]))
        meta[n] = {k: r.get(k) for k in ("VGPR_Count", "Accum_VGPR_Count", "SGPR_Count", "LDS_Block_Size", "Scratch_Size", "Workgroup_Size", "Grid_Size")}
# the bench under the profiler runs 5 warm-up launches, then the 20 launches its events time: the mean over the
# LAST `timed` dispatches (by start time) is the figure to hold against roofline.mean_launch_ms of the same run
timed = int(os.environ.get("AMCX_PROFILE_TIMED", 20))
def _last(n, v):
    order = sorted(range(len(v)), key=lambda i: start[n][i])
    tail = [v[i] for i in order[-timed:]]
    return sum(tail) / len(tail)
summary["dispatch_ns"] = {n: {"n": len(v), "mean": sum(v) / len(v), "mean_of_the_timed_launches": _last(n, v), "timed": min(timed, len(v)),
                              "min": min(v), "max": max(v), **meta[n]} for n, v in dur.items()}
# rocprofv3's VGPR_Count / Scratch_Size are copied as the profiler reports them (VGPR_Count 64 for a kernel whose code
# object says .vgpr_count 128: the profiler's field is in allocation units of its own); OCCUPANCY FOLLOWS THE CODE
# OBJECT's count (512 registers per SIMD lane / .vgpr_count waves), which is what `code_object` holds
for n, rec in summary["dispatch_ns"].items():
    sha, regs = _kernel_identity(n)
    rec["kernel_sha256"] = sha
    if regs:
        rec["code_object"] = {"vgpr_count": regs["vgpr"], "vgpr_spill_count": regs["spill"], "private_segment_fixed_size": regs["scratch"],
                              "waves_per_simd_by_registers": min(8, 512 // max(regs["vgpr"], 1))}
# counters: mean per dispatch of each counter for amcx feature kernels
ctr = defaultdict(lambda: defaultdict(list))
for d in ("pmc_fetch", "pmc_write", "pmc_sq1", "pmc_sq2"):
    for r in rows(f"{d}/**/*counter_collection.csv"):
        n = r.get("Kernel_Name", "")
        if "amcx_features18" in n:
            ctr[n][r["Counter_Name"]].append(float(r["Counter_Value"]))
summary["counters_mean_per_dispatch"] = {n: {c: sum(v) / len(v) for c, v in cs.items()} for n, cs in ctr.items()}
# HBM traffic per frame for bench.py's roofline.traffic (gfx950: FETCH_SIZE counts 64 B per 128-B
# request on wide coalesced reads -> x2, MI355X_MICROARCH.md section HBM; both counters are in KiB)
frames = int(os.environ.get("AMCX_PROFILE_FRAMES", 6 * 26 * 4096))
# the DOMINANT feature kernel of the profiled command (most time in the trace): a bench run may launch others behind its
# timed region (other_configs), and their counters must not stand in for the timed kernel's
_total = {n: v["mean"] * v["n"] for n, v in summary["dispatch_ns"].items() if "amcx_features18" in n}
_dominant = max(_total, key=_total.get) if _total else None
for n, cs in summary["counters_mean_per_dispatch"].items():
    if _dominant is not None and n != _dominant:
        continue
    if "FETCH_SIZE" in cs and "WRITE_SIZE" in cs:
        rd, wr = cs["FETCH_SIZE"] * 1024 * 2, cs["WRITE_SIZE"] * 1024
        m = re.search(r"(?:wave|short)_kernel<(\d+)>", n)
        g = re.search(r"group_kernel<(\d+)>", n)               # N = 2048 x the number of waves per frame
        fs = int(m.group(1)) if m else 2048 * int(g.group(1)) if g else 8192 if "quad_kernel" in n else \
            int(os.environ.get("AMCX_PROFILE_FRAME_SIZE", 2048))
        summary["pmc_traffic"] = {"kernel": n, "code_object_sha256": _whole, "kernel_sha256": _kernel_identity(n)[0],
                                  "frame_size": fs, "frames_per_launch": frames,
                                  "fetch_bytes_corrected": rd, "write_bytes": wr,
                                  "hbm_bytes_per_frame": (rd + wr) / frames,
                                  "algorithmic_bytes_per_frame": 8 * fs + 72,
                                  "note": "FETCH_SIZE x2 (gfx950 wide-read undercount), separate --pmc passes"}
print(json.dumps(summary, indent=1))
if "pmc_traffic" in summary:
    with open(os.path.join(out, "pmc_traffic.json"), "w") as fh:
        json.dump(summary["pmc_traffic"], fh, indent=1)

#!/usr/bin/env python3
"""profiles/<TAG>_wave_budget.json: what bench.py replays into roofline.secondary for the N = 2048 kernel -- VALU / LDS
instructions per frame, wait shares (the PMC passes of tools/profile.sh), in-kernel clock and SIMD cycles per frame
(tools/wave_clock) -- TOGETHER WITH the digests of the binary they were measured on (tools/prof_summary.py puts them into
the summary): bench.py replays the budget only while the running library's N = 2048 kernel has the same machine code.

    python tools/make_wave_budget.py profiles/r6_n2048_summary.json profiles/r6_wave_clock.txt > profiles/r6_wave_budget.json
"""
import json
import re
import sys

summary = json.load(open(sys.argv[1]))
clock_txt = open(sys.argv[2]).read() if len(sys.argv) > 2 else ""
name = next(n for n in summary["counters_mean_per_dispatch"] if "wave_kernel<2048>" in n)
c = summary["counters_mean_per_dispatch"][name]
disp = summary["dispatch_ns"][name]
frames = summary.get("pmc_traffic", {}).get("frames_per_launch", 6 * 26 * 4096)
out = {
    "kernel": "amcx_features18_wave_kernel<2048>, BASELINE configs[1] (%d frames per launch)" % frames,
    "kernel_sha256": disp.get("kernel_sha256"),
    "code_object_sha256": summary.get("binary", {}).get("code_object_sha256"),
    "code_object": disp.get("code_object"),
    "bound": "board power cap (1 400 W) first, VALU issue second",
    "valu_instr_per_frame": c["SQ_INSTS_VALU"] / frames,
    "lds_instr_per_frame": c["SQ_INSTS_LDS"] / frames,
    "vmem_rd_instr_per_frame": c["SQ_INSTS_VMEM_RD"] / frames,
    "wait_shares_of_wave_cycles": {k: c[k] / c["SQ_WAVE_CYCLES"] for k in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY")},
    "kernel_us_profiler": disp["mean_of_the_timed_launches"] / 1e3,
    "sources": [sys.argv[1]] + sys.argv[2:3],
    "method": "tools/profile.sh (rocprofv3 --pmc, separate passes) / tools/wave_clock.hip: product instruction stream, one "
              "s_memtime / s_memrealtime pair around the frame loop, >= 2.5 s of back-to-back launches, median over waves",
}
for label, key_clock, key_rate in (("random", "in_kernel_clock_GHz", "frames_per_s_random_from_HBM"),
                                   ("zeros", "in_kernel_clock_zeros_GHz", "frames_per_s_on_zero_data")):
    m = re.search(rf"{label},\s+all CUs\s+N=2048 .*?([\d.]+) M frames/s \| in-kernel clock ([\d.]+) GHz.*?SIMD cycles/frame (\d+)", clock_txt)
    if m:
        out[key_clock] = float(m.group(2))
        out[key_rate] = float(m.group(1)) * 1e6
        if label == "random":
            out["simd_cycles_per_frame"] = int(m.group(3))
if out.get("simd_cycles_per_frame"):
    # issue slots: a SIMD issues one wave-wide VALU instruction every 2 cycles at best (profiles/r1_valu_issue_rates.txt,
    # the measure rounds 2-5 reported); simd_cycles_per_frame = SIMD cycles per frame of the SIMD's four waves together
    out["valu_issue_slot_use"] = out["valu_instr_per_frame"] * 2.0 / out["simd_cycles_per_frame"]
print(json.dumps(out, indent=1))

#!/usr/bin/env python3
"""Registers / spills / scratch of every kernel IN THE BUILT LIBRARY, read from the code object's own metadata
(amcpy_amd/lib/libamcx.so -> .hip_fatbin -> gfx950 code object -> NT_AMDGPU_METADATA): what the GPU will run, not what
a rebuild would report.

    python tools/resource_usage.py [name filter]          # table
    python tools/resource_usage.py --json                 # {kernel: {vgpr, spill, scratch, sgpr, max_threads}}
    python tools/resource_usage.py --update               # rewrite amcpy_amd/csrc/kernel_resources.json

tests/test_host_cpu.py::test_kernel_resources_match_the_committed_table holds every product kernel to
amcpy_amd/csrc/kernel_resources.json (the table DESIGN.md section 4 quotes): a kernel that starts to spill, or loses a
wave per SIMD, fails the CPU suite."""
import json
import re
import subprocess
import sys
import tempfile
from pathlib import Path

REPO = Path(__file__).resolve().parents[1]
LLVM = Path("/opt/rocm/lib/llvm/bin")
TABLE = REPO / "amcpy_amd" / "csrc" / "kernel_resources.json"


def short(mangled: str) -> str:
    name = subprocess.run(["c++filt", mangled], capture_output=True, text=True).stdout.strip()
    name = name.replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "")
    for ns in ("amcx::wave::", "amcx::quad::", "amcx::pair::", "amcx::", "(anonymous namespace)::"):
        name = name.replace(ns, "")
    return name


def read(lib=None) -> dict:
    lib = Path(lib) if lib else REPO / "amcpy_amd" / "lib" / "libamcx.so"
    with tempfile.TemporaryDirectory() as d:
        fat, co = Path(d) / "fat.bin", Path(d) / "gfx950.co"
        subprocess.run([str(LLVM / "llvm-objcopy"), f"--dump-section=.hip_fatbin={fat}", str(lib)], check=True)
        subprocess.run([str(LLVM / "clang-offload-bundler"), "--unbundle", "--type=o",
                        "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--input={fat}", f"--output={co}"], check=True)
        notes = subprocess.run([str(LLVM / "llvm-readelf"), "--notes", str(co)], capture_output=True, text=True, check=True).stdout
    out = {}
    for block in notes.split("  - .agpr_count:")[1:]:
        def field(key):
            m = re.search(rf"\.{key}:\s+(\S+)", block)
            return m.group(1) if m else None
        out[short(field("name"))] = {"vgpr": int(field("vgpr_count")), "spill": int(field("vgpr_spill_count")),
                                     "scratch": int(field("private_segment_fixed_size")), "sgpr": int(field("sgpr_count")),
                                     "max_threads": int(field("max_flat_workgroup_size"))}
    return out


if __name__ == "__main__":
    rows = read()
    if "--json" in sys.argv:
        print(json.dumps(rows, indent=1, sort_keys=True))
    elif "--update" in sys.argv:
        keep = {k: {f: v[f] for f in ("vgpr", "spill", "scratch")} for k, v in sorted(rows.items())}
        TABLE.write_text(json.dumps(keep, indent=1) + "\n")
        print(f"wrote {TABLE} ({len(keep)} kernels)")
    else:
        flt = next((a for a in sys.argv[1:] if not a.startswith("--")), "")
        for k, v in sorted(rows.items()):
            if flt in k:
                waves = min(8, 512 // max(v["vgpr"], 1))
                print(f"{k:50s} VGPR {v['vgpr']:4d}  spilled {v['spill']:4d}  scratch {v['scratch']:5d} B  SGPR {v['sgpr']:4d}"
                      f"  waves/SIMD by registers {waves}")

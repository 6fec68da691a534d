#!/usr/bin/env python3
"""Build libamcx.so for gfx950 in-tree (amcpy_amd/lib/libamcx.so).

    python amcpy_amd/csrc/build.py [--force] [--save-temps]

hipcc cross-compiles without a GPU.  The library links only the HIP runtime
(libamdhip64.so.7); in a process that has imported torch first, the loader
binds it to the runtime torch already loaded, so device pointers are shared.
"""
import argparse
import os
import subprocess
import sys
from pathlib import Path

HERE = Path(__file__).resolve().parent
LIB_DIR = HERE.parent / "lib"
LIB = LIB_DIR / "libamcx.so"
SOURCES = [HERE / "amcx.hip"]
HEADERS = sorted(HERE.glob("*.h")) + [HERE.parents[1] / "include" / "amcx.h"]
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
         "-ffp-contract=off", "-fno-math-errno", "-fno-slp-vectorize", "-Wall", "-Wno-unused-function",
         "-Wl,-rpath,/opt/rocm/lib", "-Wl,-soname,libamcx.so"]


def stale() -> bool:
    if not LIB.exists():
        return True
    t = LIB.stat().st_mtime
    return any(p.stat().st_mtime > t for p in SOURCES + HEADERS + [Path(__file__)])


def build(force: bool = False, save_temps: bool = False, verbose: bool = True) -> Path:
    if not force and not stale():
        return LIB
    LIB_DIR.mkdir(exist_ok=True)
    tmp_lib = LIB.with_name(f"{LIB.name}.{os.getpid()}.tmp")     # linked aside, renamed when complete
    # AMCX_EXTRA_FLAGS: experiment macros for same-box A/B runs of the whole library (e.g. -DAMCX_EXP_WAVES12)
    cmd = [HIPCC, *FLAGS, *os.environ.get("AMCX_EXTRA_FLAGS", "").split(), *map(str, SOURCES), "-o", str(tmp_lib)]
    cwd = HERE
    if save_temps:                       # the intermediate files (.s, .bc, ...) land in csrc/build/, which is git- and gpurun-ignored
        cwd = HERE / "build"
        cwd.mkdir(exist_ok=True)
        cmd += ["-save-temps=cwd", "-Rpass-analysis=kernel-resource-usage"]
    if verbose:
        print("+", " ".join(cmd), file=sys.stderr)
    try:
        subprocess.run(cmd, check=True, cwd=str(cwd))
        os.replace(tmp_lib, LIB)
    finally:
        if tmp_lib.exists():
            tmp_lib.unlink()
    return LIB


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--force", action="store_true")
    ap.add_argument("--save-temps", action="store_true")
    a = ap.parse_args()
    print(build(a.force, a.save_temps))

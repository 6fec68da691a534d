#!/usr/bin/env python3
"""Build libamcx.so for gfx950 in-tree (amcpy_amd/lib/libamcx.so).

    python amcpy_amd/csrc/build.py [--force] [--save-temps] [--output OTHER.so]

The product library is always built from the sources alone.  An experiment build (AMCX_EXTRA_FLAGS, e.g.
-DAMCX_PRIO_MASK=3, or one of the laboratory's switches inside tools/experiments/lab) goes to ANOTHER file -- `--output amcpy_amd/lib/libamcx_exp.so`, selected at
run time with AMCX_LIB=... -- and never replaces libamcx.so: extra flags without --output are refused.

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
         "-Wl,-rpath,/opt/rocm/lib", "-Wl,-soname,libamcx.so",
         # a fixed compilation-unit id: hipcc otherwise derives it from the paths on its command line (the temporary output name
         # carries the pid), it ends up in the names of the internal-linkage kernels, and two builds of one tree differ in bytes.
         # With it the library is reproducible byte for byte, which tools/codeobj_gate.py relies on.
         "-cuid=amcx"]


def stale() -> bool:
    if not LIB.exists():
        return True
    t = LIB.stat().st_mtime
    return any(p.stat().st_mtime > t for p in SOURCES + HEADERS + [Path(__file__)])


def build(force: bool = False, save_temps: bool = False, verbose: bool = True, output=None) -> Path:
    extra = os.environ.get("AMCX_EXTRA_FLAGS", "").split()
    if output is None:
        if extra:
            raise SystemExit("AMCX_EXTRA_FLAGS builds an experiment: give it its own file (--output amcpy_amd/lib/libamcx_exp.so, "
                             "run with AMCX_LIB=that file); libamcx.so is only ever the product build")
        if not force and not stale():
            return LIB
        target = LIB
    else:
        target = Path(output).resolve()
        if target == LIB.resolve():
            raise SystemExit("--output must not be the product library")
    target.parent.mkdir(exist_ok=True)
    tmp_lib = target.with_name(f"{target.name}.{os.getpid()}.tmp")     # linked aside, renamed when complete
    cmd = [HIPCC, *FLAGS, *extra, *map(str, SOURCES), "-o", str(tmp_lib)]
    cwd = HERE
    if save_temps:                       # the intermediate files (.s, .bc, ...) land in csrc/build/, which is git- and gpurun-ignored
        cwd = HERE / "build"
        cwd.mkdir(exist_ok=True)
        cmd += ["-save-temps=cwd", "-Rpass-analysis=kernel-resource-usage"]
    if verbose:
        print("+", " ".join(cmd), file=sys.stderr)
    try:
        subprocess.run(cmd, check=True, cwd=str(cwd))
        os.replace(tmp_lib, target)
    finally:
        if tmp_lib.exists():
            tmp_lib.unlink()
    return target


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--force", action="store_true")
    ap.add_argument("--save-temps", action="store_true")
    ap.add_argument("--output", default=None, help="build to this file instead of the product library (experiments)")
    a = ap.parse_args()
    print(build(a.force, a.save_temps, output=a.output))

#!/usr/bin/env python3
"""Are two builds of libamcx.so the same GPU program?

    python tools/codeobj_gate.py A.so B.so          # exit 0 iff the gfx950 code objects are byte-identical
    python tools/codeobj_gate.py --ref REV           # builds REV's sources aside and compares with amcpy_amd/lib/libamcx.so
    python tools/codeobj_gate.py --print [LIB]       # SHA-256 of the code object and of its .text section
    python tools/codeobj_gate.py --update            # rewrite amcpy_amd/csrc/codeobj.json from the built library

The gfx950 code object is taken out of the library the way tools/resource_usage.py does (.hip_fatbin -> clang-offload-
bundler).  The build passes a fixed -cuid (amcpy_amd/csrc/build.py), so the same sources give the same bytes wherever
and under whatever name they are built: a refactoring that claims "no change to the kernels" is held to that, and
tests/test_host_cpu.py holds the built library to the committed amcpy_amd/csrc/codeobj.json."""
import hashlib
import json
import subprocess
import sys
import tempfile
from pathlib import Path

REPO = Path(__file__).resolve().parents[1]
LLVM = Path("/opt/rocm/lib/llvm/bin")
LIB = REPO / "amcpy_amd" / "lib" / "libamcx.so"
TABLE = REPO / "amcpy_amd" / "csrc" / "codeobj.json"


def digests(lib) -> dict:
    with tempfile.TemporaryDirectory() as d:
        fat, co, text = Path(d) / "fat.bin", Path(d) / "gfx950.co", Path(d) / "text.bin"
        subprocess.run([str(LLVM / "llvm-objcopy"), f"--dump-section=.hip_fatbin={fat}", str(lib)], check=True)
        subprocess.run([str(LLVM / "clang-offload-bundler"), "--unbundle", "--type=o",
                        "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--input={fat}", f"--output={co}"], check=True)
        subprocess.run([str(LLVM / "llvm-objcopy"), "-O", "binary", "--only-section=.text", str(co), str(text)], check=True)
        return {"code_object_sha256": hashlib.sha256(co.read_bytes()).hexdigest(),
                "text_sha256": hashlib.sha256(text.read_bytes()).hexdigest(), "text_bytes": text.stat().st_size}


def build_rev(rev: str, out: Path) -> None:
    with tempfile.TemporaryDirectory() as d:
        tar = subprocess.run(["git", "-C", str(REPO), "archive", rev, "amcpy_amd", "include"], check=True, capture_output=True).stdout
        subprocess.run(["tar", "-x", "-C", d], input=tar, check=True)
        subprocess.run([sys.executable, str(Path(d) / "amcpy_amd" / "csrc" / "build.py"), "--output", str(out)], check=True,
                       capture_output=True)


def main(argv) -> int:
    if argv and argv[0] == "--print":
        print(json.dumps(digests(argv[1] if len(argv) > 1 else LIB), indent=1))
        return 0
    if argv and argv[0] == "--update":
        TABLE.write_text(json.dumps(digests(LIB), indent=1) + "\n")
        print(f"wrote {TABLE}")
        return 0
    if argv and argv[0] == "--ref":
        with tempfile.TemporaryDirectory() as d:
            other = Path(d) / "ref.so"
            build_rev(argv[1], other)
            a, b, names = digests(other), digests(LIB), (argv[1], str(LIB))
    elif len(argv) == 2:
        a, b, names = digests(argv[0]), digests(argv[1]), argv
    else:
        print(__doc__)
        return 2
    same = a["code_object_sha256"] == b["code_object_sha256"]
    for n, d in zip(names, (a, b)):
        print(f"{d['code_object_sha256']}  code object   {d['text_sha256'][:16]}… .text ({d['text_bytes']} B)   {n}")
    print("IDENTICAL" if same else ("DIFFERENT (.text identical: only symbol names / notes differ)" if a["text_sha256"] == b["text_sha256"]
                                    else "DIFFERENT"))
    return 0 if same else 1


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))

#!/usr/bin/env python3
"""Are two builds of libamcx.so the same GPU program?

    python tools/codeobj_gate.py A.so B.so          # exit 0 iff the gfx950 code objects are byte-identical
    python tools/codeobj_gate.py --ref REV           # builds REV's sources aside and compares with amcpy_amd/lib/libamcx.so
    python tools/codeobj_gate.py --print [LIB]       # SHA-256 of the code object and of its .text section
    python tools/codeobj_gate.py --kernels A.so B.so # per KERNEL: which kernels' machine code differs (a change to one kernel,
                                                     # or to a header several share, must leave the others' bytes alone)
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


def kernel_digests(lib) -> dict:
    """{demangled kernel / device function name: SHA-256 of its bytes in .text} of the gfx950 code object."""
    with tempfile.TemporaryDirectory() as d:
        fat, co = Path(d) / "fat.bin", Path(d) / "gfx950.co"
        subprocess.run([str(LLVM / "llvm-objcopy"), f"--dump-section=.hip_fatbin={fat}", str(lib)], check=True)
        subprocess.run([str(LLVM / "clang-offload-bundler"), "--unbundle", "--type=o",
                        "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--input={fat}", f"--output={co}"], check=True)
        sec = subprocess.run([str(LLVM / "llvm-readelf"), "-S", "-W", str(co)], capture_output=True, text=True, check=True).stdout
        addr = off = None
        for line in sec.splitlines():
            f = line.replace("[", " ").replace("]", " ").split()
            if len(f) > 5 and f[1] == ".text":
                addr, off = int(f[3], 16), int(f[4], 16)
        syms = subprocess.run([str(LLVM / "llvm-readelf"), "-s", "-W", str(co)], capture_output=True, text=True, check=True).stdout
        blob = co.read_bytes()
        out = {}
        for line in syms.splitlines():
            f = line.split()
            if len(f) >= 8 and f[3] == "FUNC" and int(f[2]) > 0:
                a, n, name = int(f[1], 16), int(f[2]), f[7]
                nice = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip().split("(")[0]
                nice = nice.replace("void ", "").replace("(anonymous namespace)::", "")
                out[nice] = hashlib.sha256(blob[off + a - addr: off + a - addr + n]).hexdigest()
        return out


def _leaf(name: str) -> str:
    """`amcx::wave::amcx_features18_wave_kernel<2048>(HIP_vector_type<...> const*, ...)` -> `amcx_features18_wave_kernel<2048>`"""
    name = name.replace("void ", "").replace("(anonymous namespace)::", "").strip()
    depth, cut = 0, len(name)
    for i, ch in enumerate(name):                      # the argument list opens at the first '(' outside template brackets
        if ch == "<":
            depth += 1
        elif ch == ">":
            depth -= 1
        elif ch == "(" and depth == 0:
            cut = i
            break
    name = name[:cut]
    depth, start = 0, 0
    for i, ch in enumerate(name):                      # the last '::' outside template brackets
        if ch == "<":
            depth += 1
        elif ch == ">":
            depth -= 1
        elif ch == ":" and depth == 0 and name[i:i + 2] == "::":
            start = i + 2
    return name[start:]


def kernel_digest(per_kernel: dict, name: str):
    """The digest of ONE kernel out of kernel_digests(), by whatever spelling of its name (with or without namespaces,
    'void', an argument list -- what rocprofv3, the library's amcx_kernel_name and c++filt each print); None if it is
    not there or ambiguous."""
    want = _leaf(name)
    hits = {v for k, v in per_kernel.items() if k and _leaf(k) == want}
    return hits.pop() if len(hits) == 1 else None


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
    if argv and argv[0] == "--kernels" and len(argv) == 3:
        a, b = kernel_digests(argv[1]), kernel_digests(argv[2])
        changed = sorted(k for k in a.keys() & b.keys() if a[k] != b[k])
        print(f"{len(a.keys() & b.keys()) - len(changed)} kernels / device functions identical")
        for k in changed:
            print("DIFFERENT ", k)
        for k in sorted(a.keys() - b.keys()):
            print("ONLY IN A ", k)
        for k in sorted(b.keys() - a.keys()):
            print("ONLY IN B ", k)
        return 0 if not changed else 1
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

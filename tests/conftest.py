"""Shared pytest wiring: the `gpu` marker and fixture loading."""
import json
import os
import sys
from pathlib import Path

import numpy as np
import pytest

REPO = Path(__file__).resolve().parents[1]
GOLDEN = REPO / "tests" / "golden"
if str(REPO) not in sys.path:
    sys.path.insert(0, str(REPO))
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


@pytest.fixture(scope="session", autouse=True)
def _library_is_built():
    """A fresh checkout has no libamcx.so (build artefacts stay out of git), and an edited kernel
    must not be tested through a stale library: (re)build when any source is newer, as
    __graft_entry__.build() does.  Nothing here substitutes for it -- without hipcc the
    tests that need the library fail with the loader's ImportError."""
    from amcpy_amd.csrc import build as b
    if Path(b.HIPCC).exists():
        b.build(force=False, verbose=False)     # no-op unless a source is newer than the library (build.stale)


@pytest.fixture(scope="session")
def kat():
    return json.loads((GOLDEN / "kat_n10.json").read_text())


def load_npz(name):
    """A fixture as a dict.  The large ones (frame sizes above 8192: oracle/capture_golden.py SEEDED_FROM) hold the
    SHA-256 of their inputs and the name of the numpy-only recipe that rebuilds them instead of megabytes of samples:
    the inputs are rebuilt here and checked against the digest."""
    import hashlib
    d = dict(np.load(GOLDEN / name, allow_pickle=False))
    if "iq" not in d and "iq_recipe" in d:
        from oracle import capture_golden as cg            # recipes only: nothing of the reference is imported
        fn, arg = str(d["iq_recipe"]).rstrip(")").split("(")
        assert fn in ("frames_inputs", "edges_inputs", "range_inputs", "range_extreme_inputs"), fn
        x = getattr(cg, fn)(int(arg))[0]
        assert hashlib.sha256(np.ascontiguousarray(x).tobytes()).hexdigest() == str(d["iq_sha256"]), \
            f"{name}: the rebuilt inputs differ from the ones the reference was run on"
        d["iq"] = x
    return d


@pytest.fixture(scope="session", params=[128, 256, 512, 1000, 1024, 2048, 4096, 5000, 8192, 10000, 12289, 16384, 32767, 32768])
def golden_frames(request):
    return request.param, load_npz(f"frames_n{request.param}.npz")


@pytest.fixture(scope="session", params=[1000, 1024, 2048, 4096, 8192, 10000, 16384, 32767, 32768])
def golden_edges(request):
    return request.param, load_npz(f"edges_n{request.param}.npz")

"""`python -m amcpy_amd extract [--root DIR] [--frame-size N] [--num-frames F] [--snr-values L ...]
                              [--device D | --devices 0,1,...|all] [--resume]`

The `extract` sub-command of the reference's CLI (src/amcpy/main.py:32,85-87,
160-175), and only that one: plot/train/eval/quantize are outside the hot path
(SURVEY.md section 8).  The reference's dispatcher calls ``cmd_extract(cfg, args)``
on a one-argument function (main.py:175 vs :85) and raises TypeError as
written; this entry point takes the same defaults and simply works.

Several GPUs, ONE command (the reference's caller runs one command and the parallelism happens inside,
feature_extraction.py:89-97): ``--devices 0,1,2,3`` / ``--devices all`` drives one engine per device from one
host thread each inside this process (feature_extraction.DeviceFanOut) -- no launcher, no process group, no
torch.  Started under ``torch.distributed.run`` instead (RANK / WORLD_SIZE in the environment), every rank takes
the GPU of its LOCAL_RANK, joins the process group and computes its share; rank 0 writes the files.
"""
from __future__ import annotations

import argparse
import os
import time
import sys
from dataclasses import replace
from pathlib import Path

from .config import Config, Paths


def build_parser() -> argparse.ArgumentParser:
    ap = argparse.ArgumentParser(prog="amcpy_amd", description="MI355X IQ feature extraction")
    sub = ap.add_subparsers(dest="command", required=True)
    ex = sub.add_parser("extract", help="compute the 18 features for every modulation container")
    ex.add_argument("--root", type=Path, default=None, help="project root (default: cwd)")
    ex.add_argument("--frame-size", type=int, default=None)
    ex.add_argument("--num-frames", type=int, default=None)
    ex.add_argument("--snr-values", nargs="+", default=None, metavar="LABEL",
                    help="SNR labels of the container's first axis, in order (default: the 16 of SignalConfig)")
    ex.add_argument("--device", type=int, default=None, help="GPU index (default: current device)")
    ex.add_argument("--resume", action="store_true",
                    help="skip modulations whose {mod}_features.mat is already complete for this configuration")
    ex.add_argument("--devices", default=None, metavar="0,1,...|all",
                    help="several GPUs from this one process, frames cut across them (one engine and host thread each)")
    return ap


def _parse_devices(spec: str):
    from . import _lib
    if spec.strip().lower() == "all":
        n = _lib.load().amcx_device_count()
        if n < 1:
            raise SystemExit("--devices all: no gfx950 device is visible")
        return list(range(n))
    try:
        devs = [int(t) for t in spec.split(",") if t.strip() != ""]
    except ValueError:
        raise SystemExit(f"--devices {spec!r}: expected a comma-separated list of GPU indices, or 'all'")
    if not devs or min(devs) < 0:
        raise SystemExit(f"--devices {spec!r}: expected a comma-separated list of GPU indices, or 'all'")
    return devs


def _run_as_rank(cfg, args) -> None:
    """One rank of a launcher's job (torch.distributed.run: RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set): the GPU of
    this rank's LOCAL_RANK, the process group, this rank's share of every modulation (run_extraction's several-rank
    path; rank 0 gathers and writes).  AMCX_DIST_BACKEND=gloo and AMCX_SHARE_GPU=1 rehearse it on a one-GPU box."""
    import torch
    import torch.distributed as dist
    from .feature_extraction import run_extraction
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    local = int(os.environ.get("LOCAL_RANK", rank))
    n_dev = torch.cuda.device_count()
    if args.device is not None:
        dev = args.device
    elif os.environ.get("AMCX_SHARE_GPU", "0") == "1":
        dev = local % max(1, n_dev)
    else:
        dev = local
    problem = None if dev < n_dev else f"rank {rank}: no GPU {dev} ({n_dev} visible); one rank per GPU, or AMCX_SHARE_GPU=1"
    backend = os.environ.get("AMCX_DIST_BACKEND", "nccl")
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    # Every rank says whether it can take part BEFORE the process group exists: a rank that has no device must not leave
    # the others waiting in the rendezvous until the store times out (a launcher that tears the job down on the first
    # failure hides this; one that does not, hangs).  The launcher's store carries one status word per rank.
    # (torch's own env:// rendezvous: it knows whether the launcher's agent already serves the store -- torch.distributed.run
    # does -- or rank 0 has to)
    store, _, _ = next(dist.rendezvous("env://", rank=rank, world_size=world))
    # A rank that dies before it has said anything (import error, bad environment) must not hold the others for the
    # store's default timeout (minutes): the exchange has a minute of its own, and running out of it is an error that
    # names the ranks that stayed silent.
    from datetime import timedelta
    wait_s = float(os.environ.get("AMCX_STATUS_TIMEOUT", "60"))
    store.set_timeout(timedelta(seconds=wait_s))
    store.set(f"amcx/status/{rank}", problem or "ok")
    problems, silent = [], []
    for r in range(world):
        try:
            word = store.get(f"amcx/status/{r}").decode()                            # get() waits for the key
        except Exception:
            silent.append(r)
            continue
        if word != "ok":
            problems.append(word)
    if silent:
        problems.append(f"rank(s) {silent} reported nothing within {wait_s:.0f} s (died before the rendezvous?)")
    # Every rank has now read every status: say so before anyone leaves.  Without a launcher rank 0 SERVES the store, and
    # a rank 0 that exits on a problem while others are still inside store.get() takes the store down under them: they
    # would die of a connection error instead of printing the collected message.
    try:
        store.add("amcx/status/read", 1)
        if problems and rank == 0:
            deadline = time.monotonic() + min(wait_s, 30.0)
            while int(store.add("amcx/status/read", 0)) < world - len(silent) and time.monotonic() < deadline:
                time.sleep(0.02)
    except Exception:
        pass
    if problems:
        raise SystemExit("; ".join(problems))
    store.set_timeout(timedelta(seconds=1800))                                        # the process group's own default
    torch.cuda.set_device(dev)
    if backend == "nccl":
        dist.init_process_group("nccl", store=store, rank=rank, world_size=world, device_id=torch.device("cuda", dev))
    else:
        dist.init_process_group(backend, store=store, rank=rank, world_size=world)
    try:
        run_extraction(cfg, device=dev, verbose=rank == 0, resume=args.resume)
    finally:
        dist.destroy_process_group()


def main(argv=None, *, skip_torch: bool = False) -> int:
    """``skip_torch``: load libamcx on the system HIP runtime without importing torch -- what the command line
    (`python -m amcpy_amd`, amcpy_amd/__main__.py) asks for, a second faster.  An in-process caller keeps the
    default: the library then binds to torch's runtime and the tensor entry points stay usable afterwards."""
    args = build_parser().parse_args(argv)
    cfg = Config() if args.root is None else Config(paths=Paths(root=args.root))
    sig = cfg.signals
    if args.frame_size is not None:
        sig = replace(sig, frame_size=args.frame_size)
    if args.num_frames is not None:
        sig = replace(sig, num_frames=args.num_frames)
    if args.snr_values is not None:
        sig = replace(sig, snr_values={i: str(v) for i, v in enumerate(args.snr_values)})
    cfg = replace(cfg, signals=sig)
    if args.command == "extract":
        if args.devices is not None and args.device is not None:
            raise SystemExit("--device and --devices exclude each other")
        if int(os.environ.get("WORLD_SIZE", "1")) > 1 and "RANK" in os.environ:
            if args.devices is not None:
                raise SystemExit("--devices drives several GPUs from ONE process; under a launcher every rank has its own")
            _run_as_rank(cfg, args)
            return 0
        # One process, host containers in, files out: nothing here touches a torch tensor, so the import (a second of
        # start-up) can be skipped -- by loading the library first, not by changing the environment
        from . import _lib
        if skip_torch:
            _lib.load(skip_torch=True)
        from .feature_extraction import run_extraction
        if args.devices is not None:
            run_extraction(cfg, devices=_parse_devices(args.devices), resume=args.resume)
        else:
            run_extraction(cfg, device=args.device, resume=args.resume)
    return 0


if __name__ == "__main__":
    sys.exit(main(skip_torch=True))

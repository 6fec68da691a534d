"""`python -m amcpy_amd extract [--root DIR] [--frame-size N] [--num-frames F] [--snr-values L ...] [--device D]`

The `extract` sub-command of the reference's CLI (src/amcpy/main.py:32,85-87,
160-175), and only that one: plot/train/eval/quantize are outside the hot path
(SURVEY.md section 8).  The reference's dispatcher calls ``cmd_extract(cfg, args)``
on a one-argument function (main.py:175 vs :85) and raises TypeError as
written; this entry point takes the same defaults and simply works.
"""
from __future__ import annotations

import argparse
import os
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
    return ap


def main(argv=None) -> int:
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
        # One process, host containers in, files out: nothing here touches a torch tensor, so the import (a second of
        # start-up) is skipped -- unless a launcher started this as one rank of several (torch.distributed needs it).
        if "RANK" not in os.environ and "WORLD_SIZE" not in os.environ:
            os.environ.setdefault("AMCX_SKIP_TORCH", "1")
        from .feature_extraction import run_extraction
        run_extraction(cfg, device=args.device)
    return 0


if __name__ == "__main__":
    sys.exit(main())

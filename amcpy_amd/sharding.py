"""Frame sharding across the GPUs of one node.

Every frame is independent (reference feature_extraction.py:64-72 enqueues
each (snr, frame) on its own), so the flattened frame index g in [0, F) is cut
into contiguous blocks, rank r of W taking
``[r*ceil(F/W), min(F, (r+1)*ceil(F/W)))``: contiguous in memory, never
splitting a frame, and with no collective on the data path.  The only
cross-rank step is gathering the tiny (F x 18) float32 result on rank 0.
"""
from __future__ import annotations

from typing import Callable, Optional, Tuple

import numpy as np


def shard_range(n_frames: int, rank: int, world: int) -> Tuple[int, int]:
    """Half-open frame range of ``rank``; empty ranges are (k, k)."""
    if world < 1 or not 0 <= rank < world:
        raise ValueError(f"bad rank {rank} of {world}")
    if n_frames < 0:
        raise ValueError("n_frames must be >= 0")
    per = -(-n_frames // world) if n_frames else 0
    lo = min(n_frames, rank * per)
    return lo, min(n_frames, lo + per)


def gather_rows(local: np.ndarray, n_frames: int, rank: int, world: int, group=None) -> Optional[np.ndarray]:
    """Collect every rank's (n_local, 18) block on rank 0 as (n_frames, 18).

    One ``torch.distributed.gather`` of float32 tensors, each rank's block padded to the common
    ``ceil(F / W)`` rows -- a plain copy (92 MB per rank at BASELINE configs[3]), not a pickle -- on
    whatever backend the caller initialised: device tensors over RCCL for GPU jobs, host tensors
    over gloo in the CPU tests.  Returns the full matrix on rank 0 and None elsewhere."""
    if world == 1:
        return local
    import torch
    import torch.distributed as dist
    cols = local.shape[1]
    per = -(-n_frames // world) if n_frames else 0
    lo, hi = shard_range(n_frames, rank, world)
    if local.shape[0] != hi - lo:
        raise RuntimeError(f"rank {rank} holds {local.shape[0]} rows for [{lo}, {hi})")
    on_gpu = "nccl" in str(dist.get_backend(group)).lower()
    dev = torch.device("cuda", torch.cuda.current_device()) if on_gpu else torch.device("cpu")
    mine = torch.zeros((per, cols), dtype=torch.float32, device=dev)
    if hi > lo:
        mine[:hi - lo].copy_(torch.from_numpy(np.ascontiguousarray(local, dtype=np.float32)))
    parts = [torch.empty_like(mine) for _ in range(world)] if rank == 0 else None
    dist.gather(mine, parts, dst=0, group=group)
    if rank != 0:
        return None
    out = np.empty((n_frames, cols), dtype=np.float32)
    for r, blk in enumerate(parts):
        a, b = shard_range(n_frames, r, world)
        if b > a:
            out[a:b] = blk[:b - a].cpu().numpy()
    return out


def all_gather_rows(local, n_frames: int, rank: int, world: int, group=None):
    """Every rank's (n_local, C) block of a torch tensor -> the full (n_frames, C) tensor ON EVERY RANK, in the
    memory the blocks live in (device tensors over RCCL: ring / direct all-gather across xGMI; host tensors over
    gloo).  For a device-side consumer that wants the whole feature matrix -- the standardising scaler's fit, the
    per-SNR statistics (amcpy_amd/postprocess.py) -- without a trip through rank 0's host memory: 0.7 GB in all at
    BASELINE configs[3].  Blocks are padded to the common ceil(F / W) rows, as in :func:`gather_rows`."""
    import torch
    if world == 1:
        return local
    import torch.distributed as dist
    lo, hi = shard_range(n_frames, rank, world)
    if local.shape[0] != hi - lo:
        raise RuntimeError(f"rank {rank} holds {local.shape[0]} rows for [{lo}, {hi})")
    per = -(-n_frames // world) if n_frames else 0
    cols = local.shape[1]
    mine = torch.zeros((per, cols), dtype=local.dtype, device=local.device)
    if hi > lo:
        mine[:hi - lo].copy_(local)
    everything = torch.empty((world * per, cols), dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(everything, mine, group=group)
    if world * per == n_frames:
        return everything
    return torch.cat([everything[r * per:r * per + (b - a)] for r in range(world)
                      for a, b in [shard_range(n_frames, r, world)] if b > a], dim=0)


def sharded_features(frames: np.ndarray, frame_size: int, compute: Callable[[np.ndarray], np.ndarray],
                     rank: int = 0, world: int = 1, group=None) -> Optional[np.ndarray]:
    """(F, L) complex frames -> (F, 18) float32 on rank 0, each rank computing
    its contiguous block with ``compute`` (the HIP engine in production)."""
    F = frames.shape[0]
    lo, hi = shard_range(F, rank, world)
    local = compute(frames[lo:hi]) if hi > lo else np.empty((0, 18), dtype=np.float32)
    return gather_rows(np.asarray(local, dtype=np.float32), F, rank, world, group)

"""Frame sharding across the GPUs of one node.

Every frame is independent (reference feature_extraction.py:64-72 enqueues
each (snr, frame) on its own), so the flattened frame index g in [0, F) is cut
into contiguous blocks, rank r of W taking
``[r*ceil(F/W), min(F, (r+1)*ceil(F/W)))``: contiguous in memory, never
splitting a frame, and with no collective on the data path.  A container that
is still in a .mat (column-major: snr the fastest axis) is cut along its FRAME
axis instead when that balances as well (:func:`shard_by_frames`): a rank's share
is then one contiguous run of every sample plane.  The only cross-rank step
is gathering the tiny (F x 18) float32 result on rank 0.
"""
from __future__ import annotations

from typing import Callable, Optional, Tuple

import numpy as np


def collectives_forced() -> bool:
    """TEST SWITCH (AMCX_TEST_FORCE_COLLECTIVES=1): take the multi-rank code -- status words, the padded tensor gather,
    the all-gather -- even with a process group of ONE rank, instead of the world == 1 shortcuts.  One GPU is all a test
    box has; with this the branch that the first 8-GPU run will execute (device tensors over RCCL) has executed before:
    tests/test_gpu_parity.py::test_one_rank_nccl_takes_the_collective_path."""
    import os
    return os.environ.get("AMCX_TEST_FORCE_COLLECTIVES") == "1"


def shard_range(n_frames: int, rank: int, world: int) -> Tuple[int, int]:
    """Half-open frame range of ``rank``; empty ranges are (k, k)."""
    if world < 1 or not 0 <= rank < world:
        raise ValueError(f"bad rank {rank} of {world}")
    if n_frames < 0:
        raise ValueError("n_frames must be >= 0")
    per = -(-n_frames // world) if n_frames else 0
    lo = min(n_frames, rank * per)
    return lo, min(n_frames, lo + per)


def shard_by_frames(n_snr: int, n_frames: int, world: int) -> bool:
    """How a container of (n_snr, n_frames) frames is cut over ``world`` ranks: True -- every rank takes the frames
    ``shard_range(n_frames, rank, world)`` of EVERY snr row -- when that is as balanced (within 1/8) as cutting the
    snr-major flattening, False for the flattening.  A column-major container (what a .mat holds: snr the fastest
    axis) keeps a rank's frame range of all snr rows as ONE contiguous run per sample plane, which the staging
    threads read or copy at full rate; a range of the flattening leaves it n_snr / world elements per run."""
    if world <= 1 or n_frames < world:
        return False
    by_k = n_snr * (-(-n_frames // world))
    by_g = -(-(n_snr * n_frames) // world)
    return 8 * by_k <= 9 * by_g


def gather_blocks(local: np.ndarray, counts, rank: int, world: int, group=None):
    """Collect every rank's (counts[r], C) float32 block on rank 0: a list of the ``world`` blocks there, None
    elsewhere.  One ``torch.distributed.gather`` of tensors padded to ``max(counts)`` rows -- a plain copy (92 MB
    per rank at BASELINE configs[3]), not a pickle -- on whatever backend the caller initialised: device tensors
    over RCCL for GPU jobs, host tensors over gloo in the CPU tests."""
    counts = [int(c) for c in counts]
    if len(counts) != world or local.shape[0] != counts[rank]:
        raise RuntimeError(f"rank {rank} holds {local.shape[0]} rows, expected {counts}")
    if world == 1 and not collectives_forced():
        return [local]
    import torch
    import torch.distributed as dist
    cols = local.shape[1]
    per = max(counts)
    on_gpu = "nccl" in str(dist.get_backend(group)).lower()
    dev = torch.device("cuda", torch.cuda.current_device()) if on_gpu else torch.device("cpu")
    mine = torch.zeros((per, cols), dtype=torch.float32, device=dev)
    if counts[rank]:
        mine[:counts[rank]].copy_(torch.from_numpy(np.ascontiguousarray(local, dtype=np.float32)))
    parts = [torch.empty_like(mine) for _ in range(world)] if rank == 0 else None
    dist.gather(mine, parts, dst=0, group=group)
    if rank != 0:
        return None
    return [blk[:counts[r]].cpu().numpy() for r, blk in enumerate(parts)]


def gather_rows(local: np.ndarray, n_frames: int, rank: int, world: int, group=None) -> Optional[np.ndarray]:
    """Collect every rank's (n_local, 18) block of the contiguous cut (:func:`shard_range`) on rank 0 as
    (n_frames, 18): :func:`gather_blocks` + concatenation.  Returns the full matrix on rank 0 and None elsewhere."""
    if world == 1 and not collectives_forced():
        return local
    ranges = [shard_range(n_frames, r, world) for r in range(world)]
    lo, hi = ranges[rank]
    if local.shape[0] != hi - lo:
        raise RuntimeError(f"rank {rank} holds {local.shape[0]} rows for [{lo}, {hi})")
    blocks = gather_blocks(local, [b - a for a, b in ranges], rank, world, group)
    if blocks is None:
        return None
    out = np.empty((n_frames, local.shape[1]), dtype=np.float32)
    for (a, b), blk in zip(ranges, blocks):
        out[a:b] = blk
    return out


def gather_frame_columns(local: np.ndarray, n_snr: int, n_frames: int, rank: int, world: int, group=None):
    """The frame-axis cut (:func:`shard_by_frames`): rank r holds the (n_snr * K_r, C) rows of frames
    ``shard_range(n_frames, r, world)`` of every snr row, snr-major; rank 0 gets (n_snr, n_frames, C), others None."""
    ranges = [shard_range(n_frames, r, world) for r in range(world)]
    blocks = gather_blocks(local, [n_snr * (b - a) for a, b in ranges], rank, world, group)
    if blocks is None:
        return None
    out = np.empty((n_snr, n_frames, local.shape[1]), dtype=np.float32)
    for (a, b), blk in zip(ranges, blocks):
        out[:, a:b] = blk.reshape(n_snr, b - a, -1)
    return out


def all_gather_rows(local, n_frames: int, rank: int, world: int, group=None):
    """Every rank's (n_local, C) block of a torch tensor -> the full (n_frames, C) tensor ON EVERY RANK, in the
    memory the blocks live in (device tensors over RCCL: ring / direct all-gather across xGMI; host tensors over
    gloo).  For a device-side consumer that wants the whole feature matrix -- the standardising scaler's fit, the
    per-SNR statistics (amcpy_amd/postprocess.py) -- without a trip through rank 0's host memory: 0.7 GB in all at
    BASELINE configs[3].  Blocks are padded to the common ceil(F / W) rows, as in :func:`gather_rows`."""
    import torch
    if world == 1 and not collectives_forced():
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

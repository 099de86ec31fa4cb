"""Multi-GPU sharding of the texture path: one process per GPU, clips are independent.

The reference has no distributed code; frames and clips are independent, so the only
communication is one gather of per-frame result rows at the end of a job
(``gather_rows``: RCCL when the process group's backend is ``nccl``, gloo in CPU tests).
"""
from __future__ import annotations

from typing import List, Sequence

import numpy as np

ROW_WIDTH = 6  # five metrics (passive_checker.py:125-131) + the per-frame score (:145-151)


def shard_videos(n_videos: int, world: int, rank: int) -> List[int]:
    """Videos owned by ``rank``: video_id % world == rank."""
    if not (0 <= rank < world):
        raise ValueError("rank out of range")
    return list(range(rank, n_videos, world))


def max_local_videos(n_videos: int, world: int) -> int:
    return (n_videos + world - 1) // world


def rows_from_metrics(metrics: Sequence) -> np.ndarray:
    """[n][6] float64 rows from StxMetrics-like records."""
    out = np.empty((len(metrics), ROW_WIDTH), np.float64)
    for i, m in enumerate(metrics):
        out[i] = (m.lbp_variance, m.noise_level, m.compression_artifacts, m.edge_density, m.frequency_anomaly,
                  m.texture_score)
    return out


def gather_rows(local_rows: np.ndarray, n_videos: int, frames_per_video: int, device=None):
    """All-gather the per-frame rows of every rank's videos into video order.

    local_rows: [n_local_videos][frames_per_video][6] for the videos of ``shard_videos`` in order.
    Every rank pads to ``max_local_videos`` so the collective has equal counts (one
    all_gather_into_tensor, no per-batch collectives).  Returns [n_videos][frames_per_video][6].
    """
    import torch
    import torch.distributed as dist

    world = dist.get_world_size() if dist.is_initialized() else 1
    rank = dist.get_rank() if dist.is_initialized() else 0
    local_rows = np.asarray(local_rows, np.float64).reshape(-1, frames_per_video, ROW_WIDTH)
    mine = shard_videos(n_videos, world, rank)
    if local_rows.shape[0] != len(mine):
        raise ValueError(f"rank {rank} owns {len(mine)} videos but got {local_rows.shape[0]} row blocks")
    if world == 1:
        return local_rows.copy()
    cap = max_local_videos(n_videos, world)
    padded = np.zeros((cap, frames_per_video, ROW_WIDTH), np.float64)
    padded[: len(mine)] = local_rows
    t = torch.from_numpy(padded.reshape(-1))
    if device is not None:
        t = t.to(device)
    gathered = torch.empty(world * t.numel(), dtype=torch.float64, device=t.device)
    dist.all_gather_into_tensor(gathered, t)
    g = gathered.cpu().numpy().reshape(world, cap, frames_per_video, ROW_WIDTH)
    out = np.empty((n_videos, frames_per_video, ROW_WIDTH), np.float64)
    for r in range(world):
        vids = shard_videos(n_videos, world, r)
        out[vids] = g[r, : len(vids)]
    return out


def clip_scores(rows: np.ndarray, first_n: int = 10, threshold: float = 0.6):
    """compute_texture_score's reduction per clip (passive_checker.py:140, :155-156) from gathered rows:
    mean of the first ``first_n`` per-frame scores and the verdict."""
    rows = np.asarray(rows)
    avg = np.array([np.mean(v[:first_n, 5]) if v.shape[0] else 0.0 for v in rows])
    return avg, avg >= threshold

"""world_size-2 test of the multi-GPU sharding and the final score gather, on CPU with gloo.
Each rank scores its own videos (here with the CPU oracle standing in for the per-rank GPU
work) and one all-gather reassembles the rows in video order."""
import os
import socket
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _score_video(video, n_frames, h, w):
    from oracle import texture_oracle as O
    from sift_amd.sharding import rows_from_metrics
    from sift_amd.synth import synth_clip

    co = O.COracle()
    clip = synth_clip("natural", 0x5EED0003, video, n_frames, h, w)
    return rows_from_metrics([co.analyze_frame(f) for f in clip])


def _worker(rank, world, port, n_videos, n_frames, h, w, q):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist

    from sift_amd.sharding import gather_rows, shard_videos

    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    try:
        mine = shard_videos(n_videos, world, rank)
        local = np.stack([_score_video(v, n_frames, h, w) for v in mine]) if mine else np.zeros((0, n_frames, 6))
        allrows = gather_rows(local, n_videos, n_frames)
        q.put((rank, allrows))
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_shard_videos_partition():
    from sift_amd.sharding import max_local_videos, shard_videos
    for world in (1, 2, 3, 8):
        for n in (0, 1, 5, 8, 17):
            parts = [shard_videos(n, world, r) for r in range(world)]
            assert sorted(sum(parts, [])) == list(range(n))
            assert max(len(p) for p in parts) == (max_local_videos(n, world) if n else 0)
    with pytest.raises(ValueError):
        shard_videos(4, 2, 2)


def test_gather_rows_world2():
    n_videos, n_frames, h, w = 5, 3, 24, 40
    world = 2
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_videos, n_frames, h, w, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    want = np.stack([_score_video(v, n_frames, h, w) for v in range(n_videos)])
    for r in range(world):
        assert np.array_equal(got[r], want)
    from sift_amd.sharding import clip_scores
    avg, passed = clip_scores(want)
    assert avg.shape == (n_videos,) and passed.dtype == bool
    assert np.allclose(avg, want[:, :10, 5].mean(axis=1))

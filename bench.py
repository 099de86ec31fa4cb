#!/usr/bin/env python3
"""Benchmark of the texture-analysis hot path on MI355X.

Default workload (BASELINE.json configs[2]): synthetic 1080p 5 s clips (150 frames, ``natural``
distribution), full-frame texture analysis.  One step = one clip per GPU, already resident in
HBM when the timed region starts (generated on the device by stx_synth_device from
(seed, video id = step * world + rank)).  With N GPUs every rank analyses its own clips (weak
scaling, no data-path collective); the per-frame [5 metrics + score] rows of all steps are
all-gathered once, over RCCL, before the closing barrier.

``python bench.py --gpus N`` starts its own N ranks (one process per GPU) when it is not already
running under torch.distributed.run.  ``--config crops256`` / ``--config 4k_multiradius`` run
BASELINE configs[1] / configs[4] instead (single-GPU records kept under profiles/).

Prints ONE JSON line on rank 0.
"""
from __future__ import annotations

import argparse
import ctypes as C
import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak
PREWARM_STEPS = 12     # untimed clips before the W warm-up steps (workspace sizing, GPU clocks)
MAX_RESIDENT_BYTES = 48 << 30  # distinct input clips kept in HBM per rank

_shape = (0, 0)  # frame size of the run (set by main)

CONFIGS = {
    # name: frames per step, height, width, FFT crop (0 = whole frame), extended statistics, BASELINE row
    "1080p": dict(frames=150, h=1080, w=1920, fft=0, extended=False, row="configs[2]"),
    "720p": dict(frames=150, h=720, w=1280, fft=0, extended=False, row="configs[0] frame size, on the GPU"),
    "crops256": dict(frames=4096, h=256, w=256, fft=0, extended=False, row="configs[1]"),
    "4k_multiradius": dict(frames=40, h=2160, w=3840, fft=512, extended=True, row="configs[4]"),
    # 10,000 clips over all ranks by video id; a clip exists only while it is in flight (generated on the device
    # from (seed, video id) beside the analysis of the clip before it, three rotating buffers per rank)
    "clips10k": dict(frames=150, h=1080, w=1920, fft=0, extended=False, row="configs[3]", total_clips=10000),
}


def algorithmic_bytes(h: int, w: int, fh: int, fw: int) -> dict:
    """Per-frame algorithmic HBM bytes of the whole path and of each kernel group:
    BGR read once (3HW), Canny map written once and read once (2HW), fp32-complex
    half-spectrum written by the row pass and read by the column pass (2*8*fh*(fw/2+1));
    the spectrum term is 0 when the half spectrum fits one CU's 160 KiB LDS.  Further LBP
    radii add nothing (same tile, larger halo).

    Groups: "texture" = gray + stencil (+ generic LBP / extended) kernels (together they do what one
    fused texture kernel would: read the BGR frame, write the Canny map), "hysteresis" reads the
    map, "fft" = row + column passes (write + read of the half spectrum)."""
    spec = 8 * fh * (fw // 2 + 1)
    if spec <= 160 * 1024:
        spec = 0
    return {"path": 3 * h * w + 2 * h * w + 2 * spec, "texture": 3 * h * w + h * w, "hysteresis": h * w,
            "fft": 2 * spec}


GROUPS = {"texture": ("gray", "stencil", "lbp_generic", "extended"), "hysteresis": ("hysteresis",),
          "fft": ("fft_rows", "fft_cols")}


def kernel_algorithmic_bytes(h: int, w: int, fh: int, fw: int, separate_gray: bool) -> dict:
    """The same account split by kernel family (it sums to algorithmic_bytes()["path"]): the stencil pass reads the
    BGR frame and writes the Canny map (with a separate gray pass the 3HW belong to that pass), the hysteresis
    kernels read the map, the row pass writes the half spectrum and the column pass reads it.  The kernels of the
    extended statistics add nothing to the path's account (same tile, larger halo); as kernels of their own they
    are charged the one gray-plane read they cannot avoid."""
    spec = 8 * fh * (fw // 2 + 1)
    if spec <= 160 * 1024:
        spec = 0
    return {"gray": 3 * h * w if separate_gray else 0, "stencil": (h * w if separate_gray else 3 * h * w) + h * w,
            "hysteresis": h * w, "fft_rows": spec, "fft_cols": spec, "lbp_generic": h * w, "extended": h * w,
            "quality": h * w}


def measured_traffic(h: int, w: int):
    """Per-frame HBM bytes per kernel from the committed rocprofv3 PMC passes (profiles/), or None."""
    for name in ("r3_traffic.json", "r2_traffic.json", "r1_d_traffic.json"):
        try:
            with open(os.path.join(ROOT, "profiles", name)) as f:
                t = json.load(f)
            if t.get("height") == h and t.get("width") == w:
                return t["per_frame_bytes"]
        except (OSError, ValueError, KeyError):
            pass
    return None


def self_launch(args) -> int:
    """--gpus N outside torch.distributed.run: start N ranks from this process, which has not touched the
    GPU (and never will), and relay rank 0's line."""
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "1")
    return subprocess.call(cmd, env=env)


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=0,
                    help="timed steps per GPU (0 = 40; for clips10k: this rank's share of the 10,000 clips)")
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--config", choices=sorted(CONFIGS), default="1080p")
    ap.add_argument("--frames", type=int, default=0, help="frames per step (0 = the config's)")
    ap.add_argument("--height", type=int, default=0)
    ap.add_argument("--width", type=int, default=0)
    ap.add_argument("--kind", default="natural")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--host-fed", action="store_true",
                    help="also report the PCIe-inclusive rate from a pinned host batch (never the headline value)")
    ap.add_argument("--blocking", action="store_true",
                    help="one blocking call per step instead of two clips in flight through stx_submit_batch_device / "
                         "stx_wait (the default: the host queues clip i+1 and collects clip i-1 while the GPU works on "
                         "clip i, so the device does not idle between calls)")
    ap.add_argument("--time-all-kernels", action="store_true",
                    help="bracket every kernel with HIP events in the timed region, not only the dominant group's "
                         "(costs a few percent of throughput: each event pair delays its stream's next kernel)")
    ap.add_argument("--chunk", type=int, default=0, help="frames per internal chunk (0 = the library's choice)")
    ap.add_argument("--no-other-content", action="store_true",
                    help="skip the short runs on the other synthetic content classes (reported under other_content)")
    ap.add_argument("--cpu-seconds", type=float, default=24.0, help="wall-clock budget of the CPU baseline legs")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(self_launch(args))

    if os.environ.get("STX_BENCH_DRYRUN"):
        raise SystemExit(dry_run(args))

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    # STX_BENCH_BACKEND=gloo STX_BENCH_DEVICE=0 rehearses the multi-rank path on a one-GPU box
    backend = os.environ.get("STX_BENCH_BACKEND", "nccl")
    dev_override = os.environ.get("STX_BENCH_DEVICE")
    if dev_override is not None:
        local_rank = int(dev_override)
    torch.cuda.set_device(local_rank)
    if world > 1:
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)
    gather_device = "cuda" if backend == "nccl" else None

    from sift_amd._lib import StxExtended, StxMetrics
    from sift_amd.sharding import gather_rows
    from sift_amd.synth import KIND_ID
    from sift_amd.texture import TextureContext

    cfg = CONFIGS[args.config]
    streaming = "total_clips" in cfg  # clips are generated inside the loop and never kept
    if args.steps <= 0:
        args.steps = (cfg["total_clips"] + world - 1) // world if streaming else 40
    h, w = args.height or cfg["h"], args.width or cfg["w"]
    global _shape
    _shape = (h, w)
    nf = args.frames or cfg["frames"]
    fh = fw = cfg["fft"]
    fh, fw = (fh or h), (fw or w)
    ctx = TextureContext(local_rank)
    if args.chunk:
        ctx.set_chunk_frames(args.chunk)
    params = None
    if cfg["fft"]:
        params = ctx.default_params()
        params.fft_h, params.fft_w = fh, fw
    clip_bytes = nf * h * w * 3
    total_steps = max(args.steps, 1)
    n_clips = 3 if streaming else int(max(2, min(total_steps, MAX_RESIDENT_BYTES // clip_bytes)))
    clips = [torch.empty(clip_bytes, dtype=torch.uint8, device="cuda") for _ in range(n_clips)]
    seed = 0x5EED0000 + (3 if streaming else 2)
    gen_in_loop = [streaming]  # (a list: the generation-free comparison run below switches it off)

    def synth(t, video: int) -> None:
        # the generator numbers at most 1024 frames per video: longer batches continue in further videos
        parts = (nf + 1023) // 1024
        for f0 in range(0, nf, 1024):
            n = min(1024, nf - f0)
            ctx.synth_device(t.data_ptr() + f0 * h * w * 3, KIND_ID[args.kind], seed, video * parts + f0 // 1024, 0, n, h, w)

    for i, t in enumerate(clips):
        synth(t, i * world + rank)  # video id of step i on this rank
    out = (StxMetrics * nf)()
    outs = [(StxMetrics * nf)() for _ in range(2)]
    ext = (StxExtended * nf)() if cfg["extended"] else None
    rows = np.zeros((args.steps, nf, 6), np.float64)

    def keep_rows(src, keep: int) -> None:
        # the 6 doubles at the head of every stx_metrics record, without a Python loop per frame
        raw = np.frombuffer(src, dtype=np.float64).reshape(nf, C.sizeof(StxMetrics) // 8)
        rows[keep] = raw[:, :6]

    def step(i: int, keep: int = -1) -> None:
        ptr = clips[i % n_clips].data_ptr()
        if gen_in_loop[0]:
            ctx.synth_device_async(ptr, KIND_ID[args.kind], seed, i * world + rank, 0, nf, h, w)
        if cfg["extended"]:
            rc = ctx.lib.stx_analyze_extended_batch_device(ctx.handle, ptr, nf, h, w, 3 * w, 3 * w * h,
                                                           C.byref(params) if params is not None else None, out, ext)
            ctx._check(rc, "stx_analyze_extended_batch_device")
        else:
            ctx.analyze_device(ptr, nf, h, w, params=params, out=out)
        if keep >= 0:
            keep_rows(out, keep)

    # a clip must fit one asynchronous submission
    pipelined = not (args.blocking or nf > ctx.max_async_frames(h, w))
    exts = [(StxExtended * nf)() for _ in range(2)] if cfg["extended"] else None

    def run_steps(n: int, keep: bool) -> None:
        """n steps; unless --blocking, software-pipelined two deep through the asynchronous entry point: while
        the GPU works on clip i the host collects clip i-1 and queues clip i+1."""
        if not pipelined:
            for i in range(n):
                step(i, keep=i if keep else -1)
            return
        pending = None
        for i in range(n):
            ptr = clips[i % n_clips].data_ptr()
            if gen_in_loop[0]:
                # this buffer's last reader (clip i - 3) was collected two iterations ago; the generator runs beside
                # the kernels of clip i - 1, which the device is working on now
                ctx.synth_device_async(ptr, KIND_ID[args.kind], seed, i * world + rank, 0, nf, h, w)
            if cfg["extended"]:
                t = ctx.submit_device_extended(ptr, nf, h, w, outs[i % 2], exts[i % 2], params)
            else:
                t = ctx.submit_device(ptr, nf, h, w, outs[i % 2], params)
            if pending is not None:
                ctx.wait(pending[0])
                if keep:
                    keep_rows(outs[pending[1] % 2], pending[1])
            pending = (t, i)
        if pending is not None:
            ctx.wait(pending[0])
            if keep:
                keep_rows(outs[pending[1] % 2], pending[1])

    def sync() -> None:
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # untimed: the first calls size the workspace and the GPU leaves its idle clocks; then the W warm-up steps
    run_steps(PREWARM_STEPS if nf * h * w <= 150 * 1080 * 1920 else 3, keep=False)
    # one clip with every kernel back to back on one stream and an event pair around each: per-kernel durations with
    # the device to itself.  The group of kernels that takes longest here is the dominant one, and the only one
    # bracketed by events in the timed region: an event pair costs the stream a few microseconds on either side of
    # the kernel, which adds up over a chain of dependent kernels.
    ctx.set_concurrency(False)
    step(0)  # untimed: the measured clip below starts on a busy device, like every clip of the timed region
    ctx.get_timing(reset=True)
    ctx.set_timing(True)
    step(1)
    torch.cuda.synchronize()
    ctx.set_timing(False)
    solo = ctx.get_timing(reset=True)
    ctx.set_concurrency(True)
    solo_chunks = max(solo["kernels"]["stencil"]["launches"], 1)  # the stencil kernel runs once per chunk of frames
    solo_ms = {g: sum(solo["kernels"][k]["ms"] for k in ks) / solo_chunks for g, ks in GROUPS.items()}
    # The dominant kernel: the single kernel family with the longest duration per chunk (the hysteresis family is a
    # dozen short launches, the others are one launch per chunk each).
    kab = kernel_algorithmic_bytes(h, w, fh, fw, separate_gray=solo["kernels"]["gray"]["launches"] > 0)
    solo_kernel_ms = {k: v["ms"] / solo_chunks for k, v in solo["kernels"].items() if v["launches"] and kab.get(k)}
    dom = max(solo_kernel_ms, key=lambda k: solo_kernel_ms[k])
    run_steps(args.warmup, keep=False)
    analysis_only = None
    if streaming:
        # the same pipeline over the three clips left in the buffers, nothing generated: what the analysis alone does
        gen_in_loop[0] = False
        run_steps(4, keep=False)
        torch.cuda.synchronize()
        ta = time.perf_counter()
        run_steps(40, keep=False)
        torch.cuda.synchronize()
        analysis_only = 40 * nf / (time.perf_counter() - ta)
        gen_in_loop[0] = True
        run_steps(3, keep=False)
    ctx.get_timing(reset=True)
    ctx.set_timing(True, kinds=None if args.time_all_kernels else (dom,) + (("synth",) if streaming else ()))
    sync()
    t0 = time.perf_counter()
    run_steps(args.steps, keep=True)
    if world > 1:
        # the one collective of the job: every rank's clips (video id = step*world + rank) in video order
        all_rows = gather_rows(rows, args.steps * world, nf, device=gather_device)
    sync()
    dt = time.perf_counter() - t0
    ctx.set_timing(False)
    timing = ctx.get_timing()
    if world > 1:
        tmax = torch.tensor([dt], dtype=torch.float64, device=gather_device or "cpu")
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
    else:
        all_rows = rows

    if rank == 0:
        frames_total = world * args.steps * nf
        fps = frames_total / dt
        ab = algorithmic_bytes(h, w, fh, fw)
        kt = timing["kernels"]
        # frames per launch of the dominant kernel in the timed region (it runs once per chunk; the generic LBP and
        # hysteresis families count several launches per chunk, which the solo clip tells)
        per_chunk = max(solo["kernels"][dom]["launches"] // solo_chunks, 1)
        chunks = max(kt[dom]["launches"] // per_chunk, 1)
        frames_per_launch = timing["frames"] / chunks
        dom_ms = kt[dom]["ms"] / chunks
        dom_gbs = kab[dom] * frames_per_launch / (dom_ms * 1e-3) / 1e9
        dom_gbs_solo = kab[dom] * (solo["frames"] / solo_chunks) / (solo_kernel_ms[dom] * 1e-3) / 1e9
        # device time of each kernel group per chunk in the timed region (sum of its kernels' HIP-event durations):
        # known only when every kernel carried events
        group_ms = {g: (sum(kt[k]["ms"] for k in ks) / chunks if args.time_all_kernels else 0.0)
                    for g, ks in GROUPS.items()}
        solo_fpl = solo["frames"] / solo_chunks
        groups = {}
        for g in GROUPS:
            if solo_ms[g] <= 0 or ab[g] == 0:
                continue
            ach = ab[g] * frames_per_launch / (group_ms[g] * 1e-3) / 1e9 if group_ms[g] > 0 else None
            ach_solo = ab[g] * solo_fpl / (solo_ms[g] * 1e-3) / 1e9
            groups[g] = {"kernels": "+".join(k for k in GROUPS[g] if solo["kernels"][k]["launches"]),
                         "algorithmic_bytes_per_frame": ab[g], "avg_launch_ms": group_ms[g] if ach else None,
                         "achieved": ach, "frac": ach / HBM_PEAK_GBS if ach else None,
                         "avg_launch_ms_serialized": solo_ms[g], "frac_serialized": ach_solo / HBM_PEAK_GBS}
        path_gbs = fps / world * ab["path"] / 1e9
        mt = measured_traffic(h, w)
        traffic = None
        if mt is not None:
            traffic = mt[dom]["total"] * frames_per_launch if dom in mt else None
        result = {
            "metric": "texture-analysis frames/sec (whole node) at 1080p + achieved HBM GB/s % of peak",
            "value": fps,
            "unit": "frames/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u8 -> packed u16 stencils, 16-bit integer LBP decision with f64 tie-breaks, f32 FFT "
                     "(f64 for frames with near-zero spectrum bins), f64 sums",
            "data": "synthetic",
            "world_size_seen": world,
            "backend": ("rccl (torch nccl)" if backend == "nccl" else backend) if world > 1 else "none (one rank)",
            "config": {"workload": f"{args.config}: {args.kind} frames, {nf} x {w}x{h} BGR u8 per step per GPU, "
                                   f"FFT extent {fw}x{fh}"
                                   + (", LBP at (8,1) (16,2) (24,3) + Sobel / blockiness sums" if cfg["extended"] else "")
                                   + f" (BASELINE {cfg['row']})",
                       "frames_per_step_per_gpu": nf, "height": h, "width": w, "sharding": "by clip, one rank per GPU",
                       "distinct_clips_per_gpu": args.steps if streaming else n_clips, "video_id": "step * world + rank",
                       "clips_resident_per_gpu": n_clips,
                       "submission": "blocking" if not pipelined else "two clips in flight (stx_submit_batch_device / stx_wait)",
                       "untimed_steps_before_warmup": PREWARM_STEPS},
            "roofline": {
                "bound": "hbm", "kernel": dom,
                "achieved": dom_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": dom_gbs / HBM_PEAK_GBS, "traffic": traffic,
                "frac_serialized": dom_gbs_solo / HBM_PEAK_GBS,
                "algorithmic_bytes_per_frame": kab[dom], "frames_per_launch": frames_per_launch,
                "avg_launch_ms": dom_ms,
                "avg_launch_ms_serialized": solo_kernel_ms[dom],
                "note": "dominant kernel = the single kernel with the longest serialized duration (one clip before "
                        "the timed region, every kernel alone on the device with an event pair around it: "
                        "*_serialized); in the timed region only that kernel carries events (--time-all-kernels: "
                        "all), on the stream it is launched on, and its durations include whatever runs beside it "
                        "on the other stream; 'groups' and 'path' give the same account for kernel groups "
                        "(serialized) and for the whole job",
                "kernels_serialized": {k: {"ms_per_launch_group": v, "algorithmic_bytes_per_frame": kab[k],
                                           "frac": kab[k] * (solo["frames"] / solo_chunks) / (v * 1e-3) / 1e9 / HBM_PEAK_GBS}
                                       for k, v in solo_kernel_ms.items()},
                "groups": groups,
                "path": {"algorithmic_bytes_per_frame": ab["path"], "achieved": path_gbs,
                         "frac": path_gbs / HBM_PEAK_GBS},
            },
            "kernel_ms_per_step": {k: v["ms"] / args.steps for k, v in timing["kernels"].items() if v["launches"]},
            "kernel_ms_per_step_serialized": {k: v["ms"] for k, v in solo["kernels"].items() if v["launches"]},
            "fft64_frames_in_timed_region": timing["fft64_frames"],
            "mean_texture_score": float(all_rows[..., 5].mean()),
        }
        if streaming:
            result["config"]["input"] = ("generated on the device inside the timed region, clip i beside the analysis "
                                         "of clip i-1 (stx_synth_device_async); 'value' includes the generation")
            synth_ms = kt["synth"]["ms"] / max(kt["synth"]["launches"], 1)
            result["streaming"] = {
                "clips_total": world * args.steps,
                "frames_per_s_generation_included": fps,
                "frames_per_s_analysis_only_rank0": analysis_only,
                "synth_kernel_ms_per_clip_in_situ": synth_ms,
                "synth_store_GBps": clip_bytes / (synth_ms * 1e-3) / 1e9 if synth_ms > 0 else None,
            }
            if not args.no_cpu_baseline:
                result["streaming"]["spot_check"] = spot_check(all_rows, args.kind, seed, h, w, nf)
        elif world == 1 and not args.no_cpu_baseline:
            sample = min(nf, 150)
            frames = ctx.d2h(clips[0].data_ptr(), sample * h * w * 3).reshape(sample, h, w, 3)
            result["cpu_baseline"] = cpu_baseline(frames, rows[0], args.cpu_seconds * (0.7 if args.config == "1080p" else 1.0),
                                                  params)
            result["parity_ok"] = result["cpu_baseline"]["max_rel_diff_vs_gpu"] <= 1e-4
            if args.config == "1080p":
                result["config1_cpu_reference_path"] = config1(ctx, args.cpu_seconds * 0.3)
        if world == 1 and args.host_fed:
            result["host_fed"] = host_fed(ctx, clips[0], nf, h, w)
        if world == 1 and args.config == "1080p" and args.kind == "natural" and not args.no_other_content:
            result["other_content"] = other_content(ctx, clips, run_steps, nf, fps)  # (overwrites the clips: last)
        print(json.dumps(result), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    ctx.close()


def other_content(ctx, clips, run_steps, nf: int, natural_fps: float) -> dict:
    """Frames/s of the same pipeline on the other synthetic content classes (a few clips each, after the headline's
    timed region): how far the rate depends on what the frames show -- exact LBP ties (flat, tie), dense edges (noise,
    step), spectra with exactly-zero bins that send every frame through the fp64 transform (ramp)."""
    import torch

    from sift_amd.synth import KIND_ID, KINDS

    out = {"natural": natural_fps}
    ctx.get_timing(reset=True)
    for kind in KINDS:
        if kind == "natural":
            continue
        ctx.set_canny_mode("auto")  # forget what the class before taught the adaptive placement of Canny's suppression
        for i, t in enumerate(clips[:8]):  # the clips the steps below go through
            ctx.synth_device(t.data_ptr(), KIND_ID[kind], 0x5EED0000 + 2, i, 0, nf, *_shape)
        run_steps(4, False)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        run_steps(8, False)
        torch.cuda.synchronize()
        out[kind] = 8 * nf / (time.perf_counter() - t0)
    out["fft64_frames"] = ctx.get_timing(reset=True)["fft64_frames"]
    out["slowest_over_fastest"] = min(v for k, v in out.items() if k in KINDS) / max(v for k, v in out.items() if k in KINDS)
    return out


def spot_check(all_rows: np.ndarray, kind: str, seed: int, h: int, w: int, nf: int) -> dict:
    """Rows of a few late video ids of a streaming run against frames regenerated on the host (sift_amd.synth, the
    generator's numpy form) and analysed by the C oracle: the clip the GPU analysed at step i is the clip of video i."""
    from oracle import texture_oracle as O
    from sift_amd.synth import synth_frame

    co = O.COracle()
    n_videos = all_rows.shape[0]
    vids = sorted({v for v in (51, 52 + (n_videos - 52) // 2, n_videos - 1) if 0 <= v < n_videos})
    if not vids:
        vids = [n_videos - 1]
    worst, checked = 0.0, []
    for v in vids:
        for f in sorted({0, nf // 2, nf - 1}):
            m = co.analyze_frame(synth_frame(kind, seed, v, f, h, w))
            ref = np.array([m.lbp_variance, m.noise_level, m.compression_artifacts, m.edge_density, m.frequency_anomaly,
                            m.texture_score])
            rel = np.abs(all_rows[v, f] - ref) / np.maximum(np.abs(ref), 1e-12)
            worst = max(worst, float(rel.max()))
            checked.append([int(v), int(f)])
    return {"video_frame_pairs": checked, "max_rel_diff_vs_oracle": worst, "ok": worst <= 1e-4}


def dry_run(args) -> int:
    """STX_BENCH_DRYRUN=1: the launch, sharding and gather of a multi-rank run without any GPU work (rows are a
    function of the video id), over gloo -- what a CPU-only machine can check of --gpus N."""
    import torch.distributed as dist

    from sift_amd.sharding import gather_rows

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if world > 1:
        dist.init_process_group("gloo")
    if args.steps <= 0:
        args.steps = 40
    nf = args.frames or 4
    rows = np.zeros((args.steps, nf, 6), np.float64)
    for i in range(args.steps):
        rows[i, :, :] = i * world + rank  # video id of step i on this rank
    all_rows = gather_rows(rows, args.steps * world, nf) if world > 1 else rows
    ok = all(np.all(all_rows[v] == v) for v in range(args.steps * world))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps({"dry_run": True, "n_gpus": world, "world_size_seen": world, "backend": "gloo",
                          "steps": args.steps, "videos_in_order": bool(ok),
                          "mean_texture_score": float(all_rows[..., 5].mean())}), flush=True)
    return 0 if ok else 1


def host_fed(ctx, clip, nf: int, h: int, w: int) -> dict:
    """The same clip fed from page-locked host memory (sift_amd.frames.PinnedFrameBatch): host-to-device copy
    included, so this is bounded by PCIe, not by the kernels."""
    from sift_amd.frames import PinnedFrameBatch

    batch = PinnedFrameBatch(ctx, nf, h, w)
    batch.array[...] = ctx.d2h(clip.data_ptr(), nf * h * w * 3).reshape(nf, h, w, 3)
    batch.count = nf
    batch.analyze()
    t0 = time.perf_counter()
    reps = 3
    for _ in range(reps):
        batch.analyze()
    dt = (time.perf_counter() - t0) / reps
    batch.close()
    return {"frames_per_s": nf / dt, "GB_per_s": nf * h * w * 3 / dt / 1e9, "memory": "pinned", "ms_per_step": dt * 1e3}


def host_cpu() -> dict:
    model = "unknown"
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    model = line.split(":", 1)[1].strip()
                    break
    except OSError:
        pass
    total = os.cpu_count() or 1
    try:
        usable = len(os.sched_getaffinity(0))
    except AttributeError:
        usable = total
    # the job's CPU share (a one-GPU slice of the host), not every CPU the machine has
    share = int(os.environ.get("OMP_NUM_THREADS", "0") or 0)
    workers = max(1, min(usable, share) if share > 0 else min(usable, 16))
    return {"model": model, "cores_total": total, "cores_usable": usable, "workers": workers}


def _oracle_rate(co, frames, params, workers: int, budget_s: float):
    """frames/s of the C oracle over as many of `frames` as fit the time budget, `workers` threads each taking
    whole frames (the C code holds no shared state and ctypes releases the interpreter lock during the call)."""
    from concurrent.futures import ThreadPoolExecutor

    op = None
    if params is not None:
        op = co.default_params()
        op.fft_h, op.fft_w = params.fft_h, params.fft_w
    results = {}
    deadline = time.perf_counter() + budget_s

    def work(i):
        if time.perf_counter() > deadline:
            return
        results[i] = co.analyze_frame(frames[i], op)

    t0 = time.perf_counter()
    if workers == 1:
        for i in range(len(frames)):
            work(i)
    else:
        with ThreadPoolExecutor(workers) as ex:
            list(ex.map(work, range(len(frames))))
    dt = time.perf_counter() - t0
    return len(results) / dt, results, dt


def library_rate(frames, budget_s: float):
    """The reference's own library calls (passive_checker.py:100-123: cv2 + scikit-image + numpy), timed from this
    script when those libraries are importable on the host; None otherwise."""
    try:
        import cv2
        from skimage import feature
    except Exception:
        return None
    kernel = np.array([[-1, -1, -1], [-1, 8, -1], [-1, -1, -1]])
    t0 = time.perf_counter()
    n = 0
    for fr in frames:
        gray = cv2.cvtColor(fr, cv2.COLOR_BGR2GRAY)
        np.var(feature.local_binary_pattern(gray, 8, 1, method="uniform"))
        np.std(cv2.filter2D(gray, -1, kernel))
        np.mean(cv2.absdiff(gray, cv2.GaussianBlur(gray, (5, 5), 0)))
        e = cv2.Canny(gray, 50, 150)
        np.sum(e > 0) / e.size
        np.std(np.log(np.abs(np.fft.fftshift(np.fft.fft2(gray))) + 1))
        n += 1
        if time.perf_counter() - t0 > budget_s:
            break
    return n / (time.perf_counter() - t0), n


def cpu_baseline(frames: np.ndarray, gpu_rows: np.ndarray, budget_s: float, params) -> dict:
    """The C oracle (a port of the reference's CPU path) on the first frames of the same clip: one thread, then
    one thread per core of this job's CPU share, each leg until its share of the time budget is used.  Also
    re-checks the GPU rows against it."""
    from oracle import texture_oracle as O

    co = O.COracle()
    cpu = host_cpu()
    n, h, w, _ = frames.shape
    r1, res1, t1 = _oracle_rate(co, frames, params, 1, budget_s * 0.4)
    rn, resn, tn = _oracle_rate(co, frames, params, cpu["workers"], budget_s * 0.45)
    max_rel = 0.0
    for res in (res1, resn):
        for i, m in res.items():
            ref = np.array([m.lbp_variance, m.noise_level, m.compression_artifacts, m.edge_density, m.frequency_anomaly,
                            m.texture_score])
            rel = np.abs(gpu_rows[i] - ref) / np.maximum(np.abs(ref), 1e-12)
            max_rel = max(max_rel, float(rel.max()))
    out = {"value": rn, "unit": "frames/s", "cores": cpu["workers"], "kind": "port",
           "sample": f"first {len(resn)} frames of the same clip ({w}x{h}) in {tn:.1f} s on {cpu['workers']} threads, "
                     f"first {len(res1)} frames in {t1:.1f} s on one thread; C oracle (oracle/texture_oracle.c)",
           "one_thread": {"value": r1, "cores": 1, "frames": len(res1)},
           "cpu_model": cpu["model"], "cores_total": cpu["cores_total"], "cores_usable": cpu["cores_usable"],
           "max_rel_diff_vs_gpu": max_rel}
    lib = library_rate(frames, budget_s * 0.15)
    out["library"] = ({"value": lib[0], "cores": 1, "kind": "library", "frames": lib[1],
                       "sample": "cv2 + scikit-image + numpy calls of passive_checker.py:100-123, one process"}
                      if lib else None)
    return out


def config1(ctx, budget_s: float) -> dict:
    """BASELINE configs[0]: one synthetic 720p clip of 150 frames plus 150 crops of 80x80 taken at the moving centre,
    on the CPU path (C oracle, all threads of the job's share; one thread for the first-ten call), with the GPU's
    time for the same calls beside it.  'first ten' is what compute_texture_score analyses (passive_checker.py:140)."""
    from oracle import texture_oracle as O
    from sift_amd.synth import synth_frame

    co = O.COracle()
    cpu = host_cpu()
    h, w, n = 720, 1280, 150
    seed = 0x5EED0000
    t0 = time.perf_counter()
    dev = ctx.device_alloc(n * h * w * 3)
    try:
        ctx.synth_device(dev, 1, seed, 0, 0, n, h, w)
        clip = ctx.d2h(dev, n * h * w * 3).reshape(n, h, w, 3)
        ctx.analyze_device(dev, n, h, w)
        tg = time.perf_counter()
        ctx.analyze_device(dev, n, h, w)
        gpu_all = n / (time.perf_counter() - tg)
        tg = time.perf_counter()
        ctx.analyze_device(dev, 10, h, w)
        gpu_first10_ms = (time.perf_counter() - tg) * 1e3
    finally:
        ctx.device_free(dev)
    crops = []
    for f in range(n):
        cy, cx = h // 2 + 2 * f, w // 2 + 3 * f
        y0, x0 = min(max(cy - 40, 0), h - 80), min(max(cx - 40, 0), w - 80)
        crops.append(np.ascontiguousarray(clip[f, y0:y0 + 80, x0:x0 + 80]))
    crops = np.stack(crops)
    ctx.analyze_batch(crops)
    tg = time.perf_counter()
    ctx.analyze_batch(crops)
    gpu_crops = n / (time.perf_counter() - tg)
    r_all, res, t_all = _oracle_rate(co, clip, None, cpu["workers"], budget_s * 0.6)
    r10, _, t10 = _oracle_rate(co, clip[:10], None, 1, budget_s * 0.3)
    rc, resc, tc = _oracle_rate(co, crops, None, cpu["workers"], budget_s * 0.1)
    return {"clip": f"{n} frames {w}x{h} natural + {n} crops 80x80 at the moving centre",
            "cpu": {"kind": "port", "threads": cpu["workers"], "cpu_model": cpu["model"],
                    "frames_720p_per_s_all_frames": r_all, "frames_720p_done": len(res),
                    "first10_720p_ms_one_thread": t10 * 1e3, "first10_frames_per_s_one_thread": r10,
                    "crops_80x80_per_s": rc, "crops_done": len(resc)},
            "gpu": {"frames_720p_per_s_all_frames_device_resident": gpu_all, "first10_720p_ms_device_resident": gpu_first10_ms,
                    "crops_80x80_per_s_host_fed": gpu_crops},
            "seconds": time.perf_counter() - t0}


if __name__ == "__main__":
    main()

"""One 150-frame 1080p clip with every kernel alone on the device, under rocprofv3 --kernel-trace: prints each
dispatch of the last clip in order with its duration.  Run on the GPU box from the repository root:
  python3 tools/serial_trace.py run     (the traced program)
  python3 tools/serial_trace.py show <kernel_trace.csv>"""
import csv
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

if sys.argv[1] == "run":
    import torch
    from sift_amd.synth import KIND_ID
    from sift_amd.texture import TextureContext
    ctx = TextureContext(0)
    nf, h, w = 150, 1080, 1920
    kind = sys.argv[2] if len(sys.argv) > 2 else "natural"
    t = torch.empty(nf * h * w * 3, dtype=torch.uint8, device="cuda")
    ctx.synth_device(t.data_ptr(), KIND_ID[kind], 0x5EED0002, 0, 0, nf, h, w)
    ctx.set_concurrency(False)
    for _ in range(4):
        ctx.analyze_device(t.data_ptr(), nf, h, w)
    torch.cuda.synchronize()
else:
    rows = list(csv.DictReader(open(sys.argv[2])))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    # the last clip starts at the last stencil dispatch
    last = max(i for i, r in enumerate(rows) if "stencil_kernel" in r["Kernel_Name"])
    t0 = int(rows[last]["Start_Timestamp"])
    tot = 0.0
    for r in rows[last:]:
        d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        tot += d
        print(f"{(int(r['Start_Timestamp']) - t0) / 1e3:9.1f} {d:8.1f}  {r['Kernel_Name'][:60]}")
    print("sum of durations", round(tot, 1), "us")

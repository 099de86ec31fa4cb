"""Per-kernel HBM traffic from two rocprofv3 --pmc passes (FETCH_SIZE and WRITE_SIZE; they do not fit in
one pass) over tools/prof1.py, which analyses FRAMES 1080p frames per dispatch.

    python profiles/traffic_from_pmc.py <fetch_counter_collection.csv> <write_counter_collection.csv> <frames> <out.json>

Both counters come in KiB.  On gfx950 FETCH_SIZE tallies a 128-byte request as 64 bytes, so the reads of
kernels whose loads are 12 bytes per lane or wider (gray, the stencil pass that reads BGR frames, FFT rows
prefetch aside, FFT cols) are doubled; narrow-load kernels (hysteresis and the suppression on the flooded set, which
is counted with it, FFT rows byte/halfword loads) are left as counted, so their true read traffic lies between the raw
and the doubled figure.  "clears" are the chunk's one init kernel (accumulators, flags) and whatever fill kernels the runtime still launches.
"""
import collections
import csv
import json
import sys

# (substring of the kernel name, group, loads are wide); the first match counts.  stencil_kernel<.., .., true> reads
# the BGR frames with 12-byte loads, stencil_kernel<.., .., false> the gray plane with 16-byte loads
GROUPS = (("gray_kernel", "gray", True), ("stencil_kernel", "stencil", True), ("lbp_generic", "lbp_generic", False),
          ("hyst_pass_kernel", "hysteresis", False), ("hyst_finish_kernel", "hysteresis", False),
          ("nms_collect_kernel", "hysteresis", False), ("nms_apply_kernel", "hysteresis", False),
          ("fillBufferAligned", "clears", False), ("chunk_init", "clears", False), ("rows_kernel", "fft_rows", False), ("cols_kernel", "fft_cols", True),
          ("frame_reduce", "reduce", False))


def per_kernel(path, counter):
    tot = collections.defaultdict(float)
    disp = collections.defaultdict(set)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        for sub, name, _ in GROUPS:
            if sub in r["Kernel_Name"]:
                tot[name] += float(r["Counter_Value"])
                disp[name].add(r["Dispatch_Id"])
                break
    return tot, disp


def main():
    fetch_csv, write_csv, frames, out = sys.argv[1], sys.argv[2], int(sys.argv[3]), sys.argv[4]
    rd, rdisp = per_kernel(fetch_csv, "FETCH_SIZE")
    wr, _ = per_kernel(write_csv, "WRITE_SIZE")
    res = {}
    path_total = 0.0
    for _, name, wide in GROUPS:
        if name not in rd and name not in wr:
            continue
        # the profiled script runs the clip twice: divide by the frames all dispatches of a kernel covered
        runs = 2
        r = rd.get(name, 0.0) * 1024.0 * (2.0 if wide else 1.0) / (frames * runs)
        w = wr.get(name, 0.0) * 1024.0 / (frames * runs)
        res[name] = {"read": r, "write": w, "total": r + w, "read_doubled": wide}
        path_total += r + w
    res["path"] = {"total": path_total}
    json.dump({"note": __doc__.strip().split("\n\n")[-1].replace("\n", " "), "height": 1080, "width": 1920,
               "frames_per_dispatch": frames, "per_frame_bytes": res}, open(out, "w"), indent=1)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()

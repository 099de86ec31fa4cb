"""Timeline of one rocprofv3 --kernel-trace run of bench.py: for the last clips of the timed region, when each kernel ran,
what ran beside it, and how busy the device was.

    python tools/timeline.py <kernel_trace.csv> [clips]
"""
import csv
import sys

NAMES = (("stencil", "stencil"), ("gray_kernel", "gray"), ("rows_kernel", "fft_rows"), ("cols_kernel", "fft_cols"),
         ("hyst_pass", "hysteresis"), ("hyst_finish", "hysteresis"), ("nms_collect", "suppression"), ("nms_apply", "suppression"), ("frame_reduce", "reduce"), ("chunk_init", "init"), ("synth", "synth"),
         ("copyBuffer", "copy"), ("fillBuffer", "fill"), ("fft64", "fft64"))


def short(n):
    for k, v in NAMES:
        if k in n:
            return v
    return n[:24]


def main():
    rows = list(csv.DictReader(open(sys.argv[1])))
    clips = int(sys.argv[2]) if len(sys.argv) > 2 else 3
    ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"]), r["Queue_Id"]) for r in rows)
    gray = [e for e in ev if e[2] == "gray"]
    if not gray:  # the stencil pass reads the frames itself: a clip starts with it
        gray = [e for e in ev if e[2] == "stencil"]
    # the timed clips are the last ones bench.py runs (its serialized clips come before the warm-up): the window
    # runs from the start of the clip `clips` + 1 from the end to the start of the last one
    first, last = gray[-1 - clips][0], gray[-1][0]
    sel = [e for e in ev if first <= e[0] < last]
    t0 = sel[0][0]
    print("concurrent clips (times in us from the first kernel of a clip)")
    for s, e, n, q in sel:
        if n in ("fill", "copy", "reduce", "init"):
            continue
        beside = sorted({m for (s2, e2, m, _) in sel if m != n and s2 < e and e2 > s and m not in ("fill", "copy")})
        print(f"  {(s - t0) / 1e3:9.1f} {(e - t0) / 1e3:9.1f} {(e - s) / 1e3:8.1f}  q{q} {n:10s} beside: {', '.join(beside) or '-'}")
    span = (last - first) / 1e3
    busy = 0
    cur_s, cur_e = None, None
    for s, e, n, q in sel:
        if cur_e is None or s > cur_e:
            if cur_e is not None:
                busy += cur_e - cur_s
            cur_s, cur_e = s, e
        else:
            cur_e = max(cur_e, e)
    busy += cur_e - cur_s
    tot = {}
    for s, e, n, q in sel:
        tot[n] = tot.get(n, 0) + (e - s) / 1e3
    print(f"per clip: {span / clips:.1f} us wall, {busy / 1e3 / clips:.1f} us with at least one kernel running, "
          f"{(span - busy / 1e3) / clips:.1f} us idle (host between blocking calls)")
    print("kernel time per clip (us):", {k: round(v / clips, 1) for k, v in sorted(tot.items())})


if __name__ == "__main__":
    main()

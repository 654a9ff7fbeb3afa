#!/usr/bin/env python3
"""Per-(kernel, grid size) summary of a rocprofv3 --kernel-trace CSV: separates the levels of the
hierarchy (each level launches a kernel with its own grid size) so the time of an iteration can be
attributed level by level.  Usage: trace_levels.py <kernel_trace.csv> [iterations]"""
import csv
import collections
import re
import sys


def main():
    path = sys.argv[1]
    iters = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
    agg = collections.defaultdict(lambda: [0, 0.0])
    tot = 0.0
    t_first, t_last = None, None
    for r in csv.DictReader(open(path)):
        name = r["Kernel_Name"]
        m = re.search(r"(\w+_kernel)(<[^>]*>)?", name)
        short = (m.group(1) + (m.group(2) or "")) if m else name[:40]
        grid = int(r.get("Grid_Size_X", r.get("Grid_Size", 0)))
        wg = int(r.get("Workgroup_Size_X", r.get("Workgroup_Size", 1)) or 1)
        d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-3
        key = (short, grid // max(wg, 1))
        agg[key][0] += 1
        agg[key][1] += d
        tot += d
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        t_first = s if t_first is None else min(t_first, s)
        t_last = e if t_last is None else max(t_last, e)
    print(f"kernel time total {tot / 1e3:.3f} ms over {sum(v[0] for v in agg.values())} dispatches; span {(t_last - t_first) * 1e-6:.3f} ms; "
          f"per iteration ({iters:g}): {tot / iters:.1f} us busy")
    print(f"{'kernel':48s} {'wgs':>8s} {'calls':>7s} {'avg us':>9s} {'us/iter':>9s} {'%':>6s}")
    for (k, g), (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print(f"{k[:48]:48s} {g:8d} {c:7d} {t / c:9.2f} {t / iters:9.1f} {100 * t / tot:6.2f}")


if __name__ == "__main__":
    main()


def runs(path, needle="sdia_tab_kernel<2, false, 1>"):
    """Durations of the finest-level sweep kernel by position inside a run of consecutive launches of it."""
    rows = sorted(csv.DictReader(open(path)), key=lambda r: int(r["Start_Timestamp"]))
    pos, bypos, gaps = 0, collections.defaultdict(list), collections.defaultdict(list)
    prev_end = None
    for r in rows:
        if needle in r["Kernel_Name"]:
            d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-3
            bypos[pos].append(d)
            if prev_end is not None:
                gaps[pos].append((int(r["Start_Timestamp"]) - prev_end) * 1e-3)
            pos += 1
        else:
            pos = 0
        prev_end = int(r["End_Timestamp"])
    print(f"position-in-run statistics of {needle}:")
    for p in sorted(bypos):
        v = sorted(bypos[p])
        g = sorted(gaps[p]) or [0.0]
        print(f"  pos {p}: n={len(v):5d} median {v[len(v) // 2]:7.2f} us  min {v[0]:7.2f}  max {v[-1]:7.2f}   gap before (median) {g[len(g) // 2]:6.2f} us")


if __name__ == "__main__" and len(sys.argv) > 3 and sys.argv[3] == "runs":
    runs(sys.argv[1], sys.argv[4] if len(sys.argv) > 4 else "sdia_tab_kernel<2, false, 1>")

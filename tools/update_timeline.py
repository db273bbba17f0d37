#!/usr/bin/env python
"""Kernel-by-kernel timeline of ONE update inside a rocprofv3 --kernel-trace CSV: the stretch
between two consecutive occurrences of a marker kernel (default: the fused Adam step) near the
end of the trace.  Prints start offset, gap to the previous kernel's end, duration, grid.
Usage: python tools/update_timeline.py <kernel_trace.csv> [--marker FusedAdam] [--every 3]"""
import argparse, csv, re


def short(n):
    n = re.sub(r"\(anonymous namespace\)::|at::native::|void ", "", n)
    return n.split("(")[0][:80]


ap = argparse.ArgumentParser()
ap.add_argument("csv")
ap.add_argument("--marker", default="FusedAdam")
ap.add_argument("--every", type=int, default=3, help="marker launches per update")
ap.add_argument("--skip", type=int, default=0, help="updates to skip at the end of the trace (e.g. bench.py's "
                "eager, event-bracketed updates of mfma.per_launch: 6)")
a = ap.parse_args()
rows = [r for r in csv.DictReader(open(a.csv))]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if a.marker in r["Kernel_Name"]]
if len(idx) <= (2 + a.skip) * a.every:
    # (too few marker launches for this --every: say what the trace holds instead of a traceback)
    import collections

    names = collections.Counter(short(r["Kernel_Name"]) for r in rows)
    print("marker %r found %d times (need > %d); most frequent kernels:" % (a.marker, len(idx), 2 * a.every))
    for n, c in names.most_common(25):
        print("%8d  %s" % (c, n))
    raise SystemExit(0)
i1 = idx[-1 - a.every * (1 + a.skip)]
i0 = idx[-1 - a.every * (2 + a.skip)]
t0 = int(rows[i0]["End_Timestamp"])
prev = t0
busy = 0
for r in rows[i0 + 1:i1 + 1]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print("%8.1f gap %6.1f dur %6.1f  grid %6s  %s" % ((s - t0) / 1e3, (s - prev) / 1e3, (e - s) / 1e3,
                                                     r["Grid_Size_X"] if "Grid_Size_X" in r else "?", short(r["Kernel_Name"])))
    busy += e - s
    prev = max(prev, e)
print("kernels %d, span %.1f us, sum of durations %.1f us" % (i1 - i0, (prev - t0) / 1e3, busy / 1e3))

#!/usr/bin/env python
"""Summarise a rocprofv3 --kernel-trace CSV: for the last `--window-ms` of the trace, kernel
time by name, launch count, and the share of the window in which no kernel was running
(launch gaps).  Usage: python tools/trace_summary.py <kernel_trace.csv> [--window-ms 200]"""
import argparse
import csv
import re
from collections import defaultdict


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    return name.split("(")[0][:90]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("csv")
    ap.add_argument("--window-ms", type=float, default=200.0)
    ap.add_argument("--top", type=int, default=40)
    a = ap.parse_args()
    rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"])
            for r in csv.DictReader(open(a.csv))]
    rows.sort()
    t_end = max(r[1] for r in rows)
    t0 = t_end - int(a.window_ms * 1e6)
    rows = [r for r in rows if r[0] >= t0]
    busy = 0
    cur_s, cur_e = rows[0][0], rows[0][1]
    for s, e, _ in rows[1:]:
        if s > cur_e:
            busy += cur_e - cur_s
            cur_s, cur_e = s, e
        else:
            cur_e = max(cur_e, e)
    busy += cur_e - cur_s
    span = rows[-1][1] - rows[0][0]
    tot = defaultdict(lambda: [0, 0])
    for s, e, n in rows:
        k = short(n)
        tot[k][0] += e - s
        tot[k][1] += 1
    print("window %.2f ms, %d launches, kernels busy %.2f ms (%.1f %%), sum of kernel durations %.2f ms"
          % (span / 1e6, len(rows), busy / 1e6, 100.0 * busy / span,
             sum(v[0] for v in tot.values()) / 1e6))
    # where the GPU idles: gaps between consecutive kernels, by (kernel before -> kernel after)
    gaps = defaultdict(lambda: [0, 0])
    prev_e, prev_n = rows[0][1], short(rows[0][2])
    for s_, e_, n_ in rows[1:]:
        if s_ > prev_e + 3000:
            key = "%s -> %s" % (prev_n[:40], short(n_)[:40])
            gaps[key][0] += s_ - prev_e
            gaps[key][1] += 1
        if e_ > prev_e:
            prev_e, prev_n = e_, short(n_)
    print("idle gaps > 3 us: %.2f ms in %d gaps" % (sum(v[0] for v in gaps.values()) / 1e6,
                                                    sum(v[1] for v in gaps.values())))
    for k, (ns, c) in sorted(gaps.items(), key=lambda kv: -kv[1][0])[:12]:
        print("   %8.3f ms %5d x %8.1f us  %s" % (ns / 1e6, c, ns / c / 1e3, k))
    for k, (ns, c) in sorted(tot.items(), key=lambda kv: -kv[1][0])[:a.top]:
        print("%8.3f ms %6d x %8.2f us  %5.1f %%  %s" % (ns / 1e6, c, ns / c / 1e3, 100.0 * ns / span, k))


if __name__ == "__main__":
    main()

#!/usr/bin/env python
"""Print every kernel of the last `--ms` of a rocprofv3 kernel trace, one line each, collapsing
runs of the per-update kernels.  python tools/trace_slice.py trace.csv --ms 12"""
import argparse, csv, re
ap = argparse.ArgumentParser(); ap.add_argument("csv"); ap.add_argument("--ms", type=float, default=10.0)
ap.add_argument("--back-ms", type=float, default=None, help="the slice ENDS this many ms before the end of the trace")
ap.add_argument("--no-collapse", action="store_true", help="print the per-update kernels too")
a = ap.parse_args()
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(a.csv))]
rows.sort()
t_end = rows[-1][1] - int((a.back_ms or 0.0) * 1e6); t0 = t_end - int(a.ms * 1e6)
rows = [r for r in rows if t0 <= r[0] <= t_end]
def short(n):
    n = re.sub(r"\(anonymous namespace\)::|at::native::|void ", "", n); return n.split("(")[0][:60]
upd = () if a.no_collapse else ("k_conv_bwd", "k_conv_fwd<16", "k_conv_fwd<32", "k_rmsprop", "k_conv_wgrad", "k_splitk_reduce", "k_dqn_head_td")
prev = rows[0][0]; run = 0; run_t = 0
for s, e, n in rows:
    sn = short(n)
    if sn.startswith(upd):
        run += 1; run_t += e - s; prev = e; continue
    if run:
        print("        ... %d update kernels, %.1f us busy" % (run, run_t / 1e3)); run = 0; run_t = 0
    print("%9.1f gap %7.1f dur %7.1f  %s" % ((s - t0) / 1e3, (s - prev) / 1e3, (e - s) / 1e3, sn))
    prev = e

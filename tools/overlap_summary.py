#!/usr/bin/env python3
"""How much of the small kernels' run time (PlotUnit::plot, GatherUnit::accumulate, ...) fell inside the run time of a
trace kernel, from a rocprofv3 --kernel-trace CSV of an App run.  Usage: overlap_summary.py <..._kernel_trace.csv>"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
ev = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows]
trace = [(s, e) for s, e, n in ev if "rl_trace_kernel" in n]
t0, t1 = min(s for s, e, n in ev), max(e for s, e, n in ev)
print("trace kernels: %d launches, running %.1f %% of the %.1f ms between the first and the last kernel of the run"
      % (len(trace), 100.0 * sum(e - s for s, e in trace) / (t1 - t0), (t1 - t0) / 1e6))
for kind in ("rl_plot_kernel", "rl_gather_kernel", "rl_add_kernel", "rl_exposure_kernel", "rl_tonemap_kernel", "fillBuffer"):
    ks = [(s, e) for s, e, n in ev if kind in n]
    if not ks:
        continue
    total = sum(e - s for s, e in ks)
    inside = sum(max(0, min(e, te) - max(s, ts)) for s, e in ks for ts, te in trace)
    print("%-20s %4d launches, %9.1f us in total, %5.1f %% of it while a trace kernel was running" % (kind, len(ks), total / 1e3, 100.0 * inside / total))

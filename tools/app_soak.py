#!/usr/bin/env python3
"""A longer rl_app_run at the reference task size (16 workers, tonemap every 5 s): 300,000 batches un-fused, 200,000 fused.
Checks that every path is traced, prints the sustained rate and the process RSS after each run (no growth between runs)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import robigo_luculenta_amd as R  # noqa: E402


def rss():
    return int(open("/proc/self/statm").read().split()[1]) * os.sysconf("SC_PAGE_SIZE") / 1e6
print("rss before %.0f MB" % rss())
for fused, n in ((False, 300000), (True, 200000)):
    t0 = time.time()
    rgb, st = R.app_run(1280, 720, n, concurrency=16, photons_per_batch=524288, fused=fused, tonemap_interval_ms=5000, verbose=False)
    print("fused" if fused else "un-fused", "%d batches in %.1f s, %.0f Mrays/s, paths ok %s, tasks %s, rss %.0f MB"
          % (n, st["seconds"], st["segments"] / st["seconds"] / 1e6, st["paths"] == n * 524288, st["tasks"], rss()), flush=True)
h = R.batch_histogram()
print("launches", sum(h.values()), "calls", sum(k * v for k, v in h.items()))

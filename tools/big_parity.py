#!/usr/bin/env python3
"""One-off large-sample parity run on a GPU box: N million photons, GPU (C ABI) vs CPU oracle, bitwise; the chunks
alternate between the two kernel variants (open launch / plain launch).
Usage: python tools/big_parity.py [millions=32] [scene=demo|glass|replicated]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402

import _oracle as O  # noqa: E402
import robigo_luculenta_amd as R  # noqa: E402

millions = int(sys.argv[1]) if len(sys.argv) > 1 else 32
which = sys.argv[2] if len(sys.argv) > 2 else "demo"
objs, cam = {"demo": lambda: R.builtin_scene_desc(R.SCENE_DEMO), "glass": lambda: R.builtin_scene_desc(R.SCENE_GLASS_STRESS),
             "replicated": lambda: R.builtin_scene_desc(R.SCENE_DEMO, 158)}[which]()
scene = R.Scene(objs, cam)
oscene = O.Scene(objs, O.RlCameraDesc.from_buffer_copy(bytes(cam)))
N = 1 << 20
t = R.TraceUnit(0, 1920, 1080, n_photons=N)
threads = max(1, len(os.sched_getaffinity(0)))
try:
    q, p = open("/sys/fs/cgroup/cpu.max").read().split()
    if q != "max":
        threads = min(threads, max(1, int(int(q) / int(p))))
except Exception:
    pass
bad_total, segs_total, t0 = 0, 0, time.time()
for k in range(millions):
    first = 7_000_000_000 + k * N
    if k % 2 == 0:   # the blocking call: an open launch (rl_trace_unit_render)
        t.render(scene, seed=11, stream=k % 5, first_path_index=first)
    else:            # a plain launch of its own (rl_trace_unit_render_async + sync)
        t.render_async(scene, seed=11, stream=k % 5, first_path_index=first)
        t.sync()
    got = t.mapped_photons
    want, segs = oscene.render(1920, 1080, 11, k % 5, first, N, threads=threads)
    segs_total += segs
    if got.tobytes() != want.tobytes():
        a, b = want.view(np.uint32).reshape(-1, 4), got.view(np.uint32).reshape(-1, 4)
        bad = np.where((a != b).any(1))[0]
        bad_total += len(bad)
        print("chunk", k, "mismatches", len(bad), bad[:5], want[bad[:2]], got[bad[:2]])
print("%s: %d photons, %d rays, %d mismatching photons, GPU segments %d, %.1f s"
      % (which, millions * N, segs_total, bad_total, t.stats()[1], time.time() - t0))

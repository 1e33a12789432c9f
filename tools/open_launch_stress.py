#!/usr/bin/env python3
"""Stress of the open-launch protocol (rl_api.hip sessions / RlOpenCtl): 16 threads issue 3,000 blocking renders of 64
paths each -- one stash refill per call, so appends, completions, kernels closing themselves and new ones starting
race as often as they can -- with random pauses that let launches run dry.  Checks: no error, sampled results
bit-equal to the oracle, per-call path and segment counters exact in total, every call carried by some launch."""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import random
import threading
import time

import _oracle as O  # noqa: E402
import robigo_luculenta_amd as R  # noqa: E402
W, H, n, workers, rounds = 32, 18, 64, 16, 3000
objs, cam = R.builtin_scene_desc(R.SCENE_DEMO)
scene = R.Scene(objs, cam)
oscene = O.Scene(objs.view(O.OBJECT_DTYPE), O.RlCameraDesc.from_buffer_copy(bytes(cam)))
units = [R.TraceUnit(i, W, H, n_photons=n) for i in range(workers)]
kept, errors = {}, []
def work(i):
    rng = random.Random(i)
    try:
        for rnd in range(rounds):
            first = (rnd * workers + i) * n
            units[i].render(scene, seed=21, stream=0, first_path_index=first)
            if rnd % 500 == 7: kept[(i, rnd)] = (first, units[i].mapped_photons)
            r = rng.random()
            if r < 0.01: time.sleep(0.0003)      # lets launches run dry and close now and then
            elif r < 0.012: time.sleep(0.002)
    except Exception as e:
        errors.append(e)
t0 = time.time()
ts = [threading.Thread(target=work, args=(i,)) for i in range(workers)]
[t.start() for t in ts]; [t.join() for t in ts]
print("errors", errors, "%.1f s" % (time.time() - t0))
bad = 0
for (i, rnd), (first, got) in kept.items():
    want, _ = oscene.render(W, H, 21, 0, first, n, threads=1)
    bad += got.tobytes() != want.tobytes()
_, segs = oscene.render(W, H, 21, 0, 0, workers * rounds * n, threads=8)
st = [u.stats() for u in units]
print("mismatching samples", bad, "of", len(kept), "; paths", sum(s[0] for s in st), "expected", workers * rounds * n, "; segments", sum(s[1] for s in st), "expected", segs)
h = R.batch_histogram()
print("launches", sum(h.values()), "calls", sum(k * v for k, v in h.items()), "largest", max(h), "lone", h.get(1, 0))

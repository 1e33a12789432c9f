#!/usr/bin/env python3
"""Bit-exact parity over many random scenes (tests/_random_scene.py: arbitrary mixes of every surface and material
kind, nested / overlapping / huge spheres, randomly oriented prisms), both primitive-fetch modes: exercises the
conservative culls (sphere clusters, prism bounds) far away from the built-in scenes' regular layouts.
Usage (GPU box): python tools/random_scene_sweep.py [scenes=200] [photons=262144] [big]
"big": 2,000 - 8,000 spheres per scene -- too large for LDS, every record from global memory.  """
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402

import _oracle as O  # noqa: E402
import robigo_luculenta_amd as R  # noqa: E402
from _random_scene import random_scene  # noqa: E402

scenes = int(sys.argv[1]) if len(sys.argv) > 1 else 200
N = int(sys.argv[2]) if len(sys.argv) > 2 else 1 << 18
BIG = len(sys.argv) > 3 and sys.argv[3] == "big"
threads = max(1, len(os.sched_getaffinity(0)))
try:
    q, p = open("/sys/fs/cgroup/cpu.max").read().split()
    if q != "max":
        threads = min(threads, max(1, int(int(q) / int(p))))
except Exception:
    pass
rng = np.random.default_rng(2024)
bad, rays, t0 = 0, 0, time.time()
sizes = []
for k in range(scenes):
    n_spheres = int(rng.choice([2000, 2500, 3000, 4000, 5000, 6500, 8000] if BIG else [3, 9, 25, 39, 40, 41, 77, 130, 200, 333, 512, 800]))
    n_prisms = int(rng.choice([0, 0, 1, 2, 5, 9, 17, 30, 45, 80]))   # from 40 on: the prisms' second bound (CYL kernels)
    objs, cam = random_scene(1000 + k, n_spheres=n_spheres, n_prisms=n_prisms, n_planes=int(rng.integers(0, 4)),
                             n_circles=int(rng.integers(0, 4)), n_parabs=int(rng.integers(0, 3)))
    scene = R.Scene(objs, cam)
    oscene = O.Scene(objs, O.RlCameraDesc.from_buffer_copy(bytes(cam)))
    want, segs = oscene.render(640, 360, 1000 + k, k % 7, 10_000_000 * k, N, threads=threads)
    rays += segs
    sizes.append(len(objs))
    for fetch in ((R.FETCH_LDS,) if BIG else (R.FETCH_LDS, R.FETCH_GLOBAL)):   # (a big scene spills to global fetch by itself)
        t = R.TraceUnit(0, 640, 360, n_photons=N)
        t.set_fetch(fetch)
        if k % 2 == 0:   # an open launch (the blocking call) / a plain launch of its own: the two kernel variants
            t.render(scene, seed=1000 + k, stream=k % 7, first_path_index=10_000_000 * k)
        else:
            t.render_async(scene, seed=1000 + k, stream=k % 7, first_path_index=10_000_000 * k)
            t.sync()
        if t.mapped_photons.tobytes() != want.tobytes() or t.stats()[1] != segs:
            bad += 1
            print("MISMATCH scene", k, "spheres", n_spheres, "prisms", n_prisms, "fetch", fetch)
print("%d random scenes (%d..%d objects), %d photons each, %s: %d rays checked, %d mismatching renders, %.1f s"
      % (scenes, min(sizes), max(sizes), N, "too large for LDS (tables staged, spheres and objects from L2 / HBM)" if BIG else "both fetch modes", (1 if BIG else 2) * rays, bad, time.time() - t0))

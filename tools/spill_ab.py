#!/usr/bin/env python3
"""Scenes too large for LDS: the hybrid stage (RL_FETCH_GLOBAL: tables + 8-byte members staged, exact records from global memory)
against nothing staged (RL_FETCH_GLOBAL_ALL), Mrays/s of a fused launch.  Usage (GPU box): python tools/spill_ab.py"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import robigo_luculenta_amd as R  # noqa: E402
import _random_scene as RS  # noqa: E402

W, H = 1280, 720
cases = [("demo(seeds=158)", lambda: R.builtin_scene_desc(R.SCENE_DEMO, 158)), ("demo(seeds=600)", lambda: R.builtin_scene_desc(R.SCENE_DEMO, 600)),
         ("demo(seeds=1500)", lambda: R.builtin_scene_desc(R.SCENE_DEMO, 1500)), ("demo(seeds=2500)", lambda: R.builtin_scene_desc(R.SCENE_DEMO, 2500))]
for n in (2000, 5000, 8000):
    cases.append(("random(%d spheres)" % n, lambda n=n: RS.random_scene(40 + n, n_spheres=n, n_prisms=12, n_planes=2, n_circles=3, n_parabs=1)))
for name, make in cases:
    objs, cam = make()
    t0 = time.perf_counter()
    scene = R.Scene(objs, cam)
    t_scene = time.perf_counter() - t0
    row = []
    for fetch in (R.FETCH_LDS, R.FETCH_GLOBAL, R.FETCH_GLOBAL_ALL):
        t = R.TraceUnit(0, W, H, n_photons=64)
        t.set_fetch(fetch)
        p = R.PlotUnit(0, W, H)
        n = 16 * R.NUMBER_OF_PHOTONS
        t.render_fused(scene, p, n, seed=1, stream=0, first_path_index=0)
        t.sync()
        _, s0, ms0 = t.stats()
        t.render_fused(scene, p, n, seed=1, stream=0, first_path_index=n)
        t.sync()
        _, s1, ms1 = t.stats()
        row.append((s1 - s0) / ((ms1 - ms0) * 1e-3) / 1e6)
    print("%-24s %6d objects  scene %.2f s   default %7.0f   hybrid %7.0f   nothing staged %7.0f  Mrays/s" % (name, len(objs), t_scene, row[0], row[1], row[2]), flush=True)

"""Scenes too large for LDS: the default (RL_FETCH_LDS: the scene's tables staged, spheres and objects from L2 / HBM) against
nothing staged (RL_FETCH_GLOBAL), fused launches of 16 batches at 1280x720.  Usage (through gpurun): python tools/spill_ab.py"""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import robigo_luculenta_amd as R  # noqa: E402
import _random_scene as RS  # noqa: E402

W, H, N = 1280, 720, 16 * 524288


def rate(scene, fetch):
    t = R.TraceUnit(0, W, H, n_photons=64)
    t.set_fetch(fetch)
    p = R.PlotUnit(0, W, H)
    before = R.variant_launches()
    t.render_fused(scene, p, N, seed=1, stream=0, first_path_index=0)
    t.sync()
    _, s0, _ = t.stats()
    t0 = time.perf_counter()
    for k in range(3):
        t.render_fused(scene, p, N, seed=1, stream=0, first_path_index=(k + 1) * N)
    t.sync()
    dt = time.perf_counter() - t0
    _, s1, _ = t.stats()
    ran = [a - b for a, b in zip(R.variant_launches(), before)]
    where = "tables" if sum(ran[16:]) else ("all" if sum(ran[8:16]) else "none")
    return (s1 - s0) / dt / 1e9, where


scenes = [("built-in, 158 seeds (513 objects)", R.builtin_scene_desc(R.SCENE_DEMO, 158)),
          ("built-in, 600 seeds", R.builtin_scene_desc(R.SCENE_DEMO, 600)),
          ("built-in, 1500 seeds", R.builtin_scene_desc(R.SCENE_DEMO, 1500)),
          ("built-in, 2500 seeds", R.builtin_scene_desc(R.SCENE_DEMO, 2500)),
          ("random, 5000 spheres", RS.random_scene(31, n_spheres=5000, n_prisms=12, n_planes=2, n_circles=3, n_parabs=1)),
          ("random, 8000 spheres", RS.random_scene(34, n_spheres=8000, n_prisms=12, n_planes=2, n_circles=3, n_parabs=1)),
          ("random, 20000 spheres", RS.random_scene(35, n_spheres=20000, n_prisms=12, n_planes=2, n_circles=3, n_parabs=1))]
print("Grays/s, fused launches of 16 batches at %dx%d: default (what it staged) / nothing staged" % (W, H))
for name, (objs, cam) in scenes:
    scene = R.Scene(objs, cam)
    a, where = rate(scene, R.FETCH_LDS)
    b = float("nan") if os.environ.get("SPILL_QUICK") else rate(scene, R.FETCH_GLOBAL)[0]   # (SPILL_QUICK=1: the default column only)
    print("%-36s %6d objects  %6.2f (%s) / %6.2f" % (name, len(objs), a, where, b), flush=True)

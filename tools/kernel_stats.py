#!/usr/bin/env python3
"""Wave-level event counts of the trace kernel (diagnostic build `make -C robigo_luculenta_amd/csrc stats`,
-DRL_STATS): how many compaction rounds of each kind an iteration runs and how full they are, which
material branches a wave enters, how often the stash is refilled.  Sizes DESIGN.md's instruction budget.
Usage (GPU box): python tools/kernel_stats.py [batches=16] [scene=demo|glass|replicated|spill]"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ["RL_LIBRARY"] = os.path.join(ROOT, "robigo_luculenta_amd", "librobigo_luculenta_stats.so")
sys.path.insert(0, ROOT)
import robigo_luculenta_amd as R  # noqa: E402
from robigo_luculenta_amd import _lib  # noqa: E402

NAMES = ["iter", "scan_lanes", "a_rounds", "a_lanes", "b_rounds", "b_lanes", "p_rounds", "p_lanes", "shade_diffuse",
         "shade_glass", "shade_soap", "end_emitter", "end_void", "any_glass", "any_soap", "any_coloured", "any_glossy",
         "refills", "emit_batches", "emit_lanes", "a_items", "p_items", "any_diffuse", "s_rounds", "s_lanes", "s_items", "p_slow",
         "t_total", "t_refill", "t_small", "t_direct", "t_cluster", "t_tail", "t_prism", "t_shade", "t_emit", "t_a_rounds",
         "t_b_rounds", "t_p_rounds", "t_camera", "t_s_rounds"]

def _random(seed, n):  # the scenes of tools/spill_ab.py
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import _random_scene as RS
    return RS.random_scene(seed, n_spheres=n, n_prisms=12, n_planes=2, n_circles=3, n_parabs=1)


batches = int(sys.argv[1]) if len(sys.argv) > 1 else 16
which = sys.argv[2] if len(sys.argv) > 2 else "demo"
objs, cam = {"demo": lambda: R.builtin_scene_desc(R.SCENE_DEMO), "glass": lambda: R.builtin_scene_desc(R.SCENE_GLASS_STRESS),
             "replicated": lambda: R.builtin_scene_desc(R.SCENE_DEMO, 158), "spill": lambda: R.builtin_scene_desc(R.SCENE_DEMO, 1500),
             "spill2500": lambda: R.builtin_scene_desc(R.SCENE_DEMO, 2500),
             "random5k": lambda: _random(31, 5000), "random8k": lambda: _random(34, 8000), "random20k": lambda: _random(35, 20000)}[which]()
scene = R.Scene(objs, cam)
t, plot = R.TraceUnit(0, 1920, 1080, n_photons=64), R.PlotUnit(0, 1920, 1080)
read = _lib.lib.rl_stats_read
read.restype, read.argtypes = C.c_int, [C.POINTER(C.c_uint64), C.c_int]
buf = (C.c_uint64 * 48)()
read(buf, 48)  # clear
t.render_fused(scene, plot, batches * R.NUMBER_OF_PHOTONS, seed=1, stream=0, first_path_index=0)
t.sync()
assert read(buf, 48) == 0
c = dict(zip(NAMES, list(buf)))
paths, segs, ms = t.stats()
it = float(c["iter"])
print("%s: %d paths, %d rays, %d wave-iterations, %.1f rays per wave-iteration (of 64)" % (which, paths, segs, c["iter"], segs / it))
print("  scan lanes active            %5.1f %%" % (100.0 * c["scan_lanes"] / (64 * it)))
for key, label in (("s", "group (ring S) rounds "), ("a", "cluster-member rounds"), ("b", "sphere-tail rounds    "), ("p", "prism CSG rounds     ")):
    r, l = c[key + "_rounds"], c[key + "_lanes"]
    print("  %s  %.2f per iteration, %4.1f %% of lanes filled" % (label, r / it, 100.0 * l / max(1, 64 * r)))
print("  prism rounds that evaluated the Compound tree (the shortcut left a pair undecided): %.2f %%" % (100.0 * c["p_slow"] / max(1, c["p_rounds"])))
print("  (cluster, ray) pairs         %.1f per iteration = %.2f per ray;  (prism, ray) pairs %.1f = %.2f per ray"
      % (c["a_items"] / it, c["a_items"] / float(c["scan_lanes"]), c["p_items"] / it, c["p_items"] / float(c["scan_lanes"])))
hits = float(c["scan_lanes"])
print("  outcome per ray: diffuse family %.1f %%, glass %.1f %%, soap %.1f %%, ended on a light %.1f %%, void %.1f %%"
      % tuple(100.0 * c[k] / hits for k in ("shade_diffuse", "shade_glass", "shade_soap", "end_emitter", "end_void")))
print("  iterations entering a branch: diffuse %.1f %%, glass %.1f %%, soap %.1f %%, coloured (f64 exp) %.1f %%, glossy %.1f %%"
      % tuple(100.0 * c[k] / it for k in ("any_diffuse", "any_glass", "any_soap", "any_coloured", "any_glossy")))
print("  stash refills %.3f per iteration; emitter batches %.3f per iteration, %.1f %% of lanes filled"
      % (c["refills"] / it, c["emit_batches"] / it, 100.0 * c["emit_lanes"] / max(1, 64 * c["emit_batches"])))
# Where a wave's cycles go (s_memtime around each region of the main loop, summed over all waves; the
# reads themselves wait for outstanding LDS/scalar loads, so this build runs ~10 % slower than the product).
tt = float(c["t_total"])
if tt > 0:
    print("  wave cycles per iteration %.0f (shader clock); share of a wave's time by region:" % (tt / it))
    rows = (("refill (stash hand-out + camera rays)", "t_refill"), ("  of which rl_begin_path", "t_camera"),
            ("planes / circles / paraboloids", "t_small"), ("direct spheres", "t_direct"),
            ("cluster group culls + rounds", "t_cluster"), ("  of which member rounds", "t_a_rounds"),
            ("group (ring S) rounds, clusters and prisms", "t_s_rounds"),
            ("final sphere-tail flush", "t_tail"), ("  all sphere-tail rounds", "t_b_rounds"),
            ("prism culls + CSG rounds", "t_prism"), ("  of which CSG rounds", "t_p_rounds"),
            ("bounce (hit completion, material, roulette)", "t_shade"), ("emitter queue + splat", "t_emit"))
    for label, key in rows:
        print("    %-46s %5.1f %%  (%6.0f cycles per iteration)" % (label, 100.0 * c[key] / tt, c[key] / it))

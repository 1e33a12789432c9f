"""What the DEVICE does un-fused, without the worker pool: plain launches of 64 batches into mapped_photons + PlotUnit::plot, against
the same paths fused; 1280x720, built-in scene.  (The App's un-fused figure cannot beat the first.)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import robigo_luculenta_amd as R
W, H, B = 1280, 720, 524288
objs, cam = R.builtin_scene_desc(R.SCENE_DEMO)
scene = R.Scene(objs, cam)
n = 64 * B
plot = R.PlotUnit(0, W, H)
for mode in ("fused", "un-fused, 1 unit", "un-fused, 2 units alternating"):
    units = [R.TraceUnit(i, W, H, n_photons=n) for i in range(1 if "2 units" not in mode else 2)]
    nxt = 0
    def step(k):
        global nxt
        u = units[k % len(units)]
        if mode == "fused":
            u.render_fused(scene, plot, n, seed=1, stream=0, first_path_index=nxt)
        else:
            u.render_async(scene, seed=1, stream=0, first_path_index=nxt)
            plot.plot([u])
        nxt += n
    for k in range(2): step(k)
    for u in units: u.sync()
    plot.sync()
    s0 = sum(u.stats()[1] for u in units)
    t0 = time.perf_counter()
    for k in range(8): step(k)
    for u in units: u.sync()
    plot.sync()
    t1 = time.perf_counter()
    s1 = sum(u.stats()[1] for u in units)
    print("%-32s %6d Mrays/s" % (mode, round((s1 - s0) / (t1 - t0) / 1e6)), flush=True)
    del units

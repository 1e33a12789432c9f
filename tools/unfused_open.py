"""Where does the un-fused App lose against the fused one?  16 host threads issue blocking renders of one batch each (open launches,
the un-fused OPEN kernel) for 1280x720 -- with and without a PlotUnit::plot behind every 8 renders."""
import os, sys, time, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import robigo_luculenta_amd as R
W, H, B = 1280, 720, 524288
objs, cam = R.builtin_scene_desc(R.SCENE_DEMO)
scene = R.Scene(objs, cam)
T, ROUNDS, PER = 16, 24, 8
for with_plot in (False, True, False, True):
    units = [[R.TraceUnit(t * PER + i, W, H, n_photons=B) for i in range(PER)] for t in range(T)]
    plots = [R.PlotUnit(t, W, H) for t in range(T)]
    def work(t):
        for r in range(ROUNDS):
            for i, u in enumerate(units[t]):
                u.render_begin(scene, seed=1, stream=0, first_path_index=((r * T + t) * PER + i) * B)
            if with_plot:
                plots[t].plot(units[t])     # ends the begun renders, then one plot kernel per unit on the plot stream
            else:
                for u in units[t]: u.render_end()
    t0 = time.perf_counter()
    ts = [threading.Thread(target=work, args=(t,)) for t in range(T)]
    [x.start() for x in ts]; [x.join() for x in ts]
    for p in plots: p.sync()
    t1 = time.perf_counter()
    segs = sum(u.stats()[1] for us in units for u in us)
    print("un-fused open launches, %s: %6d Mrays/s" % ("render + plot" if with_plot else "render only  ", round(segs / (t1 - t0) / 1e6)), flush=True)
    del units, plots

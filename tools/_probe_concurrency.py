# Experiment: can small kernels on other streams run while an open trace launch is resident?
import time, os, threading
import numpy as np
import robigo_luculenta_amd as R
W, H = 1280, 720
scene = R.Scene.builtin()
tu = R.TraceUnit(0, W, H)
tu2 = R.TraceUnit(1, W, H)
extra = [R.TraceUnit(2 + i, W, H) for i in range(int(os.environ.get("EXTRA_UNITS", "0")))]
pa, pb = R.PlotUnit(0, W, H), R.PlotUnit(1, W, H)
g = R.GatherUnit(W, H)
big = 64 * 524288
tu.render_fused_sync(scene, pa, big)
tu2.render(scene); pb.plot([tu2]); pb.sync(); g.accumulate(pb); g.sync()
def timed(f):
    t0 = time.perf_counter(); f(); return (time.perf_counter() - t0) * 1e3
print("plot alone %.3f ms" % timed(lambda: (pb.plot([tu2]), pb.sync())))
print("gather alone %.3f ms" % timed(lambda: (g.accumulate(pb), g.sync())))
for rep in range(3):
    th = threading.Thread(target=lambda: tu.render_fused_sync(scene, pa, big))
    t0 = time.perf_counter()
    th.start()
    time.sleep(0.003)
    tp = timed(lambda: (pb.plot([tu2]), pb.sync()))
    tg = timed(lambda: (g.accumulate(pb), g.sync()))
    t_mid = (time.perf_counter() - t0) * 1e3
    th.join()
    t_all = (time.perf_counter() - t0) * 1e3
    print("during an open launch: plot %.3f ms, gather %.3f ms (done at %.2f ms); trace call returned at %.2f ms" % (tp, tg, t_mid, t_all))

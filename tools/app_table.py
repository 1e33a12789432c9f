"""rl_app_run at the reference's task size: scheduler depth (concurrency) x host threads, fused and un-fused, blocking or not.
Usage (through gpurun): python tools/app_table.py [quick]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import robigo_luculenta_amd as R
quick = len(sys.argv) > 1
print("rl_app_run, built-in scene, 1280x720, 4096 batches of 524288 paths (trace_unit.rs:67); seconds include the final tonemap")
print("depth = RlAppConfig::concurrency (3 x depth trace units, depth / 2 plot units, task_scheduler.rs:95-96), threads = host workers")
def run(fused, depth, threads, blocking=False):
    rgb, st = R.app_run(1280, 720, 4096, concurrency=depth, threads=threads, photons_per_batch=524288, fused=fused, blocking_trace=blocking, verbose=False)
    print("  %-8s %-9s depth %2d threads %2d  %.3f s  %6d Mrays/s  %6d batches/s  tasks %s" % (
        "fused" if fused else "un-fused", "blocking" if blocking else "", depth, threads or depth, st["seconds"], round(st["segments"] / st["seconds"] / 1e6),
        round(st["paths"] / 524288 / st["seconds"]), {k: v for k, v in st["tasks"].items() if k != "tonemap"}), flush=True)
for fused in (False, True):
    for depth, threads in ((1, 1), (4, 4), (16, 16), (16, 1), (16, 2), (16, 4), (32, 1), (32, 2), (32, 4), (64, 2), (64, 4)):
        if quick and (depth, threads) in ((1, 1), (4, 4), (64, 4)): continue
        run(fused, depth, threads)
for fused in (False, True):
    for depth, threads in ((16, 16), (32, 32), (48, 48)):   # (more workers than this box has cores: the waits sleep, rl_api.hip session_wait)
        run(fused, depth, threads, True)
print("open launches so far, {calls carried: launches}:", R.batch_histogram())

import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import robigo_luculenta_amd as R
print("rl_app_run, built-in scene, 1280x720, 4096 batches of 524288 paths")
for fused in (False, True):
    for c in (1, 4, 8, 16):
        rgb, st = R.app_run(1280, 720, 4096, concurrency=c, photons_per_batch=524288, fused=fused, verbose=False)
        print(" ", "fused" if fused else "un-fused", "workers", c, round(st["seconds"], 3), "s", round(st["segments"] / st["seconds"] / 1e6), "Mrays/s", flush=True)

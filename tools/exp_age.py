import os, sys
sys.path.insert(0, os.getcwd())
import robigo_luculenta_amd as R
for fused, depth, threads, blocking in ((False, 16, 16, True), (False, 32, 1, False), (False, 16, 16, False)):
    rgb, st = R.app_run(1280, 720, 4096, concurrency=depth, threads=threads, photons_per_batch=524288, fused=fused, blocking_trace=blocking, verbose=False)
    print(os.environ.get("RL_LIBRARY", "tree")[-12:], "un-fused", "blocking" if blocking else "", depth, threads, round(st["segments"] / st["seconds"] / 1e6), flush=True)

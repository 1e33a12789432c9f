"""A Trace task that WAITS for its own batch, as a literal reference worker does (RlAppConfig::blocking_trace), with more workers than
the host has cores (the reference starts num_cpus::get() workers, app.rs:55; the waits sleep, they do not spin).
Usage (through gpurun): python tools/exp_blocking_workers.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import robigo_luculenta_amd as R
for fused in (False, True):
    for threads in (16, 24, 32, 48, 64):
        rgb, st = R.app_run(1280, 720, 4096, concurrency=threads, threads=threads, photons_per_batch=524288, fused=fused, blocking_trace=True, verbose=False)
        print("fused" if fused else "un-fused", "blocking, workers", threads, round(st["segments"] / st["seconds"] / 1e6), "Mrays/s", flush=True)

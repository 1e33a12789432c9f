#!/usr/bin/env python3
"""Image-level parity report on a GPU box (SURVEY 8d "Image parity"): the fused GPU path vs the CPU oracle at
the same (seed, stream, path budget), through plot -> Kahan gather -> tonemap -> sRGB.
Prints max |delta| of the float sRGB (north star: <= 1e-3), max |delta| of the RGB8 bytes and the PSNR.
Usage: python tools/image_parity.py [width=1280] [height=720] [batches=32] [scene=demo|glass]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402

import _oracle as O  # noqa: E402
import robigo_luculenta_amd as R  # noqa: E402

W = int(sys.argv[1]) if len(sys.argv) > 1 else 1280
H = int(sys.argv[2]) if len(sys.argv) > 2 else 720
batches = int(sys.argv[3]) if len(sys.argv) > 3 else 32
which = sys.argv[4] if len(sys.argv) > 4 else "demo"
N = R.NUMBER_OF_PHOTONS
objs, cam = R.builtin_scene_desc(R.SCENE_DEMO if which == "demo" else R.SCENE_GLASS_STRESS)
scene = R.Scene(objs, cam)
oscene = O.Scene(objs, O.RlCameraDesc.from_buffer_copy(bytes(cam)))
threads = max(1, len(os.sched_getaffinity(0)))
try:
    q, p = open("/sys/fs/cgroup/cpu.max").read().split()
    if q != "max":
        threads = min(threads, max(1, int(int(q) / int(p))))
except Exception:
    pass

t, plot, gather, tm = R.TraceUnit(0, W, H, n_photons=64), R.PlotUnit(0, W, H), R.GatherUnit(W, H), R.TonemapUnit(W, H)
acc = np.zeros((W * H, 3), np.float32)
comp = np.zeros_like(acc)
t0 = time.time()
for k in range(batches):
    t.render_fused(scene, plot, N, seed=1, stream=0, first_path_index=k * N)
    gather.accumulate(plot)                                   # one gather per batch on both sides
    photons, _ = oscene.render(W, H, 1, 0, k * N, N, threads=threads)
    O.accumulate(acc, comp, O.plot(W, H, photons))
tm.tonemap(gather)
got, got_max = tm.srgb_float()
rgb_want, want, want_max = O.tonemap(acc, W, H)
d = np.abs(got.astype(np.float64) - want.astype(np.float64))
d8 = np.abs(tm.rgb_buffer.astype(int).reshape(-1) - np.asarray(rgb_want).astype(int).reshape(-1))
mse = float((d8.astype(np.float64) ** 2).mean())
psnr = float("inf") if mse == 0 else 10 * np.log10(255.0 ** 2 / mse)
xyz = gather.tristimulus_buffer
rel = np.abs(xyz - acc).max() / np.abs(acc).max()
print("%s %dx%d, %d batches (%d paths), seed 1: max |d sRGB float| = %.3g (bar 1e-3), mean = %.3g, max |d RGB8| = %d LSB, "
      "differing bytes = %d of %d, PSNR = %s dB, exposure %.9g vs %.9g, max |d XYZ| / max XYZ = %.3g, %.1f s"
      % (which, W, H, batches, batches * N, d.max(), d.mean(), d8.max(), int((d8 != 0).sum()), d8.size,
         "inf" if mse == 0 else "%.1f" % psnr, got_max, want_max, rel, time.time() - t0))
